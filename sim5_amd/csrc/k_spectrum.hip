// k_spectrum.hip -- observed spectrum of a thin disk: image x energy grid in one pass (gfx950).
//
// What the reference does in Python per ray (python/sim5diskraytrace.py:96-123): find the disk crossing
// (first order only, :239-242), T_eff = (F/sigma)^1/4 (python/sim5diskmodel.py:47), local frame of the
// disk surface (tetrad_surface with Omega from the disk's specific angular momentum, :340-348), redshift
// g = k_t / (k.U) (:353-361), emission cosine mu_e = (k.N)/(k.U) (:377-390), and add
// I_nu(E_j / g) g^3 dOmega to every energy bin j with the black-body radiance of
// python/sim5diskspectrum.py:54-88 (colour hardening f, limb darkening 1/2 + 3/4 mu_e).
//
// Kernel: a workgroup traces a 32 x 8 pixel tile with the fused thin-disk routine, each lane leaves
// (T, g, limb-darkening factor) of its pixel in LDS, then the workgroup is re-used TRANSPOSED: thread t owns
// energy bin j = t mod EB and pixel sub-set t / EB, loops over the staged pixels (LDS broadcast reads, no
// conflicts), and accumulates its bin in a register -- no cross-lane reduction per energy.  The sub-set
// partial sums are combined through LDS in a fixed order and the workgroup writes one partial spectrum;
// a second kernel sums the partial spectra over workgroups in index order.  The result is deterministic
// (no atomics).  Output: sum over pixels of I_nu(E_j/g) g^3 in erg cm^-2 s^-1 keV^-1 srad^-1 per pixel; the
// caller multiplies by the pixel solid angle.
#include "s5_thindisk.hpp"
#include "kernels.hpp"

namespace S5NS {

using namespace s5abi;

constexpr int SPEC_TILE_W = 32, SPEC_TILE_H = 8;

S5_DEV double planck_python(double T, double limbf, double f, double E)
{
    // python/sim5diskspectrum.py:72-86 (its own constants; note kev2freq twice, not 1/freq2kev)
    const double planck_h = 6.626069e-27, kev2freq = 2.417990e+17, c2 = 8.987554e+20, kB = 1.380650e-16;
    const double nu = kev2freq * E;
    return limbf * 2.0 * planck_h * (nu * nu * nu) / c2 / (f * f * f * f) *
           1. / (exp((planck_h * kev2freq * E) / (kB * f * T)) - 1.0) * kev2freq;
}

__global__ __launch_bounds__(256, 2)
void disk_spectrum_kernel(ImageParams p, SpectrumParams sp, const double* __restrict__ energies,
                          double* __restrict__ partial)
{
    __shared__ double sT[256], sG[256], sL[256];
    __shared__ double sAcc[256];
    const int tid = threadIdx.x;
    const int lane_x = tid % SPEC_TILE_W, lane_y = tid / SPEC_TILE_W;
    const int ix = blockIdx.x * SPEC_TILE_W + lane_x;
    const int lr = blockIdx.y * SPEC_TILE_H + lane_y;
    double T = 0.0, g = 0.0, limbf = 0.0;
    if (ix < p.nx && lr < p.nrows) {
        const int iy = p.y0 + lr;
        const double alpha = (((double)(ix) + .5) / (double)(p.nx) - 0.5) * 2.0 * p.rmax;
        const double beta = (((double)(iy) + .5) / (double)(p.ny) - 0.5) * 2.0 * p.rmax *
                            ((double)p.ny / (double)p.nx);
        ThinRay t;
        trace_thin_disk<true>(p, alpha, beta, t);            // max_order = 1, rms = 0: first crossing, any radius
        if (t.cls == PX_HIT0 && t.flux != 0.0) {
            double k[4], U[4], N[4];
            const double e0[4] = { 1.0, 0.0, 0.0, 0.0 }, e2[4] = { 0.0, 0.0, 1.0, 0.0 };
            photon_momentum(p.a, t.r, 0.0, t.l, t.q, t.dP, 1.0, k);           // ref py :250
            Metric mt;
            kerr_metric(p.a, t.r, 0.0, mt);
            Tetrad tt;
            tetrad_surface(mt, omega_from_ell(disk_ell(p.disk, t.r), mt), 0.0, 0.0, tt);
            on2bl(e0, U, tt);
            on2bl(e2, N, tt);
            const double kU = dot(k, U, mt);
            double gg = mdiv(k[0] * mt.g00 + k[3] * mt.g03, kU);
            double mue = mdiv(dot(k, N, mt), kU);
            if ((mue < 0.0) && (mue > -1e-2)) mue = 1e-3;                            // ref py :387
            if (gg > 0.0) {
                g = gg;
                T = sqrt(sqrt(t.flux / 5.670400e-05));
                limbf = (sp.limb_darkening > 0) ? ((mue >= 0.0) ? 0.5 + 0.75 * mue : 1.0) : 1.0;
            }
        }
    }
#if S5_FAST
    // per-pixel factors of the Planck expression, so that a (pixel, energy) pair costs one exp, one division
    // and a few multiplications: x = E * sT, I_nu g^3 = sL * E^3 / (exp(x) - 1) with
    //   sT = h kev2freq / (kB f T g),   sL = limbf 2 h kev2freq^4 / (c^2 f^4)   (g^3 from nu^3 cancels the g^3 weight)
    {
        const double planck_h = 6.626069e-27, kev2freq = 2.417990e+17, c2 = 8.987554e+20, kB = 1.380650e-16;
        const bool on = (g > 0.0) && !(T < 1e2);                                 // ref py :76
        const double f = sp.hardening;
        const double xs = on ? mdiv(planck_h * kev2freq, kB * f * T * g) : 0.0;
        const double amp = on ? limbf * mdiv(2.0 * planck_h * (kev2freq * kev2freq * kev2freq) * kev2freq, c2 * (f * f * f * f)) : 0.0;
        // pixels that contribute nothing keep a harmless exponent (x = E) and amplitude 0: no branch below
        sT[tid] = on ? xs : 1.0; sG[tid] = on ? 1.0 : 0.0; sL[tid] = amp;
    }
#else
    sT[tid] = T; sG[tid] = g; sL[tid] = limbf;
#endif
    __syncthreads();

    // transposed phase: EB energy bins x (256 / EB) pixel sub-sets
    const int EB = sp.bins_per_pass;                     // power of two, <= 256
    const int groups = 256 / EB;
    const int jj = tid % EB, grp = tid / EB;
    for (int j0 = 0; j0 < sp.n_energies; j0 += EB) {
        const int j = j0 + jj;
        double acc = 0.0;
        if (j < sp.n_energies) {
            const double E = energies[j];
#if S5_FAST
            const double E3 = E * E * E;
            // x capped at 700: exp would overflow to inf, which the Newton division cannot take (the term is 0 to
            // 300 digits there either way).  Unrolled so that the LDS broadcasts of several pixels are in flight.
#pragma unroll 4
            for (int q = grp; q < 256; q += groups)
                acc += mdiv(sL[q] * E3, mexp(fmin(E * sT[q], 700.0)) - 1.0);
#else
            for (int q = grp; q < 256; q += groups) {
                const double gq = sG[q];
                if (gq > 0.0) {
                    const double Tq = sT[q];
                    if (!(Tq < 1e2))                         // ref py :76
                        acc += planck_python(Tq, sL[q], sp.hardening, mdiv(E, gq)) * (gq * gq * gq);
                }
            }
#endif
        }
        sAcc[tid] = acc;
        __syncthreads();
        if (grp == 0 && j < sp.n_energies) {
            double tot = 0.0;
            for (int s = 0; s < groups; ++s) tot += sAcc[s * EB + jj];
            const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            partial[blk * (size_t)sp.n_energies + j] = tot;
        }
        __syncthreads();
    }
}

// one level of the (deterministic) tree sum over workgroup partials: row c of dst = sum of rows
// [c * SPEC_FAN, (c + 1) * SPEC_FAN) of src, energies across the lanes (coalesced)
constexpr int SPEC_FAN = 64;
__global__ __launch_bounds__(256)
void spectrum_reduce_kernel(const double* __restrict__ src, size_t rows, int n_energies, double* __restrict__ dst)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_energies) return;
    const size_t r0 = (size_t)blockIdx.y * SPEC_FAN;
    const size_t r1 = (r0 + SPEC_FAN < rows) ? r0 + SPEC_FAN : rows;
    double tot = 0.0;
    for (size_t b = r0; b < r1; ++b) tot += src[b * (size_t)n_energies + j];
    dst[(size_t)blockIdx.y * n_energies + j] = tot;
}

} // namespace S5NS

#if S5_FAST
int s5_launch_disk_spectrum_fast(const s5abi::ImageParams& p, const s5abi::SpectrumParams& sp,
                                 const double* energies, double* partial, double* spectrum, hipStream_t stream)
#else
int s5_launch_disk_spectrum_strict(const s5abi::ImageParams& p, const s5abi::SpectrumParams& sp,
                                   const double* energies, double* partial, double* spectrum, hipStream_t stream)
#endif
{
    using namespace S5NS;
    const dim3 grid((p.nx + SPEC_TILE_W - 1) / SPEC_TILE_W, (p.nrows + SPEC_TILE_H - 1) / SPEC_TILE_H);
    hipLaunchKernelGGL(disk_spectrum_kernel, grid, dim3(256), 0, stream, p, sp, energies, partial);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    // tree sum, fan-in 64 per level, ping-pong between the partial rows and the scratch rows behind them
    // (workspace = (nblocks + ceil(nblocks / 64)) rows); the last level writes the spectrum
    const size_t nblocks = (size_t)grid.x * grid.y;
    double* bufs[2] = { partial, partial + nblocks * (size_t)sp.n_energies };
    size_t rows = nblocks;
    int cur = 0;
    for (;;) {
        const size_t out_rows = (rows + SPEC_FAN - 1) / SPEC_FAN;
        double* dst = (out_rows == 1) ? spectrum : bufs[cur ^ 1];
        hipLaunchKernelGGL(spectrum_reduce_kernel, dim3((sp.n_energies + 255) / 256, (unsigned)out_rows), dim3(256), 0, stream,
                           bufs[cur], rows, sp.n_energies, dst);
        if ((e = hipGetLastError()) != hipSuccess) return (int)e;
        if (out_rows == 1) break;
        rows = out_rows; cur ^= 1;
    }
    return 0;
}
