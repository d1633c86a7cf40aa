// k_surface.hip -- where does a ray from infinity meet the photosphere of a geometrically thick disk?
//
// Batched counterpart of the reference's Python surface search (python/sim5diskraytrace.py:
// DiskRaytrace.geodesic :214-253 with flat=False and __find_surface :257-335): start far out on the
// incoming branch (geodesic_P_int at r0), walk down the geodesic with geodesic_follow in steps tied to the
// height above the surface, back up and refine when the surface is crossed, fall back to the equatorial
// crossing when the ray reaches the midplane, retry from further out when it escapes.  One lane follows one
// ray through the same sequence of SIM5 calls, decisions and constants as the Python code; the three
// nested levels (retry, walk, sub-steps of geodesic_follow) all carry hard caps.
//
// The disk surface H(R) is what the reference gets from the Python disk model's h(R).  Here it is a table
// (R_i ascending, H_i) staged once per workgroup into LDS and interpolated linearly, with a constant opening
// angle beyond the last point and H[0] below the first; every lane evaluates it many hundred times.
#define S5_LADDER_IN_LDS 1            // 256-thread 1-D workgroups: Landen rungs of position_rad / position_pol in LDS
#include "s5_disk.hpp"
#include "kernels.hpp"

namespace S5NS {

using namespace s5abi;

constexpr int SURF_MAX_TABLE = 4096;        // 2 x 32 KiB of LDS

// slope of the same piecewise-linear surface (what a disk model built on the table returns as dhdr) and
// the node values of another profile interpolated the same way
S5_DEV void surface_segment(const double* sR, int n, double R, int& lo, int& hi)
{
    lo = 0; hi = n - 1;                      // invariant sR[lo] < R <= sR[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (sR[mid] < R) lo = mid; else hi = mid;
    }
}

S5_DEV double surface_height(const double* sR, const double* sH, int n, double R)
{
    if (!(R > sR[0])) return sH[0];
    if (R >= sR[n - 1]) return sH[n - 1] * (R / sR[n - 1]);
    int lo = 0, hi = n - 1;                  // invariant sR[lo] < R <= sR[hi] ... bisection
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (sR[mid] < R) lo = mid; else hi = mid;
    }
    const double w = (R - sR[lo]) / (sR[hi] - sR[lo]);
    return sH[lo] + w * (sH[hi] - sH[lo]);
}

#ifndef S5_SURF_WAVES
#define S5_SURF_WAVES 2
#endif
__global__ __launch_bounds__(256, S5_SURF_WAVES)
void disk_surface_kernel(SurfaceParams p, const double* __restrict__ tabR, const double* __restrict__ tabH,
                         const double* __restrict__ alpha, const double* __restrict__ beta,
                         double* __restrict__ outP, double* __restrict__ outR, double* __restrict__ outM,
                         double* __restrict__ outK, int* __restrict__ outStatus)
{
    extern __shared__ double lds[];
    double* sR = lds;
    double* sH = lds + p.n_table;
    for (int i = threadIdx.x; i < p.n_table; i += 256) { sR[i] = tabR[i]; sH[i] = tabH[i]; }
    __syncthreads();

    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;

    int status = 0;                          // 1 = surface point found, 0 = no intersection / error
    double P = NAN, r = 0.0, m = 0.0;
    double kout[4] = { NAN, NAN, NAN, NAN };

    Geod gd;
    GeodCache cache;
    int err = 0;
    const bool ok = init_inf(p.incl, p.sin_i, p.cos_i, p.a, alpha[i], beta[i], gd, err, cache);
    if (ok) {
        const double accuracy = 1e-2;                                            // ref py :268
        const double rbh = r_horizon(p.a);
        const double disk_theta = atan(surface_height(sR, sH, p.n_table, 1e6) / 1e6);   // :263
        bool found = false, failed = false;
        for (int iteration = 0; iteration <= 3 && !found && !failed; ++iteration) {     // :258, :329
            double r0 = fmax(fmax(200.0, 1.1 * gd.rp),
                             (0.5 + iteration) * sqrt(gd.alpha * gd.alpha + gd.beta * gd.beta) /
                                 cos(gd.incl + disk_theta));                     // :265
            double P1 = NAN, r1 = NAN, m1 = NAN, H1 = NAN, Hd = NAN;
            for (int grow = 0; grow < 64; ++grow) {                              // :271-280
                P1 = P_int(gd, r0, 0);
                r1 = position_rad(gd, P1);
                m1 = position_pol(gd, P1);
                const double R1 = r1 * sqrt(1. - m1 * m1);
                H1 = r1 * m1;
                Hd = surface_height(sR, sH, p.n_table, R1);
                if ((Hd < H1) || (r0 > 5e6)) break;
                r0 = 2.0 * r0;
            }
            if (!(Hd < H1)) { failed = true; break; }                            // :283 (Hd >= H1, or NaN)
            P = P1; r = r1; m = m1;
            double step_factor = 1.0;
            bool again = false;
            for (long it = 0; it < 2000000; ++it) {                              // :296-331
                const double step = fmax(accuracy / 2., fmin((H1 - Hd) / 2., 0.5 * (sqrt(r) - 0.99) * step_factor));
                int st = 0;
                follow(gd, step, P, r, m, st);
                if (!st) { failed = true; break; }
                const double R1 = r * sqrt(1. - m * m);
                H1 = r * m;
                Hd = surface_height(sR, sH, p.n_table, R1);
                if (H1 <= Hd) {                                                  // surface hit? :307
                    if (step < accuracy) {
                        follow(gd, -step / 2., P, r, m, st);
                        found = true; break;
                    }
                    follow(gd, -step, P, r, m, st);
                    step_factor = step_factor / 5.;
                    continue;
                }
                if (H1 < 1e-4) {                                                 // equatorial plane hit? :316
                    P = midplane_crossing(gd, 0, cache);
                    r = position_rad(gd, P);
                    m = position_pol(gd, P);
                    found = true; break;
                }
                if (r < 1.05 * rbh) { failed = true; break; }                    // :324
                if (r > 1.1 * r0) { again = true; break; }                       // :325 retry from further out
                if (m < 0.0) { failed = true; break; }                           // :326
                if (step < accuracy / 2.) { failed = true; break; }              // :327, then :331
            }
            if (!found && !failed && !again) failed = true;
        }
        if (found && !isnan(r) && !isnan(P)) {                                   // ref py :244-248
            status = 1;
            photon_momentum(p.a, r, m, gd.l, gd.q, gd.Rpc - P, 1.0, kout);       // :250
        } else {
            P = NAN; r = 0.0; m = 0.0;
        }
    }
    if (p.out_g) {
        // local frame of the disk surface at the point found: the reference's __tetrad / __gfactor /
        // __emission_angle (python/sim5diskraytrace.py:340-390) with the slope of the tabulated surface, the
        // Novikov-Thorne angular momentum and flux, and the tabulated radial velocity
        double gfac = NAN, mue = NAN, F = NAN;
        if (status == 1) {
            const double R = r * sqrt(1. - m * m);
            F = disk_flux(p.disk, R);
            double dhdr = 0.0, V = 0.0;
            const int nt = p.n_table;
            if (!(R > sR[0])) { dhdr = 0.0; V = p.tab_vr ? p.tab_vr[0] : 0.0; }
            else if (R >= sR[nt - 1]) { dhdr = sH[nt - 1] / sR[nt - 1]; V = p.tab_vr ? p.tab_vr[nt - 1] : 0.0; }
            else {
                int lo, hi;
                surface_segment(sR, nt, R, lo, hi);
                dhdr = (sH[hi] - sH[lo]) / (sR[hi] - sR[lo]);
                if (p.tab_vr) { const double w = (R - sR[lo]) / (sR[hi] - sR[lo]); V = p.tab_vr[lo] + w * (p.tab_vr[hi] - p.tab_vr[lo]); }
            }
            if (!(m > 0.0)) dhdr = 0.0;                                           // ref py :346
            Metric mt;
            kerr_metric(p.a, r, m, mt);
            Tetrad tt;
            tetrad_surface(mt, omega_from_ell(disk_ell(p.disk, R), mt), V, dhdr, tt);
            const double e0[4] = { 1.0, 0.0, 0.0, 0.0 }, e2[4] = { 0.0, 0.0, 1.0, 0.0 };
            double U[4], N[4];
            on2bl(e0, U, tt);
            on2bl(e2, N, tt);
            const double kU = dot(kout, U, mt);
            gfac = (kout[0] * mt.g00 + kout[3] * mt.g03) / kU;
            if (!(gfac > 0.0)) gfac = 0.0;                                        // ref py :360
            mue = dot(kout, N, mt) / kU;
            if ((mue < 0.0) && (mue > -1e-2)) mue = 1e-3;                         // ref py :387
        }
        p.out_g[i] = gfac; p.out_mue[i] = mue; p.out_flux[i] = F;
    }
    outP[i] = P; outR[i] = r; outM[i] = m; outStatus[i] = status;
    if (outK) { outK[4 * i] = kout[0]; outK[4 * i + 1] = kout[1]; outK[4 * i + 2] = kout[2]; outK[4 * i + 3] = kout[3]; }
}

} // namespace S5NS

#if S5_FAST
int s5_launch_disk_surface_fast(const s5abi::SurfaceParams& p, const double* tabR, const double* tabH,
#else
int s5_launch_disk_surface_strict(const s5abi::SurfaceParams& p, const double* tabR, const double* tabH,
#endif
                                  const double* alpha, const double* beta, double* P, double* r, double* m,
                                  double* k, int* status, hipStream_t stream)
{
    using namespace S5NS;
    const unsigned blocks = (unsigned)((p.n + 255) / 256);
    const size_t lds = 2 * sizeof(double) * (size_t)p.n_table;
    hipLaunchKernelGGL(disk_surface_kernel, dim3(blocks), dim3(256), lds, stream, p, tabR, tabH, alpha, beta,
                       P, r, m, k, status);
    return (int)hipGetLastError();
}
