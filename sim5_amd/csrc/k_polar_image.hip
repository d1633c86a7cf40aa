// k_polar_image.hip -- thin-disk image with Walker-Penrose polarization transport (gfx950).
//
// Per ray: the thin-disk trace of k_disk_image.hip, then at the emission point
//   k      = geodesic_momentum(P, r, m=0)                      ref src/sim5kerr-geod.c:787-840
//   frame  = tetrad_azimuthal(kerr_metric(a,r,0), OmegaK(r,a)) ref src/sim5kerr.c:75,766,1037
//   n      = bl2on(k)                                          ref src/sim5kerr.c:926
//   f_loc  = N x n  with N the disk normal (local z)           -> (0, n_y, 0, -n_x)
//   f      = on2bl(f_loc), normalised to f.f = 1               ref src/sim5kerr.c:948,553
//   kappa  = polarization_constant(k, f)                       ref src/sim5polarization.c:145-158
//   chi    = polarization_angle_rotation(a, i, alpha, beta, kappa)          ref :272-285
//   I = F g^4,  Q = delta I cos 2chi,  U = delta I sin 2chi
// The reference has no end-to-end caller for this chain (its unit test only checks that kappa is
// conserved, src/sim5unittests.c:113-140); the chain is assembled from its public routines and is
// checked (tests/) against the same chain evaluated with the reference library.
#include "s5_thindisk.hpp"
#include "s5_polar.hpp"
#include "kernels.hpp"

namespace S5NS {

using namespace s5abi;

// the polarization chain of one traced ray (header comment): Stokes I, Q, U and the angle
// AUX = false: the instantiation for jobs that take the Stokes planes only (no chi plane, no full-precision planes): six
// pointers fewer held in SGPRs through the kernel, no atan2, no tests around the stores (as k_disk_image.hip)
template <bool AUX, class PRM>
S5_DEV void polarize_ray(const PRM& p, double alpha, double beta, const ThinRay& t, double& I, double& Q, double& U,
                         double& chi)
{
    I = 0.0; Q = 0.0; U = 0.0; chi = NAN;
    if (!(t.cls == PX_HIT0 || t.cls == PX_HIT1)) return;
    const double g2 = t.g * t.g;
    I = t.flux * (g2 * g2);
    // geodesic_momentum(gd, P, r, m = 0): sign of dm/dP from the polar phase, radial sign from P vs Rpc
    double wp[2];
#if S5_FAST
    // The polar phase at an EQUATORIAL crossing needs no bookkeeping.  The reference counts the polar turning points passed
    // by stepping T in units of Tpp until P <= T + Tpp (ref src/sim5kerr-geod.c:385-389); with P = mK ((2n+1) K +- cn^-1(u))
    // for the crossing of order n, Tpp = 2 mK K and Tip = mK cn^-1(u) that is n + 1 steps for beta > 0 and n for beta < 0 --
    // each test decided by a margin of mK K, no rounding can move it -- so sign(dm/dP) = -(-1)^n for either sign of beta.
    // With it the chain needs NOTHING of the geodesic but r, the crossing's order and the radial direction: l and q are formed
    // here from the pixel's coordinates (the expressions of ref :76-77, as trace_thin_disk_impl forms them), and the kernel no
    // longer carries Tpp, Tip, P, l, q, a, beta of both rays of a pair through the trace (148 -> see k_polar_image.hip's bounds).
    const double sdm = (t.cls == PX_HIT1) ? +1.0 : -1.0;
    const double a_in_ = p.a;
    const double t_a = fmax(1e-4, a_in_);
    const double beta_e = (beta == 0.0) ? +1e-6 : beta;
    const double t_l = -alpha * p.sin_i;
    const double t_q = sq(beta_e) + sq(p.cos_i) * (sq(alpha) - sq(a_in_));
#else
    double sdm = (t.beta >= 0.0) ? +1.0 : -1.0;
    double T = (sdm > 0.0) ? -(t.Tpp - t.Tip) : -(t.Tip);
    for (int it = 0; it < 4096 && (t.P > T + t.Tpp); ++it) { T += t.Tpp; sdm = -sdm; }
#endif
#if S5_FAST
    // The chain photon_momentum -> kerr_metric -> tetrad_azimuthal -> bl2on -> on2bl -> normalise -> polarization_constant of
    // the strict branch below, WRITTEN OUT for its one use: an emitter in the equatorial plane (m = 0) on a Keplerian orbit.
    // Called with the literal m = 0 the generic routines still multiply through their sixteen-entry tetrad and the cos(theta)
    // terms (x * 0.0 is not foldable in IEEE arithmetic), ~300 issue slots with ten divisions and square roots; here ~110.
    // The Walker-Penrose constant is bilinear in (k, f) and f is linear in k, and only the DIRECTION of (wp0, wp1) enters
    // chi (the angle of ref src/sim5polarization.c:272-285 is an atan2 of two numbers with a common denominator), so every
    // positive common factor is dropped: 1/r^2 of the momentum, u^t and 1/sqrt(Delta) of the tetrad, 1/sqrt(g11), the
    // normalisation of f, the factor r of kappa.  Same numbers up to rounding; NaN where the generic chain gives NaN (k^theta
    // imaginary, orbit not time-like).
    {
        const double r = t.r, r2 = r * r;
        const double ak = t_a, am = p.a;                   // the geodesic's spin (clamped at 1e-4) and the job's, as below
        // r^2 k^mu  (ref src/sim5kerr.c:1151-1213 at m = 0)
        const double Dk = r2 - 2. * r + ak * ak;
        const double Tk = r2 + ak * ak - ak * t_l;
        double Rk = Tk * Tk - Dk * (sq(t_l - ak) + t_q);
        double Mk = t_q;
        if ((Mk < 0.0) && (-Mk < 1e-8)) Mk = 0.0;
        if ((Rk < 0.0) && (-Rk < 1e-8)) Rk = 0.0;
        const double iDk = mrcp(Dk);
        const double TD = Tk * iDk;
        const double K0 = (r2 + ak * ak) * TD - ak * (ak - t_l);
        const double K3 = ak * TD - (ak - t_l);
        const double K1 = (t.dP > 0.0) ? -msqrt(Rk) : msqrt(Rk);
        const double K2 = (sdm < 0.0) ? -msqrt(Mk) : msqrt(Mk);           // NaN for Mk < 0, as there
        // metric of the equatorial plane (ref :75-101 at m = 0)
        const double ir = mrcp(r);
        const double g00 = -1. + 2. * ir, g03 = -2. * am * ir, g33 = r2 + am * am + 2. * (am * am) * ir;
        const double g11 = r2 * ((ak == am) ? iDk : mrcp(r2 - 2. * r + am * am));
        // Keplerian orbit: u ~ (1, 0, 0, Omega); c1 = u.d_t / u^t, c2 = u.d_phi / u^t; the azimuthal leg of the tetrad is
        // sign(c1) (-c2, 0, 0, c1) / (u^t sqrt(Delta))  (ref :766-814)
        const double Om = mrcp(am + r * sqrt_pos(r));
        const double c1 = g00 + Om * g03, c2 = g03 + Om * g33;
        const double sg = (c1 >= 0.0) ? 1.0 : -1.0;
        const bool timelike = (c1 + Om * c2) < 0.0;
        // local components of k along the radial and azimuthal legs, f = n3 e_r - n1 e_phi  (ref: the recipe of SURVEY 3.4)
        const double N1 = K1 * g11;
        const double N3 = sg * (c1 * (K3 * g33 + K0 * g03) - c2 * (K0 * g00 + K3 * g03));
        const double F0 = sg * (N1 * c2), F3 = -sg * (N1 * c1);           // F1 = N3, F2 = 0
        // kappa / r  (ref src/sim5polarization.c:145-168 at m = 0)
        const double A1 = (K0 * N3 - K1 * F0) + am * (K1 * F3 - K3 * N3);
        const double A2 = K2 * (am * F0 - (r2 + am * am) * F3);
        wp[0] = timelike ? A1 : NAN;
        wp[1] = timelike ? -A2 : NAN;
    }
#else
    double k[4], n[4], floc[4], fv[4];
    photon_momentum(t.a, t.r, 0.0, t.l, t.q, (t.dP > 0.0 ? -1. : +1.), sdm, k);
    Metric mt;
    kerr_metric(p.a, t.r, 0.0, mt);
    Tetrad tt;
    tetrad_azimuthal(mt, omega_kepler(t.r, p.a), tt);
    bl2on(k, n, tt);
    floc[0] = 0.0; floc[1] = n[3]; floc[2] = 0.0; floc[3] = -n[1];
    on2bl(floc, fv, tt);
    normalize_to(fv, 1.0, mt);
    polarization_constant(k, fv, mt, wp);
#endif
#if S5_FAST
    // chi = atan2(Y, X) with X, Y sharing the positive denominator S^2 + T^2 (ref src/sim5polarization.c:279-283):
    // the angle needs neither division, cos 2chi = (X^2 - Y^2)/(X^2 + Y^2) and sin 2chi = 2XY/(X^2 + Y^2) need no
    // angle at all; atan2 is evaluated only when the caller asked for the chi plane.
    {
        const double S = -alpha - p.a * p.sin_i, Tb = +beta;
        const double Xn = -S * wp[1] - Tb * wp[0], Yn = -S * wp[0] + Tb * wp[1];
        const double rn = mrcp(Xn * Xn + Yn * Yn);
        Q = p.pol_degree * I * ((Xn - Yn) * (Xn + Yn) * rn);
        U = p.pol_degree * I * (2.0 * Xn * Yn * rn);
        if (AUX && p.chi) chi = matan2(Yn, Xn);
    }
#else
    chi = polarization_angle_rotation(p.a, p.sin_i, alpha, beta, wp);
    double s2c, c2c;
    msincos(2.0 * chi, s2c, c2c);
    Q = p.pol_degree * I * c2c;
    U = p.pol_degree * I * s2c;
#endif
}

template <bool AUX, class PRM>
S5_DEV void store_polarized(const PRM& p, size_t o, const ThinRay& t, double I, double Q, double U, double chi)
{
    const size_t npix = (size_t)p.nrows * (size_t)p.nx;
    p.stokes[o] = I;
    p.stokes[npix + o] = Q;
    p.stokes[2 * npix + o] = U;
    if (AUX) {
        if (p.chi) p.chi[o] = chi;
        if (p.cls) p.cls[o] = (uint8_t)t.cls;
        if (p.gtype) p.gtype[o] = (int8_t)t.gtype;
        if (p.r) p.r[o] = t.r;
        if (p.g) p.g[o] = t.g;
        if (p.flux) p.flux[o] = t.flux;
    }
}

template <bool AUX>
__global__ __launch_bounds__(256, 2)
void disk_image_polarized_kernel(ImageParams p)
{
    const int lane_x = threadIdx.x & 15;
    const int lane_y = threadIdx.x >> 4;
    const int ix = blockIdx.x * 16 + lane_x;
    const int lr = blockIdx.y * 16 + lane_y;                     // packed (local) row
    if (ix >= p.nx || lr >= p.nrows) return;
    const int iy = image_row(p, lr);
    const double alpha = pixel_alpha(p, ix), beta = pixel_beta(p, iy);       // as the unpolarized kernel
    ThinRay t;
    trace_thin_disk<true>(p, alpha, beta, t, iy);
    double I, Q, U, chi;
    polarize_ray<AUX>(p, alpha, beta, t, I, Q, U, chi);
    store_polarized<AUX>(p, (size_t)lr * (size_t)p.nx + (size_t)ix, t, I, Q, U, chi);
}

#if S5_FAST
// symmetric row sets (k_disk_image.hip: disk_image_mirror_kernel): the pixel and its mirror image in beta share the geodesic;
// the polarization chain runs for each of the two, as a loop of two passes over ONE inlined copy
#define S5_LB_POLAR_MIRROR 4                // FOUR waves per SIMD since round 4: 120 VGPRs, no scratch -- the polarization chain takes
                                            // nothing of the geodesic but r, the crossing's order and the radial direction (polarize_ray),
                                            // so the trace no longer keeps Tpp, Tip, P, l, q, a, beta of both rays alive (round 3: 148
                                            // VGPRs, three waves; capped at 128 it spilled 20-30 registers, and spills are refused here)
template <bool AUX>
__global__ __launch_bounds__(256, S5_LB_POLAR_MIRROR)
void disk_image_polarized_mirror_kernel(ImageParams p_arg)
{
    // the job read through the constant address space where its values are used (as the job-list kernel of k_disk_image.hip):
    // the argument block is this kernel's only parameter, at the head of the argument segment
    const S5_AS4 ImageParams& p = *(const S5_AS4 ImageParams*)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane_x = threadIdx.x & 15;
    const int lane_y = threadIdx.x >> 4;
    const int ix = blockIdx.x * 16 + lane_x;
    const int lr = (int)(gridDim.y - 1u - blockIdx.y) * 16 + lane_y;    // local row in the upper half; row tiles from the middle of the
                                                                        // image outwards: the expensive rays first (k_disk_image.hip)
    const int half = (p.nrows + 1) / 2;
    if (ix >= p.nx || lr >= half) return;
    const int lr2 = p.nrows - 1 - lr;
    const int iy = image_row_top(p, lr);
    const double alpha = pixel_alpha(p, ix), beta = pixel_beta(p, iy);
    ThinRay t, t2;
    trace_thin_disk_impl<true, true>(p, alpha, beta, t, t2, iy);
#pragma unroll
    for (int member = 0; member < 2; ++member) {
        if (member == 1 && !wave_any(lr2 != lr)) break;
        // a copy of the member's record: the chain is instantiated once
        ThinRay m = t;
        if (member == 1) m = t2;
        const double b = (member == 0) ? beta : -beta;
        double I, Q, U, chi;
        const auto& pp = param_reload(p);                  // (the chain's few parameters and the output pointers: loaded here)
        polarize_ray<AUX>(pp, alpha, b, m, I, Q, U, chi);
        if (member == 0 || lr2 != lr)
            store_polarized<AUX>(pp, (size_t)(member == 0 ? lr : lr2) * (size_t)pp.nx + (size_t)ix, m, I, Q, U, chi);
    }
}
#endif

} // namespace S5NS

#if S5_FAST
int s5_launch_disk_image_polarized_fast(const s5abi::ImageParams& p, hipStream_t stream)
#else
int s5_launch_disk_image_polarized_strict(const s5abi::ImageParams& p, hipStream_t stream)
#endif
{
    using namespace S5NS;
    const bool aux = p.chi || p.cls || p.gtype || p.r || p.g || p.flux;
#if S5_FAST
    if ((p.mirror || (p.stripe_rows == 0 && p.y0 + p.y1 == p.ny)) && p.nrows >= 2) {
        const dim3 grid((p.nx + 15) / 16, ((p.nrows + 1) / 2 + 15) / 16);
        if (aux) hipLaunchKernelGGL(disk_image_polarized_mirror_kernel<true>, grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(disk_image_polarized_mirror_kernel<false>, grid, dim3(256), 0, stream, p);
        return (int)hipGetLastError();
    }
#endif
    const dim3 grid((p.nx + 15) / 16, (p.nrows + 15) / 16);
    if (aux) hipLaunchKernelGGL(disk_image_polarized_kernel<true>, grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(disk_image_polarized_kernel<false>, grid, dim3(256), 0, stream, p);
    return (int)hipGetLastError();
}
