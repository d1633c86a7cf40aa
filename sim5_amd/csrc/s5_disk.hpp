// s5_disk.hpp -- Novikov-Thorne thin disk (Page & Thorne 1974 flux) for the gfx950 kernels.
//
// Restated from the reference (ref: /root/reference/src/sim5disk-nt.c:37-146, 260-266).  The
// reference keeps the model in file-static *float* variables; everything that depends only on
// them is folded on the host (sim5gpu_disk_nt_setup / the image launchers, in double, with the
// host libm, in the reference's operation order) into DiskConsts, which reaches the kernel as
// kernel arguments, i.e. in SGPRs: the profile is a closed form (4 logs per hit), so there is no
// radial table to stage in LDS -- the constants are wave-uniform scalars.
#pragma once
#include "s5_geod.hpp"
#include "kernels.hpp"

namespace S5NS {

using s5abi::DiskConsts;
using s5abi::FT_N;
using s5abi::FT_DEG;

// The closed form, ref :110-146 -- ONE body for both arithmetic variants, the reference's statement sequence in IEEE
// operations (correctly rounded sqrt and divisions, no contraction).  Near the inner edge the four terms cancel: each is
// O(x - x0) while their sum is O((x - x0)(x0 - x_ms)), and the quotients x / x0, (x - x_i) / (x0 - x_i) are rounded to
// doubles next to 1 -- 1.1e-16 of noise per term against a sum of 1e-10: the reference's own value is its rounding
// pattern to 1e-6 and more (2.9e-5 outside the edge: 1.6e-6 between a quotient and a product with the reciprocal, the
// fast variant's form until round 5).  Parity at the bar therefore needs the reference's ROUNDINGS, i.e. its operations
// on its bits: x = sqrt(r) correctly rounded, true divisions, the sums in its order.  The logarithm itself is not the
// problem: it is taken of 1 + delta, and an error of an ulp of log(1 + delta) ~ delta is 1e-16 RELATIVE to the term.
// Only lanes the flux table does not serve come here in the fast variant (within 2e-4 of the edge in x, beyond x = 16,
// spins without a table): rare, so the IEEE sequences cost nothing measurable.
struct ClosedFormConsts { double a, x0, x1, x2, x3, p1, p2, p3, d1, d2, d3, mdot, mass; };

S5_DEV double disk_flux_closed_form_ieee(const ClosedFormConsts& c, double r)
{
    S5_FPC_PRAGMA_OFF
    const double a = c.a;
    const double x = __builtin_sqrt(r);                         // IEEE in both variants (never the refined-seed root)
    const double f0 = x - c.x0 - 1.5 * a * log(x / c.x0);
    const double f1 = c.p1 * log((x - c.x1) / c.d1);
    const double f2 = c.p2 * log((x - c.x2) / c.d2);
    const double f3 = c.p3 * log((x - c.x3) / c.d3);
    const double F = 1. / (4. * M_PI * r) * 1.5 / (x * x * (x * x * x - 3. * x + 2. * a)) * (f0 - f1 - f2 - f3);
    return 9.1721376255e+28 * F * c.mdot / c.mass;
}

S5_DEV double disk_flux_closed_form(const DiskConsts& d, double r)
{
    const ClosedFormConsts c = { d.a, d.x0, d.x1, d.x2, d.x3, d.p1, d.p2, d.p3, d.d1, d.d2, d.d3, d.mdot, d.mass };
    return disk_flux_closed_form_ieee(c, r);
}

#if S5_FAST
S5_DEV double disk_flux_x(const DiskConsts& d, double r, double x, double rx);
#endif

S5_DEV double disk_flux(const DiskConsts& d, double r)                 // ref :110-146
{
    if (r <= d.rms) return 0.0;
#if S5_FAST
    double x, rx;
    sqrt_rsqrt_pos(r, x, rx);                         // r > rms > 0
    return disk_flux_x(d, r, x, rx);
#else
    return disk_flux_closed_form(d, r);
#endif
}

#if S5_FAST
// The radial profile from the host's table (kernels.hpp, capi_core.hip: F / (scale (x - x0)) as polynomials of
// degree FT_DEG on FT_N equal intervals of w = x0 / x): one reciprocal square root, eight loads of one 64-byte
// row, seven FMAs -- instead of four logarithms and a division.  Lanes the table must not serve -- within 2e-4
// of the inner edge in x, where the reference's own double evaluation is rounding noise that parity reproduces,
// and beyond x = 16 -- are reported through `closed_form` and take disk_flux_closed_form(d, r, x).
template <class DISK>
S5_DEV double disk_flux_table(const DISK& d, double r, double x, double rx, bool& closed_form)
{
    S5_FPC_GFLUX
    closed_form = false;
    if (r <= d.rms) return 0.0;
    const double t = x - d.x0;
    const double w = d.x0 * rx;
    const bool tab = (d.ftab != nullptr) && (t > 2e-4) && (w > d.ft_wmin);
    closed_form = !tab;
    double F = 0.0;
    if (tab) {
        const double u = (w - d.ft_wmin) * d.ft_inv_dw;
        int i = (int)u;
        i = i < FT_N - 1 ? i : FT_N - 1;
        const double tau = 2.0 * (u - (double)i) - 1.0;
        const double* c = d.ftab + (size_t)i * (FT_DEG + 1);
        double acc = c[FT_DEG];
#pragma unroll
        for (int k = FT_DEG - 1; k >= 0; --k) acc = __builtin_fma(acc, tau, c[k]);
        F = d.scale * (t * acc);
    }
    return F;
}

// the closed form with its constants read from the disk model's device block (DiskConsts::cold: a, x0, x1, x2, x3, p1,
// p2, p3, d1, d2, d3, mdot, mass -- the members of ClosedFormConsts in order): same operations, same values
S5_DEV double disk_flux_closed_form_mem(const double* __restrict__ c, double r)
{
    const ClosedFormConsts k = { c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], c[10], c[11], c[12] };
    return disk_flux_closed_form_ieee(k, r);
}

// the same with x = sqrt(r) and 1/x supplied by the caller (the g-factor of the same point needs sqrt(r) too); the lanes the
// table does not serve take the closed form here, their wave with them.  (The image kernels call disk_flux_table and
// do that AFTER their per-ray loop: inlined into the loop, the cold closed form cost its hot path 7 % of its VALU
// instructions in moves around a body that 0.09 % of the waves run -- measured; out of line as a function it needs a stack.)
S5_DEV double disk_flux_x(const DiskConsts& d, double r, double x, double rx)
{
    bool cf;
    double F = disk_flux_table(d, r, x, rx, cf);
    if (wave_any(cf)) {
        if (cf) F = disk_flux_closed_form(d, r);
    }
    return F;
}
#endif

S5_DEV double disk_ell(const DiskConsts& d, double r)                  // ref :260-266
{
    const double a = d.a;
    r = fmax(d.rms, r);
#if S5_FAST
    const double x = sqrt_pos(r);                                      // (r >= rms > 0; one root for the three)
    return mdiv(r * r - 2. * a * x + a * a, x * r - 2. * x + a);
#else
    return (r * r - 2. * a * sqrt(r) + a * a) / (sqrt(r) * r - 2. * sqrt(r) + a);
#endif
}

// Column density of the two inner zones (ref :204-250), the reference's expressions term by term
// (3.*(x1-a)*(x1-a)/... associates differently from the flux's 3.*sqr(x1-a)/..., so the prefactors are formed here).
S5_DEV double disk_sigma(const DiskConsts& d, double r)
{
    if (r < d.rms) return 0.0;
    const double a = d.a;
    const double x = sqrt(r);
    const double x0 = d.x0, x1 = d.x1, x2 = d.x2, x3 = d.x3;
    const double xA = 1. + (a * a) / (r * r) + 2. * (a * a) / (r * r * r);
    const double xB = 1. + a / (x * x * x);
    const double xC = 1. - 3. / (x * x) + 2. * a / (x * x * x);
    const double xD = 1. - 2. / r + (a * a) / (r * r);
    const double xE = 1. + 4. * (a * a) / (r * r) - 4. * (a * a) / (r * r * r) + 3. * (a * a * a * a) / (r * r * r * r);
    const double f0 = x - x0 - 1.5 * a * log(x / x0);
    const double f1 = 3. * (x1 - a) * (x1 - a) / (x1 * (x1 - x2) * (x1 - x3)) * log((x - x1) / (x0 - x1));
    const double f2 = 3. * (x2 - a) * (x2 - a) / (x2 * (x2 - x1) * (x2 - x3)) * log((x - x2) / (x0 - x2));
    const double f3 = 3. * (x3 - a) * (x3 - a) / (x3 * (x3 - x2) * (x3 - x1)) * log((x - x3) / (x0 - x3));
    const double xL = (1. + a / (x * x * x)) / sqrt(1. - 3. / (x * x) + 2. * a / (x * x * x)) / x * (f0 - f1 - f2 - f3);
    const double xMdot = d.mdot * d.mass * 2.225475942e+18 / 1e17;                  // Mdot_Edd, ref src/sim5const.h:49
    const double r_im = 40. * (pow(d.alpha, 2. / 21.) / pow(d.mass / 3., 2. / 3.) * pow(xMdot, 16. / 20.)) * pow(xA, 20. / 21.) *
                        pow(xB, -36. / 21.) * pow(xD, -8. / 21.) * pow(xE, -10. / 21.) * pow(xL, 16. / 21.);
    if (r < r_im)
        return 20. * (d.mass / 3.) / xMdot / d.alpha * sqrt(r * r * r) * 1. / (xA * xA) * pow(xB, 3.) * sqrt(xC) * xE * 1. / xL;
    return 5e4 * pow(d.mass / 3., -2. / 5.) * pow(xMdot, 3. / 5.) * pow(d.alpha, -4. / 5.) * pow(r, -3. / 5.) *
           pow(xB, -4. / 5.) * sqrt(xC) * pow(xD, -4. / 5.) * pow(xL, 3. / 5.);
}

// integrand of the luminosity integral over log r: 2 pi r 2 (-U_t) F(r) r  (ref :167-179)
S5_DEV double disk_lumi_integrand(const DiskConsts& d, double log_r)
{
    const double a = d.a;
    const double r = exp(log_r);
    const double gtt = -1. + 2. / r;
    const double gtf = -2. * a / r;
    const double gff = r * r + d.a2f + 2. * d.a2f / r;
    const double Omega = 1. / (a + pow(r, 1.5));
    const double U_t = sqrt(-1.0 / (gtt + 2. * Omega * gtf + (Omega * Omega) * gff)) * (gtt + Omega * gtf);
    const double F = disk_flux(d, r);
    return 2. * M_PI * r * 2.0 * (-U_t) * F * r;
}

} // namespace S5NS
