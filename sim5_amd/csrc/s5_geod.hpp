// s5_geod.hpp -- null geodesics of the Kerr metric through elliptic integrals, gfx950 device code.
//
// Restated from the reference (ref: /root/reference/src/sim5kerr-geod.c, src/sim5polyroots.c:278
// for the root ordering, src/sim5math.c:50 for the slack clamp).  What is different from the
// reference's shape, on purpose:
//  * the four roots of R(r) are produced already ordered from the two square-root discriminants
//    (no complex type, no generic sort): a pair is real iff its discriminant is >= 0, which is the
//    C99 csqrt(x + 0i) behaviour the reference's `cimag(z) == 0` test relies on;
//  * K(mm) and cn^-1(cos_i/sqrt(m2p) | mm) are evaluated once per ray in init and cached next to
//    the geodesic record: the reference re-evaluates the same expressions in
//    geodesic_find_midplane_crossing (twice for two crossings).  Same inputs, same routine, same
//    bits -- 3 Carlson R_F evaluations per ray instead of 6.3;
//  * the record lives in registers; fields nobody reads are removed by the compiler per kernel.
//  * the polar roots m2m, m2p follow the reference's HOST branch (:1125-1131: x87 long double, each result rounded to 64
//    bits and again to 53) through an integer restatement of those operations (s5_x87.hpp): the strict variant for every
//    ray, the fast variant for the rays whose range tests (:1140, :1153) that rounding decides (l = 0: m2p = 1 in real
//    arithmetic; beta = 0 -> 1e-6: the observer on the polar turning point).
#pragma once
#include "s5_kerr.hpp"
#include "s5_x87.hpp"

namespace S5NS {

enum : int { T_RR = 40, T_RR_DBL = 41, T_RR_BH = 42, T_RC = 2, T_CC = 0 };
enum : int {
    GD_OK = 0, GD_E_UNKNOWN = 3, GD_E_RR_DOUBLE = 4, GD_E_Q_RANGE = 7, GD_E_MUPLUS = 8,
    GD_E_MU0 = 9, GD_E_MM = 10, GD_E_INCL = 11, GD_E_SPIN = 12
};

// byte-identical to sim5gpu_geodesic / the reference's struct geodesic (240 B)
struct Geod {
    double a, alpha, beta, incl, cos_i;
    double l, q;
    double r1[2], r2[2], r3[2], r4[2];
    int nrr, type;
    double m2p, m2m, mm, mK;
    double rp, dmdp_inf;
    double Rpc, Tpp, Tip;
    double k[4];
    double p;
};
static_assert(sizeof(Geod) == 240, "geodesic record must keep the SIM5 layout");

// values the image kernels reuse between init and the crossing search
struct GeodCache {
    double K;        // K(mm)
    double icn_i;    // cn^-1(cos_i / sqrt(m2p) | mm)
    double u_i;      // cos_i / sqrt(m2p)
    bool   valid;
};

S5_DEV double pol_integral(const Geod& g, double x) { return g.mK * inv_cn(mdiv(x, msqrt(g.m2p)), g.mm); }  // ref :29
S5_DEV double pol_inverse(const Geod& g, double T) { return msqrt(g.m2p) * jac_cn(mdiv(T, g.mK), g.mm); }   // ref :30

// ---------------------------------------------------------------------------------------
// roots of R(r), geodesic class, pericentre and the radial integral to infinity  (ref :986-1104)
// ---------------------------------------------------------------------------------------
S5_DEV bool radial_roots(Geod& g, double r0, int& err)
{
    const double a = g.a, l = g.l, q = g.q;
    const double a2 = a * a, l2 = l * l;
    double A;
    const double C = sq(a - l) + q;
    const double D = 2. / 3. * (q + l2 - a2);
    const double E = 9. / 4. * sq(D) - 12. * a2 * q;
    const double F = -27. / 4. * (D * D * D) - 108. * a2 * q * D + 108. * sq(C);
    const double X = sq(F) - 4. * (E * E * E);
    if (X >= 0) {
        const double sX = msqrt(X);
        A = (F > sX ? +1 : -1) * 1. / 3. * mcbrt(fabs(F - sX) / 2.) +
            (F > -sX ? +1 : -1) * 1. / 3. * mcbrt(fabs(F + sX) / 2.);
    } else {
        const double sX = S5_DIVC(msqrt(-X), 54.);
        const double F54 = S5_DIVC(F, 54.);
        const double Z = msqrt(sq(F54) + sq(sX));
        const double z = matan2(sX, F54);
        A = mcbrt(Z) * 2. * mcos(S5_DIVC(z, 3.));
    }
    const double B = msqrt(A + D);
    const double CB = mdiv(4. * C, B);
    const double w_hi = -A + 2. * D - CB;             // discriminant of the pair around +B/2
    const double w_lo = -A + 2. * D + CB;             // discriminant of the pair around -B/2
    const bool hi_real = (w_hi >= 0.0), lo_real = (w_lo >= 0.0);
    // csqrt(w + 0i): (sqrt(w), 0) for w >= 0, (0, sqrt(-w)) for w < 0; halves as in `.5*csqrt()`
    const double h_hi = .5 * msqrt(fabs(w_hi)), h_lo = .5 * msqrt(fabs(w_lo));
    const double c_hi = +B / 2., c_lo = -B / 2.;

    g.nrr = (hi_real ? 2 : 0) + (lo_real ? 2 : 0);
    {
        // The four cases -- both pairs real (four real roots, descending: within a pair the "+" root is the larger one, the two
        // ordered pairs merged), the upper pair real, the lower pair real, none -- as SELECTS over values every lane forms: written
        // as four branches that store into the struct, the compiler turned the imaginary parts into an indexed store to a
        // private array (32 bytes of scratch in every kernel that inlines this routine).  Same values, same signs of zero.
        const double p0 = c_hi + h_hi, p1 = c_hi - h_hi, p2 = c_lo + h_lo, p3 = c_lo - h_lo;
        const double s0 = fmax(p0, p2), t0 = fmin(p0, p2);
        const double s3 = fmin(p1, p3), t3 = fmax(p1, p3);
        const double s1 = fmax(t0, t3), s2 = fmin(t0, t3);
        const bool both = hi_real && lo_real, none = !hi_real && !lo_real;
        g.r1[0] = both ? s0 : hi_real ? p0 : lo_real ? p2 : c_hi;
        g.r2[0] = both ? s1 : hi_real ? p1 : lo_real ? p3 : c_hi;
        g.r3[0] = both ? s2 : lo_real ? c_hi : c_lo;
        g.r4[0] = both ? s3 : lo_real ? c_hi : c_lo;
        g.r1[1] = none ? +h_hi : 0.0;
        g.r2[1] = none ? -h_hi : 0.0;
        g.r3[1] = both ? 0.0 : lo_real ? +h_hi : +h_lo;
        g.r4[1] = both ? 0.0 : lo_real ? -h_hi : -h_lo;
    }

    if (g.nrr == 4) {
        g.type = T_RR;
        if ((r0 < g.r3[0]) || ((r0 > g.r2[0]) && (r0 < g.r1[0]))) { err = GD_E_UNKNOWN; return false; }
        if (fabs(g.r1[0] - g.r2[0]) < 1e-8) { g.type = T_RR_DBL; err = GD_E_RR_DOUBLE; return false; }
        if ((r0 >= g.r3[0]) && (r0 <= g.r2[0])) g.type = T_RR_BH;
    } else if (g.nrr == 2) {
        g.type = T_RC;
    } else {
        g.type = T_CC;
    }

    if (g.type == T_RR || g.type == T_RR_BH) {
        const double r1 = g.r1[0], r2 = g.r2[0], r3 = g.r3[0], r4 = g.r4[0];
        const double mm = mdiv((r2 - r3) * (r1 - r4), (r2 - r4) * (r1 - r3));
        const double pre = mdiv(2., msqrt((r1 - r3) * (r2 - r4)));
        if (g.type == T_RR) {
            g.rp = r1;
            g.Rpc = pre * inv_sn(msqrt(mdiv(r2 - r4, r1 - r4)), mm);
        } else {
            g.rp = r2;
            g.Rpc = pre * ell_K(mm);
        }
    } else if (g.type == T_RC) {
        const double r1 = g.r1[0], r2 = g.r2[0], u = g.r3[0], v = g.r3[1];
        const double Aq = msqrt(sq(r1 - u) + sq(v));
        const double Bq = msqrt(sq(r2 - u) + sq(v));
        const double mm = mdiv(sq(Aq + Bq) - sq(r1 - r2), 4. * Aq * Bq);
        g.rp = r1;
        g.Rpc = mdiv(1., msqrt(Aq * Bq)) * inv_cn(mdiv(Aq - Bq, Aq + Bq), mm);
    } else {
        const double b1 = g.r1[0], b2 = g.r3[0], a1 = g.r1[1], a2c = g.r3[1];
        const double Aq = msqrt(sq(b1 - b2) + sq(a1 + a2c));
        const double Bq = msqrt(sq(b1 - b2) + sq(a1 - a2c));
        const double g1 = msqrt(mdiv(4. * sq(a1) - sq(Aq - Bq), sq(Aq + Bq) - 4. * sq(a1)));
        const double mm = mdiv(4. * Aq * Bq, sq(Aq + Bq));
        g.rp = b1 - a1 * g1;
        g.Rpc = mdiv(2., Aq + Bq) * inv_tn(mdiv(-1., g1), mm);
    }
    return true;
}

// ---------------------------------------------------------------------------------------
// roots of the polar potential  (ref :1110-1184, device branch)
// ---------------------------------------------------------------------------------------
// m2m = X / 2a^2, m2p = 2q / X, X = sqrt(qla^2 + 4 q a^2) + qla (ref :1125-1131), with the reference's host roundings
S5_DEV void polar_m2_host_rounding(double q, double l2, double a2, double& m2m, double& m2p)
{
    const double qla = q + l2 - a2;
    const double c4 = 4. * q * a2;
    if (s5x87::polar_roots_x87(qla, c4, a2 + a2, q + q, m2m, m2p)) return;
    const double X = sqrt(sq(qla) + c4) + qla;             // (operands outside the emulation's range: NaN / infinity as in double)
    m2m = X / (a2 + a2);
    m2p = (q + q) / X;
}

// The two range tests on m2p that rounding can decide: m2p against 1 (ref :1140) and |m| against sqrt(m2p) (ref :1153).
// How far rounding moves m2p depends on the sum X = sqrt(qla^2 + 4 q a^2) + qla: with qla < 0 and |q| << a^2 it cancels, and X --
// with it m2p = 2q / X -- carries a relative error of ~ulp |qla| / |X| (found by the campaign of round 6: alpha = 0 and
// beta^2 ~ a^2 cos^2 i gave the reference m2p = 1 + 1.4e-12 -- rejected -- and this arithmetic 1 - 3e-12).  The margin is
// 1e-12 scaled by that factor; l = 0 (m2p = 1 in real arithmetic) is marginal whatever came out.
S5_DEV bool polar_tests_marginal(double m2p, double s_m2p, double m, double l2, double qla, double X)
{
    const double wide = 1e-12 * fmax(fabs(X), fabs(qla));           // 1e-12 max(1, |qla| / |X|), times |X|
    return (l2 == 0.0) || (fabs(m2p - 1.0) * fabs(X) < wide) || (fabs(s_m2p - fabs(m)) * fabs(X) < wide);
}

// Carter's constant of a ray from infinity (ref :77; the caller's spin, not the clamped one), in the reference's roundings
S5_DEV double constant_q(double beta, double cos_i, double alpha, double a_in)
{
    const double b2 = rounded_product(sq(beta)), al2 = rounded_product(sq(alpha)), as2 = rounded_product(sq(a_in));
    return b2 + rounded_product(rounded_product(sq(cos_i)) * (al2 - as2));
}

// the range tests of the polar roots alone (ref :1140-1176), in the reference's order
S5_DEV int polar_range_error(double q, double a2, double m2m, double m2p, double s_m2p, double m)
{
    if ((m2p <= 0.0) || (m2p >= 1.0)) return GD_E_MUPLUS;
    if (q > 0.0) {
        const double mm = m2p / (m2p + m2m);
        if ((mm < 0.0) || (mm >= 1.0)) return GD_E_MM;
        if (fabs(m) > s_m2p) return GD_E_MU0;
    } else if (q < 0.0) {
        const double mm = (m2p + m2m) / m2p;
        if ((mm < 0.0) || (mm >= 1.0)) return GD_E_MM;
        if ((fabs(m) > s_m2p) || (fabs(m) < sqrt(-m2m))) return GD_E_MU0;
    } else return GD_E_Q_RANGE;
    return GD_OK;
}

S5_DEV bool polar_roots(Geod& g, double m, int& err)
{
    const double a = g.a, l = g.l, q = g.q;
    const double a2 = rounded_product(a * a), l2 = rounded_product(l * l);
#if S5_FAST
    const double qla = q + l2 - a2;
    const double X = msqrt(sq(qla) + 4. * q * a2) + qla;
    g.m2m = mdiv(X, a2 + a2);
    g.m2p = mdiv(q + q, X);
    double s_m2p = msqrt(g.m2p);
    if (polar_tests_marginal(g.m2p, s_m2p, m, l2, qla, X)) {  // rounding decides: the reference's roundings, an IEEE root
        polar_m2_host_rounding(q, l2, a2, g.m2m, g.m2p);
        s_m2p = sqrt(g.m2p);
    }
#else
    polar_m2_host_rounding(q, l2, a2, g.m2m, g.m2p);
    const double s_m2p = msqrt(g.m2p);
#endif
    if ((g.m2p <= 0.0) || (g.m2p >= 1.0)) { err = GD_E_MUPLUS; return false; }
    if (q > 0.0) {
        g.mm = mdiv(g.m2p, g.m2p + g.m2m);
        if ((g.mm < 0.0) || (g.mm >= 1.0)) { err = GD_E_MM; return false; }
        if (fabs(m) > s_m2p) { err = GD_E_MU0; return false; }
        g.mK = mdiv(1., msqrt(a2 * (g.m2p + g.m2m)));
    } else if (q < 0.0) {
        g.mm = mdiv(g.m2p + g.m2m, g.m2p);
        if ((g.mm < 0.0) || (g.mm >= 1.0)) { err = GD_E_MM; return false; }
        if ((fabs(m) > s_m2p) || (fabs(m) < msqrt(-g.m2m))) { err = GD_E_MU0; return false; }
        g.mK = mdiv(1., msqrt(a2 * g.m2p));
    } else {
        err = GD_E_Q_RANGE;
        return false;
    }
    return true;
}

// ---------------------------------------------------------------------------------------
// geodesic from impact parameters at infinity  (ref :42-100).  sin_i / cos_i are sin/cos of the
// inclination evaluated by the caller (the image kernels take them from the host's libm so that
// they are the very numbers the reference uses; the batch kernel evaluates them per ray).
// ---------------------------------------------------------------------------------------
S5_DEV bool init_inf(double incl, double sin_i, double cos_i, double a, double alpha, double beta,
                     Geod& g, int& err, GeodCache& cache)
{
    cache.valid = false;
    if ((a < 0.0) || (a > 1. - 1e-6)) { err = GD_E_SPIN; return false; }
    if ((incl <= 0.0) || (incl >= 1.57079632679)) { err = GD_E_INCL; return false; }
    if (beta == 0.0) beta = +1e-6;

    g.a = fmax(1e-4, a);
    g.incl = incl;
    g.cos_i = cos_i;
    g.alpha = alpha;
    g.beta = beta;
    g.l = -alpha * sin_i;
    g.q = constant_q(beta, cos_i, alpha, a);                // caller's a, not the clamped one
    if (g.q == 0.0) { err = GD_E_Q_RANGE; return false; }

    if (!radial_roots(g, DBL_MAX, err)) return false;
    if (!polar_roots(g, g.cos_i, err)) return false;

    // Tpp = 2 mK cn^-1(0|mm) = 2 mK K(mm);  Tip = mK cn^-1(cos_i/sqrt(m2p)|mm)
    cache.K = ell_K(g.mm);                                   // cn^-1(0|mm) with mm in [0,1)
    cache.u_i = mdiv(g.cos_i, msqrt(g.m2p));
    cache.icn_i = inv_cn(cache.u_i, g.mm);
    cache.valid = true;
    g.Tpp = 2. * (g.mK * cache.K);
    g.Tip = g.mK * cache.icn_i;
    err = GD_OK;
    return true;
}

// position integral from infinity to radius r  (ref :179-263)
S5_DEV double P_int(const Geod& g, double r, int ppc)
{
    if (r == g.rp) return g.Rpc;
    if (g.type == T_RR || g.type == T_RR_BH) {
        const double r1 = g.r1[0], r2 = g.r2[0], r3 = g.r3[0], r4 = g.r4[0];
        const double mm = mdiv((r2 - r3) * (r1 - r4), (r2 - r4) * (r1 - r3));
        const double z = (g.type == T_RR) ? msqrt(mdiv((r2 - r4) * (r - r1), (r1 - r4) * (r - r2)))
                                          : msqrt(mdiv(mdiv(r1 - r3, r2 - r3) * (r2 - r), r1 - r));
        const double R = mdiv(2., msqrt((r1 - r3) * (r2 - r4))) * inv_sn(z, mm);
        return (ppc) ? g.Rpc + R : g.Rpc - R;
    }
    if (g.type == T_RC) {
        const double r1 = g.r1[0], r2 = g.r2[0], u = g.r3[0], v = g.r3[1];
        const double A = msqrt(sq(r1 - u) + sq(v));
        const double B = msqrt(sq(r2 - u) + sq(v));
        const double mm = mdiv(sq(A + B) - sq(r1 - r2), 4. * A * B);
        const double R = mdiv(1., msqrt(A * B)) *
                         inv_cn(mdiv((A - B) * r + r1 * B - r2 * A, (A + B) * r - r1 * B - r2 * A), mm);
        return g.Rpc - R;
    }
    if (g.type == T_CC) {
        const double b1 = g.r1[0], b2 = g.r3[0], a1 = g.r1[1], a2c = g.r3[1];
        const double A = msqrt(sq(b1 - b2) + sq(a1 + a2c));
        const double B = msqrt(sq(b1 - b2) + sq(a1 - a2c));
        const double g1 = msqrt(mdiv(4. * sq(a1) - sq(A - B), sq(A + B) - 4. * sq(a1)));
        const double mm = mdiv(4. * A * B, sq(A + B));
        const double R = mdiv(2., A + B) * inv_tn(mdiv(r - b1 + a1 * g1, a1 + b1 * g1 - g1 * r), mm);
        return g.Rpc - R;
    }
    return NAN;
}

// r(P)  (ref :291-357)
S5_DEV double position_rad(const Geod& g, double P)
{
    if ((P <= 0.0) || (P >= 2. * g.Rpc)) return NAN;
    if (P == g.Rpc) return g.rp;
    if (g.type == T_RR) {
        const double r1 = g.r1[0], r2 = g.r2[0], r3 = g.r3[0], r4 = g.r4[0];
        const double m4 = mdiv((r2 - r3) * (r1 - r4), (r2 - r4) * (r1 - r3));
        const double x4 = 0.5 * fabs(P - g.Rpc) * msqrt((r2 - r4) * (r1 - r3));
        const double sn = jac_sn(x4, m4);
        const double sn2 = sn * sn;
        return mdiv(r1 * (r2 - r4) - r2 * (r1 - r4) * sn2, r2 - r4 - (r1 - r4) * sn2);
    }
    if (g.type == T_RC) {
        if (P > g.Rpc) return NAN;
        const double r1 = g.r1[0], r2 = g.r2[0], u = g.r3[0], v = g.r3[1];
        const double A = msqrt(sq(r1 - u) + sq(v));
        const double B = msqrt(sq(r2 - u) + sq(v));
        const double m2 = mdiv(sq(A + B) - sq(r1 - r2), 4. * A * B);
        const double cn = jac_cn(msqrt(A * B) * (g.Rpc - P), m2);
        return mdiv(r2 * A - r1 * B - (r2 * A + r1 * B) * cn, (A - B) - (A + B) * cn);
    }
    return NAN;
}

// sign of d(cos theta)/dP at P and the start T of the polar half-period containing P (ref :737-781)
S5_DEV double polar_phase(const Geod& g, double P, double& T)
{
    double sdm = (g.beta >= 0.0) ? +1.0 : -1.0;
    T = (sdm > 0.0) ? -(g.Tpp - g.Tip) : -(g.Tip);
    // bounded walk: each turn advances T by Tpp > 0; NaN comparisons stop it at once
    for (int it = 0; it < 4096 && (P > T + g.Tpp); ++it) { T += g.Tpp; sdm = -sdm; }
    return sdm;
}

S5_DEV bool escapes(const Geod& g) { return g.type == T_RR || g.type == T_RC || g.type == T_CC; }

S5_DEV double position_pol(const Geod& g, double P)            // ref :363-407
{
    if (!escapes(g)) return NAN;
    double T;
    const double sdm = polar_phase(g, P, T);
    return -sdm * pol_inverse(g, P - T);
}

S5_DEV double dm_sign(const Geod& g, double P)                 // ref :737-781
{
    if (!escapes(g)) return NAN;
    double T;
    return polar_phase(g, P, T);
}

S5_DEV void momentum(const Geod& g, double P, double r, double m, double k[4])   // ref :787-840
{
    if ((r == 0.0) && (m == 0.0)) {
        r = position_rad(g, P);
        m = position_pol(g, P);
    }
    if (escapes(g)) {
        const double dm = dm_sign(g, P);
        photon_momentum(g.a, r, m, g.l, g.q, (P < g.Rpc ? -1. : +1.), dm, k);
    } else if (g.type == T_RR_DBL || g.type == T_RR_BH) {
        k[0] = k[1] = k[2] = k[3] = NAN;
    }
}

// P at the order-th crossing of the equatorial plane  (ref :846-885).  With a valid cache the two
// elliptic evaluations are reused from init (see header comment).
S5_DEV double midplane_crossing(const Geod& g, int order, const GeodCache& cache)
{
    if (g.q <= 0.0) return NAN;
    double u = mdiv(g.cos_i, msqrt(g.m2p));
    if (u < -1.0 - 1e-4) return NAN;                           // slack clamp, ref src/sim5math.c:50-58
    if (u > +1.0 + 1e-4) return NAN;
    if (u < -1.0) u = -1.0;
    if (u > +1.0) u = +1.0;
    const double K = cache.valid ? cache.K : ell_K(g.mm);
    double pos;
    if (g.beta > 0.0 || g.beta < 0.0) {
        const double icn = (cache.valid && u == cache.u_i) ? cache.icn_i : inv_cn(u, g.mm);
        pos = (g.beta > 0.0) ? g.mK * ((2. * (double)order + 1.) * K + icn)
                             : g.mK * ((2. * (double)order + 1.) * K - icn);
    } else {
        pos = g.mK * ((2. * (double)order + 1.) * K);
    }
    if (pos > 2. * g.Rpc) pos = NAN;
    return pos;
}

// init_src  (ref :106-173)
S5_DEV bool init_src(double a, double r, double m, const double k[4], int ppc, Geod& g, int& err)
{
    double l, q;
    photon_motion_constants(a, r, m, k, l, q);
    g.a = fmax(1e-8, a);
    g.l = l;
    g.q = q;
    g.cos_i = g.alpha = g.beta = NAN;
    if (!radial_roots(g, r, err)) return false;
    if (!polar_roots(g, m, err)) return false;
    if (r > g.rp) {
        const double Tmp = pol_integral(g, m);
        const double Tpp = 2. * pol_integral(g, 0.0);
        double T = P_int(g, r, ppc);
        double sdm = (k[2] < 0.0) ? +1.0 : -1.0;
        T += (sdm > 0.0) ? Tpp - Tmp : Tmp;
        for (int it = 0; it < 4096 && (T > Tpp); ++it) { T -= Tpp; sdm = -sdm; }
        g.cos_i = -sdm * pol_inverse(g, T);
        g.incl = acos(g.cos_i);
        g.alpha = -g.l / sqrt(1.0 - sq(g.cos_i));
        g.beta = -sdm * sqrt(g.q - sq(g.cos_i) * (sq(g.alpha) - sq(g.a)));
    }
    g.Tpp = 2. * pol_integral(g, 0.0);
    g.Tip = pol_integral(g, g.cos_i);
    err = GD_OK;
    return true;
}

// ---------------------------------------------------------------------------------------
// r(P) and mu(P) of ONE geodesic evaluated many times (geodesic_follow walks, surface searches).
// Everything in position_rad / position_pol that does not depend on P is a constant of the ray: the two
// elliptic moduli (hence the AGM rungs of their Landen ladders), the root combinations, sqrt(m2p), the polar
// phase bookkeeping.  GeodTrack forms them once -- with the very expressions of position_rad / position_pol
// above, so each later value is the one those routines return, bit for bit -- and keeps the rungs of the two
// ladders in LDS (LadderLdsAt<STRIDE>: 2 x NST x 16 B per lane; NST = 8 serves every modulus in [0, 1), a shorter
// ladder serves all but the rays within 3e-6 of the critical curve and reports those through deep()).  An evaluation is then one sincos and
// the descent of the ladder instead of ~10 square roots of the climb plus the divisions of the constants.
// ---------------------------------------------------------------------------------------
template <int STRIDE, int NST = LADDER_RUNGS_VALID>
struct GeodTrack {
    LadderLdsAt<STRIDE> lad_r, lad_m;            // NST rungs each; deep() tells whether NST were enough
    LadderState st_r, st_m;
    // radial part
    int type;
    double Rpc, rp;
    double fac;                  // RR: sqrt((r2-r4)(r1-r3));  RC: sqrt(A B)
    double c0, c1, c2, c3;       // RR: r1 (r2-r4), r2 (r1-r4), r2-r4, r1-r4;  RC: r2 A - r1 B, r2 A + r1 B, A - B, A + B
    // polar part
    bool escapes_;
    double sq_m2p, mK, Tpp, Tip, beta;

    S5_DEV void build(const Geod& g, double* lds_lane)
    {
        lad_r.base = lds_lane;
        lad_m.base = lds_lane + 2 * NST * STRIDE;
        type = g.type; Rpc = g.Rpc; rp = g.rp;
        st_r.degenerate = false; st_r.flipped = false; st_r.c = 0.0; st_r.d = 1.0; st_r.top = 0;
        fac = c0 = c1 = c2 = c3 = 0.0;
        double m_rad = 0.0;
        if (g.type == T_RR) {
            const double r1 = g.r1[0], r2 = g.r2[0], r3 = g.r3[0], r4 = g.r4[0];
            const double m4 = mdiv((r2 - r3) * (r1 - r4), (r2 - r4) * (r1 - r3));
            fac = msqrt((r2 - r4) * (r1 - r3));
            c0 = r1 * (r2 - r4); c1 = r2 * (r1 - r4); c2 = r2 - r4; c3 = r1 - r4;
            ladder_climb<LadderLdsAt<STRIDE>, NST>(lad_r, m4, st_r);
            m_rad = m4;
        } else if (g.type == T_RC) {
            const double r1 = g.r1[0], r2 = g.r2[0], u = g.r3[0], v = g.r3[1];
            const double A = msqrt(sq(r1 - u) + sq(v));
            const double B = msqrt(sq(r2 - u) + sq(v));
            const double m2 = mdiv(sq(A + B) - sq(r1 - r2), 4. * A * B);
            fac = msqrt(A * B);
            c0 = r2 * A - r1 * B; c1 = r2 * A + r1 * B; c2 = A - B; c3 = A + B;
            ladder_climb<LadderLdsAt<STRIDE>, NST>(lad_r, m2, st_r);
            m_rad = m2;
        }
        escapes_ = escapes(g);
        sq_m2p = msqrt(g.m2p); mK = g.mK; Tpp = g.Tpp; Tip = g.Tip; beta = g.beta;
        ladder_climb<LadderLdsAt<STRIDE>, NST>(lad_m, g.mm, st_m);
#if S5_FAST
        build_along(g, m_rad);
#else
        (void)m_rad;
#endif
    }

    // true if a ladder did not converge within the NST rungs kept (moduli within 3e-6 of 1 need 7 or 8): the caller
    // must not use rad() / pol() then
    S5_DEV bool deep() const { return st_r.incomplete || st_m.incomplete; }

    S5_DEV double rad(double P) const                               // = position_rad(g, P)
    {
        if ((P <= 0.0) || (P >= 2. * Rpc)) return NAN;
        if (P == Rpc) return rp;
        double sn, cn, dn;
        if (type == T_RR) {
            const double x4 = 0.5 * fabs(P - Rpc) * fac;
            ladder_descend<LadderLdsAt<STRIDE>, NST>(lad_r, st_r, x4, sn, cn, dn);
            const double sn2 = sn * sn;
            return mdiv(c0 - c1 * sn2, c2 - c3 * sn2);
        }
        if (type == T_RC) {
            if (P > Rpc) return NAN;
            ladder_descend<LadderLdsAt<STRIDE>, NST>(lad_r, st_r, fac * (Rpc - P), sn, cn, dn);
            return mdiv(c0 - c1 * cn, c2 - c3 * cn);
        }
        return NAN;
    }

    S5_DEV double pol(double P) const                               // = position_pol(g, P)
    {
        if (!escapes_) return NAN;
        double sdm = (beta >= 0.0) ? +1.0 : -1.0;
        double T = (sdm > 0.0) ? -(Tpp - Tip) : -(Tip);
        for (int it = 0; it < 4096 && (P > T + Tpp); ++it) { T += Tpp; sdm = -sdm; }
        double sn, cn, dn;
        ladder_descend<LadderLdsAt<STRIDE>, NST>(lad_m, st_m, mdiv(P - T, mK), sn, cn, dn);
        return -sdm * (sq_m2p * cn);
    }

#if S5_FAST
    // ---- values along a walk ------------------------------------------------------------------------------
    // A walk (geodesic_follow, ref :891-925) evaluates r and mu at a sequence P_1, P_2 = P_1 + dP, ... of nearby
    // points: hundreds of sub-steps of 0.05 sqrt(r) each.  Both arguments of the elliptic functions are linear in P,
    //     u_r = kr (P - Rpc)   (kr = fac / 2 for RR, -fac for RC),     u_p = (P - T) / mK,
    // so a sub-step moves them by the small amounts v = kr dP and dP / mK, and the addition theorems
    //     sn(u+v) = (s c' d' + s' c d) / D,  cn(u+v) = (c c' - s d s' d') / D,  dn(u+v) = (d d' - m s c s' c') / D,
    //     D = 1 - m s^2 s'^2
    // give the new triple from the old one and sn, cn, dn of the SMALL argument, for which the Maclaurin series of
    // sn to v^11 is exact to rounding when max(1, |m|) v^2 <= 5.6e-3 (next term 3.6e-3 v^12) -- about sixty
    // operations instead of sincos and the descent of the ladder; a longer move is made in 2 or 4 equal parts.  `Along` carries the two triples; for the polar
    // one it is -sdm sn and -sdm cn that are kept: the half-period shifts of position_pol (T += Tpp, sdm = -sdm)
    // change the sign of sn, cn and sdm together, so these products are continuous functions of P and the phase
    // bookkeeping drops out.  The caller re-anchors with the full evaluation (anchor(): the very values of rad(), pol())
    // every few dozen sub-steps, which bounds the accumulated rounding (a few ulp per sub-step) near 2e-14, and
    // whenever step_is_small() says the series does not apply.
    struct Along { double Sr, Cr, Dr, Sp, Mp, Dp; };
    double kr, inv_mK, m_r, m_p, dP_small;
    double ser_r[5], ser_p[5];                   // sn(v | m) = v (1 + s0 v^2 + s1 v^4 + s2 v^6 + s3 v^8 + s4 v^10)

    static S5_DEV void sn_series(double m, double s[5])
    {
        s[0] = -(1. + m) * (1. / 6.);
        s[1] = (1. + m * (14. + m)) * (1. / 120.);
        s[2] = -(1. + m * (135. + m * (135. + m))) * (1. / 5040.);
        s[3] = (1. + m * (1228. + m * (5478. + m * (1228. + m)))) * (1. / 362880.);
        s[4] = -(1. + m * (11069. + m * (165826. + m * (165826. + m * (11069. + m))))) * (1. / 39916800.);
    }

    S5_DEV void build_along(const Geod& g, double m_rad)
    {
        kr = (type == T_RR) ? 0.5 * fac : -fac;
        inv_mK = 1.0 / mK;
        m_r = m_rad; m_p = g.mm;
        sn_series(m_r, ser_r); sn_series(m_p, ser_p);
        const double wr = fabs(kr) * sqrt(fmax(1.0, fabs(m_r))), wp = fabs(inv_mK) * sqrt(fmax(1.0, fabs(m_p)));
        dP_small = 0.04 / fmax(wr, wp);
        if (!(type == T_RR || type == T_RC) || !escapes_ || !(dP_small > 0.0) || m_r == 1.0 || m_p == 1.0 ||
            st_r.degenerate || st_m.degenerate) dP_small = 0.0;                   // always anchor
    }

    S5_DEV bool step_is_small(double dP) const { return fabs(dP) <= dP_small; }

    // sqrt(1 - x) for |x| <= 1.6e-3 (next term 2e-2 x^6; with S5_ALONG_LONG 5.6e-3 and 1.6e-2 x^7);
    // TINY: |x| <= 1.6e-5, three terms (next 4e-2 x^4)
    template <bool TINY>
    static S5_DEV double sqrt_one_minus(double x)
    {
        double p;
        if (TINY) p = fma(x, sconst(-0.0625), -0.125);
        else {
            p = fma(x, sconst(-0.02734375), -0.0390625);
            p = fma(x, p, sconst(-0.0625));
            p = fma(x, p, sconst(-0.125));
        }
        p = fma(x, p, -0.5);
        return fma(x, p, 1.0);
    }

    // (S, C, D) = (sn, cn, dn)(u)  ->  (sn, cn, dn)(u + v).  Written with explicit fused operations: this unit is
    // compiled without contraction and the forty operations below are the inner loop of the walk.  TINY: every series
    // two terms shorter, for max(1, |m|) v^2 <= 1.6e-5 (the far part of a walk: most sub-steps).
    template <bool TINY>
    static S5_DEV void add_small(double m, const double ser[5], double v, double& S, double& C, double& D)
    {
        const double v2 = v * v;
        double p;
        if (TINY) p = fma(v2, ser[1], ser[0]);
        else {
            p = fma(v2, ser[3], ser[2]);
            p = fma(v2, p, ser[1]);
            p = fma(v2, p, ser[0]);
        }
        const double sv = v * fma(v2, p, 1.0);
        const double x = sv * sv, mx = m * x;
        const double cv = sqrt_one_minus<TINY>(x), dv = sqrt_one_minus<TINY>(mx);
        // 1 / (1 - y), y = m s^2 s'^2 <= 1.6e-3 (5.6e-3; 1.6e-5): the geometric series to y^5 (y^6; y^3) instead of a reciprocal
        const double y = mx * (S * S);
        double q = y + 1.0;
        if (!TINY) {
            q = fma(y, q, 1.0);
            q = fma(y, q, 1.0);
        }
        q = fma(y, q, 1.0);
        const double inv = fma(y, q, 1.0);
        const double svC = sv * C, svS = sv * S;
        const double Sn = fma(S, cv * dv, svC * D) * inv;
        const double Cn = fma(C, cv, -(svS * D) * dv) * inv;
        const double Dn = fma(D, dv, -(m * svS) * (C * cv)) * inv;
        S = Sn; C = Cn; D = Dn;
    }

    S5_DEV double rad_from(const Along& t, double P) const          // the exits of rad() in its order, as selects
    {
        const bool rr = (type == T_RR);
        const double z = rr ? t.Sr * t.Sr : t.Cr;
        double r = mdiv(fma(-c1, z, c0), fma(-c3, z, c2));
        if (!rr && (P > Rpc)) r = NAN;
        if (P == Rpc) r = rp;
        if ((P <= 0.0) || (P >= 2. * Rpc)) r = NAN;
        return r;
    }

    // full evaluation at P: r, mu as rad(P), pol(P) return them, and the triples for advance()
    S5_DEV void anchor(double P, Along& t, double& r, double& mu) const
    {
        double sn, cn, dn;
        const double u = kr * (P - Rpc);
        ladder_descend<LadderLdsAt<STRIDE>, NST>(lad_r, st_r, fabs(u), sn, cn, dn);
        t.Sr = (u < 0.0) ? -sn : sn; t.Cr = cn; t.Dr = dn;
        r = rad_from(t, P);
        double sdm = (beta >= 0.0) ? +1.0 : -1.0;
        double T = (sdm > 0.0) ? -(Tpp - Tip) : -(Tip);
        for (int it = 0; it < 4096 && (P > T + Tpp); ++it) { T += Tpp; sdm = -sdm; }
        ladder_descend<LadderLdsAt<STRIDE>, NST>(lad_m, st_m, mdiv(P - T, mK), sn, cn, dn);
        t.Sp = -sdm * sn; t.Mp = -sdm * cn; t.Dp = dn;
        mu = escapes_ ? sq_m2p * t.Mp : NAN;
    }

    // the same after a small move dP (step_is_small) that ended at P
    S5_DEV void advance(double dP, double P, Along& t, double& r, double& mu) const
    {
        if (!wave_any(fabs(dP) > 0.1 * dP_small)) {               // v^2 a hundred times below the bound: the short series
            add_small<true>(m_r, ser_r, kr * dP, t.Sr, t.Cr, t.Dr);
            add_small<true>(m_p, ser_p, dP * inv_mK, t.Sp, t.Mp, t.Dp);
        } else {
            add_small<false>(m_r, ser_r, kr * dP, t.Sr, t.Cr, t.Dr);
            add_small<false>(m_p, ser_p, dP * inv_mK, t.Sp, t.Mp, t.Dp);
        }
        r = rad_from(t, P);
        mu = sq_m2p * t.Mp;
    }
#endif

    // one call of geodesic_follow on the tracked geodesic  (ref :891-925)
    S5_DEV void follow(double a, double step, double& P, double& r, double& m, int& status) const
    {
        const double cap = 5e-2;
        for (int it = 0; it < 100000; ++it) {
            const double truestep = mdiv(step, fabs(step)) * fmin(fabs(step), cap * msqrt(r));
            P = P + mdiv(truestep, sq(r) + sq(a * m));
            r = rad(P);
            m = pol(P);
            if (r < 1.01 * r_horizon(a)) { status = 0; return; }
            if ((P < 0.0) || (P > 2. * Rpc)) { status = 0; return; }
            step -= truestep;
            if (!(fabs(step) > 1e-5)) break;
        }
        status = 1;
    }
};

// one call of geodesic_follow  (ref :891-925)
S5_DEV void follow(const Geod& g, double step, double& P, double& r, double& m, int& status)
{
    const double cap = 5e-2;
    for (int it = 0; it < 100000; ++it) {
        const double truestep = mdiv(step, fabs(step)) * fmin(fabs(step), cap * msqrt(r));
        P = P + mdiv(truestep, sq(r) + sq(g.a * m));
        r = position_rad(g, P);
        m = position_pol(g, P);
        if (r < 1.01 * r_horizon(g.a)) { status = 0; return; }
        if ((P < 0.0) || (P > 2. * g.Rpc)) { status = 0; return; }
        step -= truestep;
        if (!(fabs(step) > 1e-5)) break;
    }
    status = 1;
}

} // namespace S5NS
