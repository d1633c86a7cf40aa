// s5_azimuth.hpp -- geodesic_position_azm and geodesic_timedelay with the Legendre
// integrals of the 2nd / 3rd kind and the Byrd & Friedman integrals they are built from.
//
// ref: src/sim5kerr-geod.c:463-664 (callers), src/sim5elliptic.c:255-450 (F, E, Pi through Carlson's R_F, R_D,
// R_J), :637-816 (integrals of Jacobi functions), :826-1161 (radial and polar integrals).  Expression order
// follows the reference; one place differs by construction: integral_R1 goes through C99 complex csqrt/catan
// on the reference's host path (:776) and keeps the real part -- here the real part is written out for the two
// signs of the argument (atan for a real argument, atanh for an imaginary one), which is the same number up
// to the rounding of libm.
#pragma once
#include "s5_geod.hpp"

namespace S5NS {

// F(acos(c), m)                                                                     ref :255-271
S5_DEV double ell_F_cos(double c, double m)
{
    if (m == 1.0) m = 0.99999999;
    if (c == 1.0) return 0.0;
    double whole = 0.0;
    if (c < 0.0) {
        c = -c;
        whole = 2.0 * carlson_rf(0.0, 1.0 - m, 1.0);
    }
    const double s2 = 1.0 - sq(c);
    return whole + ((whole == 0.0) ? (+1) : (-1)) * sqrt(s2) * carlson_rf(1.0 - s2, 1.0 - s2 * m, 1.0);
}

// E(acos(c), m)                                                                     ref :319-337
S5_DEV double ell_E_cos(double c, double m)
{
    if (m == 1.0) m = 0.99999999;
    if (c == 1.0) return 0.0;
    double whole = 0.0;
    if (c < 0.0) {
        c = -c;
        whole = 2.0 * (carlson_rf(0.0, 1.0 - m, 1.0) - m * carlson_rd(0.0, 1.0 - m, 1.0) / 3.0);
    }
    const double c2 = sq(c);
    const double s = sqrt(1.0 - c2);
    const double q = 1.0 - m + c2 * m;
    return whole + ((whole == 0.0) ? (+1) : (-1)) * s * (carlson_rf(c2, q, 1.0) - sq(s * sqrt(m)) * carlson_rd(c2, q, 1.0) / 3.0);
}

// complete Pi(n, m), Mathematica's sign of n                                        ref :366-378
S5_DEV double ell_Pi_complete(double n, double m)
{
    if (isinf(n)) return 0.0;
    if (m == 1.0) m = 0.99999999;
    if (n == 1.0) n = 0.99999999;
    const double q = 1.0 - m;
    return carlson_rf(0.0, q, 1.0) + n * carlson_rj(0.0, q, 1.0, 1.0 - n) / 3.0;
}

// Pi(acos(c), n, m)                                                                 ref :426-450
S5_DEV double ell_Pi_cos(double c, double n, double m)
{
    if (isinf(n)) return 0.0;
    if (c == 1.0) return 0.0;
    if (c == 0.0) return ell_Pi_complete(n, m);
    if (m == 1.0) m = 0.99999999;
    double whole = 0.0;
    if (c < 0.0) {
        c = -c;
        whole = 2.0 * ((carlson_rf(0.0, 1.0 - m, 1.0) + n * carlson_rj(0.0, 1.0 - m, 1.0, 1.0 - n) / 3.0));
    }
    const double c2 = sq(c);
    const double s = sqrt(1.0 - c2);
    const double ns2 = -n * (1.0 - c2);
    const double q = 1.0 - (1.0 - c2) * m;
    return whole + ((whole == 0.0) ? (+1) : (-1)) * s * (carlson_rf(c2, q, 1.0) - ns2 * carlson_rj(c2, q, 1.0, 1.0 + ns2) / 3.0);
}

// ---- integrals of Jacobi functions (Byrd & Friedman 312, 340, 341) ----
S5_DEV double int_C2_cos(double cn_u, double m)                                     // ref :668-673
{
    return 1. / m * (ell_E_cos(cn_u, m) - (1. - m) * ell_F_cos(cn_u, m));
}

S5_DEV double int_C2(double u, double m)                                            // ref :657-664
{
    double sn, cn, dn;
    sncndn(u, m, sn, cn, dn);
    return 1. / m * (ell_E_cos(cn, m) - (1. - m) * u);
}

S5_DEV double int_Z1(double a, double b, double u, double m)                        // ref :677-690
{
    double sn, cn, dn;
    sncndn(u, m, sn, cn, dn);
    return 1. / a * ((a - b) * ell_Pi_cos(cn, a, m) + b * u);
}

S5_DEV double int_Z2(double a, double b, double u, double m)                        // ref :694-715
{
    double sn, cn, dn;
    sncndn(u, m, sn, cn, dn);
    const double V1 = ell_Pi_cos(cn, a, m);
    const double V2 = 0.5 / ((a - 1.) * (m - a)) * (
                          a * ell_E_cos(cn, m) + (m - a) * u +
                          (2. * a * m + 2. * a - a * a - 3. * m) * V1 -
                          (a * a * sn * cn * dn) / (1. - a * sn * sn));
    const double ab = a - b;
    return 1. / sq(a) * (sq(b) * u + 2. * b * ab * V1 + ab * ab * V2);
}

S5_DEV double int_Rm1(double a, double u, double m)                                 // ref :719-729
{
    return u + a / sqrt(m) * acos(jac_dn(u, m));
}

S5_DEV double int_Rm2(double a, double u, double m)                                 // ref :733-745
{
    const double a2 = sq(a);
    double sn, cn, dn;
    sncndn(u, m, sn, cn, dn);
    return 1 / m * ((m - a2 * (1. - m)) * u + a2 * ell_E_cos(cn, m) + 2 * a * sqrt(m) * acos(dn));
}

// int du / (1 + a cn u), B&F 341.03 / 361.54                                        ref :756-792
S5_DEV double int_R1(double a, double u, double m)
{
    const double a2 = sq(a);
    const double n = a2 / (a2 - 1.);
    double sn, cn, dn;
    sncndn(u, m, sn, cn, dn);
    const double mma = (m + (1. - m) * a2) / (1. - a2);
    const double t = sn / dn;
    double f1;                                    // Re[ csqrt(1/mma) catan(csqrt(mma) t) ]
    if (!(fabs(mma) > 1e-5)) f1 = t;
    else if (mma > 0.0) f1 = sqrt(1. / mma) * atan(sqrt(mma) * t);
    else {
        // csqrt(mma) = i sqrt|mma|, catan(i y) = i atanh(y) (|y| < 1) or +-pi/2 + i atanh(1/y) (|y| > 1);
        // times csqrt(1/mma) = i sqrt(1/|mma|): the real part is -sqrt(1/|mma|) atanh(y or 1/y)
        const double w = sqrt(1. / -mma);
        const double y = sqrt(-mma) * t;
        const double ay = fabs(y);
        const double ath = (ay < 1.0) ? 0.5 * log1p(2. * ay / (1. - ay)) : 0.5 * log1p(2. / (ay - 1.));
        f1 = -w * ((y < 0.0) ? -ath : ath);
    }
    const double ellpi = ell_Pi_cos(cn, n, m);
    return 1. / (1. - a2) * (ellpi + a * f1);
}

S5_DEV double int_R2(double a, double u, double m)                                  // ref :796-816
{
    const double a2 = sq(a);
    const double mma = (m + (1. - m) * a2);
    double sn, cn, dn;
    sncndn(u, m, sn, cn, dn);
    return 1 / (a2 - 1.) / mma * (
               (a2 * (2. * m - 1.) - 2. * m) * int_R1(a, u, m) +
               2. * m * int_Rm1(a, u, m) -
               m * int_Rm2(a, u, m) +
               a * a2 * sn * dn / (1. + a * cn));
}

// ---- radial integrals: four real roots a > b > c > d (B&F 258) ----
S5_DEV double R_r0_re(double a, double b, double c, double d, double X)             // ref :826-838
{
    const double m4 = ((b - c) * (a - d)) / ((a - c) * (b - d));
    const double sn = sqrt(((b - d) * (X - a)) / ((a - d) * (X - b)));
    return 2.0 / sqrt((a - c) * (b - d)) * inv_sn(sn, m4);
}

S5_DEV double R_r0_re_inf(double a, double b, double c, double d)                   // ref :842-854
{
    const double m4 = ((b - c) * (a - d)) / ((a - c) * (b - d));
    const double sn = sqrt((b - d) / (a - d));
    return 2.0 / sqrt((a - c) * (b - d)) * inv_sn(sn, m4);
}

S5_DEV double R_r1_re(double a, double b, double c, double d, double X)             // ref :893-905
{
    const double m2 = ((b - c) * (a - d)) / ((a - c) * (b - d));
    const double sn = sqrt(((b - d) * (X - a)) / ((a - d) * (X - b)));
    const double u = inv_sn(sn, m2);
    const double a2 = (a - d) / (b - d);
    const double b2 = ((a - d) * b) / (a * (b - d));
    const double Z = int_Z1(a2, b2, u, m2) - int_Z1(a2, b2, 0, m2);
    return a * 2.0 / sqrt((a - c) * (b - d)) * Z;
}

S5_DEV double R_r2_re(double a, double b, double c, double d, double X)             // ref :955-967
{
    const double m2 = ((b - c) * (a - d)) / ((a - c) * (b - d));
    const double sn = sqrt(((b - d) * (X - a)) / ((a - d) * (X - b)));
    const double u = inv_sn(sn, m2);
    const double a2 = (a - d) / (b - d);
    const double b2 = ((a - d) * b) / (a * (b - d));
    const double Z = int_Z2(a2, b2, u, m2) - int_Z2(a2, b2, 0, m2);
    return sq(a) * 2.0 / sqrt((a - c) * (b - d)) * Z;
}

// X = infinity is passed as to_inf (the reference has separate *_inf functions: sn = sqrt((b-d)/(a-d)))
S5_DEV double R_rp_re(double a, double b, double c, double d, double p, double X, bool to_inf)   // ref :1017-1043
{
    const double m2 = ((b - c) * (a - d)) / ((a - c) * (b - d));
    const double sn = to_inf ? sqrt((b - d) / (a - d)) : sqrt(((b - d) * (X - a)) / ((a - d) * (X - b)));
    const double u1 = inv_sn(sn, m2);
    const double a2 = (a - d) / (b - d);
    const double c2 = ((p - b) * (a - d)) / ((p - a) * (b - d));
    return -2.0 / sqrt((a - c) * (b - d)) / (p - a) * (int_Z1(c2, a2, u1, m2) - int_Z1(c2, a2, 0, m2));
}

// ---- radial integrals: two real roots a > b and a complex pair u +- i v (B&F 260) ----
struct CcForm { double A, B, m, g; };
S5_DEV CcForm cc_form(double a, double b, double cu, double cv)
{
    CcForm f;
    const double v2 = sq(cv);
    f.A = sqrt(sq(a - cu) + v2);
    f.B = sqrt(sq(b - cu) + v2);
    f.m = (sq(f.A + f.B) - sq(a - b)) / (4. * f.A * f.B);
    f.g = 1. / sqrt(f.A * f.B);
    return f;
}
S5_DEV double cc_cn(const CcForm& f, double a, double b, double X)
{
    return (X * (f.A - f.B) + a * f.B - b * f.A) / (X * (f.A + f.B) - a * f.B - b * f.A);
}

S5_DEV double R_r0_cc(double a, double b, double cu, double cv, double X)           // ref :858-872
{
    const CcForm f = cc_form(a, b, cu, cv);
    return 1. / sqrt(f.A * f.B) * inv_cn(cc_cn(f, a, b, X), f.m);
}

S5_DEV double R_r0_cc_inf(double a, double b, double cu, double cv)                 // ref :876-889
{
    const CcForm f = cc_form(a, b, cu, cv);
    return 1. / sqrt(f.A * f.B) * inv_cn((f.A - f.B) / (f.A + f.B), f.m);
}

S5_DEV double R_r1_cc(double a, double b, double cu, double cv, double X1, double X2)   // ref :910-930
{
    const CcForm f = cc_form(a, b, cu, cv);
    const double A = f.A, B = f.B;
    const double alpha1 = (B * a + b * A) / (B * a - b * A);
    const double alpha2 = (B + A) / (B - A);
    const double u1 = ell_F_cos(cc_cn(f, a, b, X1), f.m);
    const double u2 = ell_F_cos(cc_cn(f, a, b, X2), f.m);
    const double t0 = alpha1 * (u2 - u1);
    const double t1 = (alpha2 - alpha1) * (int_R1(alpha2, u2, f.m) - int_R1(alpha2, u1, f.m));
    return (B * a - b * A) / (B + A) * f.g * (t0 + t1);
}

S5_DEV double R_r2_cc(double a, double b, double cu, double cv, double X1, double X2)   // ref :972-993
{
    const CcForm f = cc_form(a, b, cu, cv);
    const double A = f.A, B = f.B;
    const double alpha1 = (B * a + b * A) / (B * a - b * A);
    const double alpha2 = (B + A) / (B - A);
    const double u1 = ell_F_cos(cc_cn(f, a, b, X1), f.m);
    const double u2 = ell_F_cos(cc_cn(f, a, b, X2), f.m);
    const double t0 = pow(alpha1, 2.) * (u2 - u1);
    const double t1 = 2. * alpha1 * (alpha2 - alpha1) * (int_R1(alpha2, u2, f.m) - int_R1(alpha2, u1, f.m));
    const double t2 = pow(alpha2 - alpha1, 2.) * (int_R2(alpha2, u2, f.m) - int_R2(alpha2, u1, f.m));
    return pow((B * a - b * A) / (B + A), 2.) * f.g * (t0 + t1 + t2);
}

S5_DEV double R_rp_cc2(double a, double b, double cu, double cv, double p, double X1, double X2, bool to_inf)   // ref :1048-1113
{
    const CcForm f = cc_form(a, b, cu, cv);
    const double A = f.A, B = f.B;
    const double alpha1 = (B * a + b * A - p * A - p * B) / (B * a - b * A + p * A - p * B);
    const double alpha2 = (B + A) / (B - A);
    const double u1 = ell_F_cos(cc_cn(f, a, b, X1), f.m);
    const double u2 = to_inf ? ell_F_cos((A - B) / (A + B), f.m) : ell_F_cos(cc_cn(f, a, b, X2), f.m);
    const double t0 = alpha2 * (u2 - u1);
    const double t1 = (alpha1 - alpha2) * (int_R1(alpha1, u2, f.m) - int_R1(alpha1, u1, f.m));
    return (B - A) * f.g / (B * a + b * A - p * A - p * B) * (t0 + t1);
}

// ---- polar integrals (B&F 213) ----
S5_DEV double T_m0(double a2, double b2, double X)                                  // ref :1122-1129
{
    const double m = b2 / (a2 + b2);
    return 1. / sqrt(a2 + b2) * inv_cn(X / sqrt(b2), m);
}

S5_DEV double T_m2(double a2, double b2, double X)                                  // ref :1133-1141
{
    const double m = b2 / (a2 + b2);
    const double cn = X / sqrt(b2);
    return b2 / sqrt(a2 + b2) * (int_C2_cos(cn, m) - int_C2(0, m));
}

S5_DEV double T_mp(double a2, double b2, double p, double X)                        // ref :1145-1161
{
    const double m = b2 / (a2 + b2);
    const double n = b2 / (b2 - p);
    if (X >= 0.0)
        return 1. / sqrt(a2 + b2) / (p - b2) * ell_Pi_cos(X / sqrt(b2), n, m);
    else
        return 1. / sqrt(a2 + b2) / (p - b2) * (2. * ell_Pi_complete(n, m) - ell_Pi_cos(-X / sqrt(b2), n, m));
}

// change of azimuth between infinity and the point (r, m) at position integral P    ref src/sim5kerr-geod.c:463-556
S5_DEV double position_azm(const Geod& g, double r, double m, double P)
{
    double phi = 0.0;
    const bool ppc = (g.nrr > 0) && (P > g.Rpc);
    const double a2 = sq(g.a);
    const double rp = 1. + sqrt(1. - a2);
    const double rm = 1. - sqrt(1. - a2);
    if (g.type == T_RR) {
        const double r1 = g.r1[0], r2 = g.r2[0], r3 = g.r3[0], r4 = g.r4[0];
        const double A = R_rp_re(r1, r2, r3, r4, rp, 0.0, true) + (ppc ? +1 : -1) * R_rp_re(r1, r2, r3, r4, rp, r, false);
        const double B = R_rp_re(r1, r2, r3, r4, rm, 0.0, true) + (ppc ? +1 : -1) * R_rp_re(r1, r2, r3, r4, rm, r, false);
        phi += 1. / sqrt(1. - a2) * (A * (g.a * rp - g.l * a2 / 2.) - B * (g.a * rm - g.l * a2 / 2.));
    } else if (g.type == T_RC) {
        const double r1 = g.r1[0], r2 = g.r2[0];
        const double A = R_rp_cc2(r1, r2, g.r3[0], g.r3[1], rp, r, 0.0, true);
        const double B = R_rp_cc2(r1, r2, g.r3[0], g.r3[1], rm, r, 0.0, true);
        phi += 1. / sqrt(1. - a2) * (A * (g.a * rp - g.l * a2 / 2.) - B * (g.a * rm - g.l * a2 / 2.));
    } else if (g.type == T_RR_DBL || g.type == T_RR_BH || g.type == T_CC) {
        return NAN;
    }
    const double phi_pp = 2.0 * g.l / g.a * T_mp(g.m2m, g.m2p, 1.0, 0.0);
    const double phi_ip = g.l / g.a * T_mp(g.m2m, g.m2p, 1.0, g.cos_i);
    const double phi_mp = g.l / g.a * T_mp(g.m2m, g.m2p, 1.0, m);
    double T;
    double sign_dm = (g.beta >= 0.0) ? +1.0 : -1.0;
    if (sign_dm > 0.0) {
        T = -(g.Tpp - g.Tip);
        phi -= phi_pp - phi_ip;
    } else {
        T = -g.Tip;
        phi -= phi_ip;
    }
    if (P >= T + g.Tpp) {                 // the reference's while-loop leaves after one pass (:545-550)
        T += g.Tpp;
        phi += phi_pp;
        sign_dm = -sign_dm;
    }
    phi += (sign_dm < 0) ? phi_mp : phi_pp - phi_mp;
    return phi;
}

// light-travel time between two points of a geodesic; the reference evaluates the radial part only (its polar
// part is commented out)                                                            ref src/sim5kerr-geod.c:560-664
S5_DEV double timedelay(const Geod& g, double P1, double r1, double m1, double P2, double r2, double m2)
{
    if (P1 > P2) {
        double tmp;
        tmp = P2; P2 = P1; P1 = tmp;
        tmp = r2; r2 = r1; r1 = tmp;
        tmp = m2; m2 = m1; m1 = tmp;
    }
    if (r1 == 0) { r1 = position_rad(g, P1); m1 = position_pol(g, P1); }
    if (r2 == 0) { r2 = position_rad(g, P2); m2 = position_pol(g, P2); }
    const double a2 = sq(g.a);
    const double rp = 1. + sqrt(1. - a2);
    const double rm = 1. - sqrt(1. - a2);
    const double ra = g.r1[0], rb = g.r2[0], rc = g.r3[0], rd = g.r4[0];
    double R0, R1, R2, RA, RB;
    if (g.type == T_RR) {
        const double s = (((P1 > g.Rpc) && (P2 < g.Rpc)) || ((P1 < g.Rpc) && (P2 > g.Rpc))) ? +1 : -1;
        R0 = R_r0_re(ra, rb, rc, rd, r1) + s * R_r0_re(ra, rb, rc, rd, r2);
        R1 = R_r1_re(ra, rb, rc, rd, r1) + s * R_r1_re(ra, rb, rc, rd, r2);
        R2 = R_r2_re(ra, rb, rc, rd, r1) + s * R_r2_re(ra, rb, rc, rd, r2);
        RA = R_rp_re(ra, rb, rc, rd, rp, r1, false) + s * R_rp_re(ra, rb, rc, rd, rp, r2, false);
        RB = R_rp_re(ra, rb, rc, rd, rm, r1, false) + s * R_rp_re(ra, rb, rc, rd, rm, r2, false);
    } else if (g.type == T_RC) {
        const double cu = g.r3[0], cv = g.r3[1];
        const double lo = (r1 < r2) ? r1 : r2, hi = (r1 < r2) ? r2 : r1;
        R0 = R_r0_cc(ra, rb, cu, cv, r1) - R_r0_cc(ra, rb, cu, cv, r2);
        R1 = R_r1_cc(ra, rb, cu, cv, lo, hi);
        R2 = R_r2_cc(ra, rb, cu, cv, lo, hi);
        RA = R_rp_cc2(ra, rb, cu, cv, rp, lo, hi, false);
        RB = R_rp_cc2(ra, rb, cu, cv, rm, lo, hi, false);
    } else if (g.type == T_RR_DBL || g.type == T_RR_BH || g.type == T_CC) {
        return NAN;
    } else {
        return 0.0;
    }
    const double A = (-g.a * g.l + 4.) * rp - 2. * a2;
    const double B = (+g.a * g.l - 4.) * rm + 2. * a2;
    return 0.0 + (4. * fabs(R0) + 2. * fabs(R1) + fabs(R2) + (A * fabs(RA) + B * fabs(RB)) / sqrt(1. - a2));
}

} // namespace S5NS
