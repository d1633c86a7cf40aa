/*
 * sim5_oracle.c -- CPU restatement of the SIM5 per-ray hot path (see sim5_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP kernels, never the product.
 * Parity status: PINNED against oracle/_ref/libsim5ref.so (the unmodified reference)
 * and the golden vectors in tests/golden/.
 *
 * The arithmetic below follows the evaluation order of the reference expression by
 * expression (the order of floating-point operations is part of the algorithm when
 * parity is bit-level); naming, control flow and layout are this project's own.
 * Each routine cites the reference lines it restates ("ref:" = /root/reference/).
 * Diagnostics that the reference prints to stderr on the hot path are dropped; return
 * values in those situations are kept.
 */
#include "sim5_oracle.h"

#include <complex.h>
#include <float.h>
#include <math.h>
#include <string.h>

#define SQ(x) ((x) * (x))

/* ================================================================================== */
/*  Carlson symmetric integrals                                                       */
/* ================================================================================== */

/* R_F by the duplication theorem; ref: src/sim5elliptic.c:19-52 */
double orc_rf(double x, double y, double z)
{
    const double tol = 0.0003, third = 1.0 / 3.0;
    double dx, dy, dz, mu;
    for (;;) {
        double sx = sqrt(x), sy = sqrt(y), sz = sqrt(z);
        double lam = sx * (sy + sz) + sy * sz;
        x = 0.25 * (x + lam);
        y = 0.25 * (y + lam);
        z = 0.25 * (z + lam);
        mu = third * (x + y + z);
        dx = (mu - x) / mu;
        dy = (mu - y) / mu;
        dz = (mu - z) / mu;
        if (!(fmax(fmax(fabs(dx), fabs(dy)), fabs(dz)) > tol)) break;
    }
    double e2 = dx * dy - dz * dz;
    double e3 = dx * dy * dz;
    return (1.0 + ((1.0 / 24.0) * e2 - 0.1 - (3.0 / 44.0) * e3) * e2 + (1.0 / 14.0) * e3) / sqrt(mu);
}

/* R_D; ref: src/sim5elliptic.c:59-98 */
double orc_rd(double x, double y, double z)
{
    const double tol = 0.0003;
    const double c1 = 3.0 / 14.0, c2 = 1.0 / 6.0, c3 = 9.0 / 22.0, c4 = 3.0 / 26.0;
    const double c5 = 0.25 * c3, c6 = 1.5 * c4;
    double acc = 0.0, w = 1.0, dx, dy, dz, mu;
    for (;;) {
        double sx = sqrt(x), sy = sqrt(y), sz = sqrt(z);
        double lam = sx * (sy + sz) + sy * sz;
        acc += w / (sz * (z + lam));
        w = 0.25 * w;
        x = 0.25 * (x + lam);
        y = 0.25 * (y + lam);
        z = 0.25 * (z + lam);
        mu = 0.2 * (x + y + 3.0 * z);
        dx = (mu - x) / mu;
        dy = (mu - y) / mu;
        dz = (mu - z) / mu;
        if (!(fmax(fmax(fabs(dx), fabs(dy)), fabs(dz)) > tol)) break;
    }
    double ea = dx * dy, eb = dz * dz, ec = ea - eb, ed = ea - 6.0 * eb, ee = ed + ec + ec;
    return 3.0 * acc + w * (1.0 + ed * (-c1 + c5 * ed - c6 * dz * ee)
        + dz * (c2 * ee + dz * (-c3 * ec + dz * c4 * ea))) / (mu * sqrt(mu));
}

/* R_C (Cauchy principal value for y<0); ref: src/sim5elliptic.c:105-137 */
double orc_rc(double x, double y)
{
    const double tol = 0.0003, third = 1.0 / 3.0;
    double xt, yt, w, mu, s;
    if (y > 0.0) {
        xt = x; yt = y; w = 1.0;
    } else {
        xt = x - y; yt = -y; w = sqrt(x) / sqrt(xt);
    }
    do {
        double lam = 2.0 * sqrt(xt) * sqrt(yt) + yt;
        xt = 0.25 * (xt + lam);
        yt = 0.25 * (yt + lam);
        mu = third * (xt + yt + yt);
        s = (yt - mu) / mu;
    } while (fabs(s) > tol);
    return w * (1.0 + s * s * (0.3 + s * ((1.0 / 7.0) + s * (0.375 + s * (9.0 / 22.0))))) / sqrt(mu);
}

/* R_J; ref: src/sim5elliptic.c:145-206 */
double orc_rj(double x, double y, double z, double p)
{
    const double tol = 0.0003;
    const double tiny = pow(5.0 * DBL_MIN, 1. / 3.), big = 0.3 * pow(0.1 * DBL_MAX, 1. / 3.);
    const double c1 = 3.0 / 14.0, c2 = 1.0 / 3.0, c3 = 3.0 / 22.0, c4 = 3.0 / 26.0;
    const double c5 = 0.75 * c3, c6 = 1.5 * c4, c7 = 0.5 * c2, c8 = c3 + c3;
    if ((fmin(fmin(x, y), z) < 0.0) || (fmin(fmin(x + y, x + z), fmin(y + z, fabs(p))) < tiny) ||
        (fmax(fmax(x, y), fmax(z, fabs(p))) > big))
        return 0.0;
    double a = 0.0, b = 0.0, rcx = 0.0, acc = 0.0, w = 1.0;
    double xt, yt, zt, pt, dx, dy, dz, dp, mu;
    if (p > 0.0) {
        xt = x; yt = y; zt = z; pt = p;
    } else {
        xt = fmin(fmin(x, y), z);
        zt = fmax(fmax(x, y), z);
        yt = x + y + z - xt - zt;
        a = 1.0 / (yt - p);
        b = a * (zt - yt) * (yt - xt);
        pt = yt + b;
        double rho = xt * zt / yt;
        double tau = p * pt / yt;
        rcx = orc_rc(rho, tau);
    }
    for (;;) {
        double sx = sqrt(xt), sy = sqrt(yt), sz = sqrt(zt);
        double lam = sx * (sy + sz) + sy * sz;
        double al = SQ(pt * (sx + sy + sz) + sx * sy * sz);
        double be = pt * SQ(pt + lam);
        acc += w * orc_rc(al, be);
        w = 0.25 * w;
        xt = 0.25 * (xt + lam);
        yt = 0.25 * (yt + lam);
        zt = 0.25 * (zt + lam);
        pt = 0.25 * (pt + lam);
        mu = 0.2 * (xt + yt + zt + pt + pt);
        dx = (mu - xt) / mu;
        dy = (mu - yt) / mu;
        dz = (mu - zt) / mu;
        dp = (mu - pt) / mu;
        if (!(fmax(fmax(fabs(dx), fabs(dy)), fmax(fabs(dz), fabs(dp))) > tol)) break;
    }
    double ea = dx * (dy + dz) + dy * dz;
    double eb = dx * dy * dz;
    double ec = dp * dp;
    double ed = ea - 3.0 * ec;
    double ee = eb + 2.0 * dp * (ea - ec);
    double ans = 3.0 * acc + w * (1.0 + ed * (-c1 + c5 * ed - c6 * ee) + eb * (c7 + dp * (-c8 + dp * c4))
        + dp * ea * (c2 - dp * c3) - c2 * dp * ec) / (mu * sqrt(mu));
    if (p <= 0.0) ans = a * (b * ans + 3.0 * (rcx - orc_rf(xt, yt, zt)));
    return ans;
}

/* ================================================================================== */
/*  Legendre / Jacobi functions on top of R_F                                         */
/* ================================================================================== */

/* K(m); ref: src/sim5elliptic.c:218-225 */
double orc_elliptic_k(double m)
{
    if (m == 1.0) m = 1.0 - 1e-8;
    return orc_rf(0, 1.0 - m, 1.0);
}

/* F(asin(s), m); ref: src/sim5elliptic.c:274-284 */
double orc_elliptic_f_sin(double s, double m)
{
    if (m == 1.0) m = 0.99999999;
    if (s == 0.0) return 0.0;
    double s2 = SQ(s);
    return s * orc_rf(1. - s2, 1.0 - s2 * m, 1.0);
}

/* sn^-1(z|m); ref: src/sim5elliptic.c:481-486 */
double orc_jacobi_isn(double z, double m)
{
    if (fabs(m - 0.0) < 1e-8) return asin(z);
    if (fabs(m - 1.0) < 1e-8) return log(sqrt((1. + z) / (1. - z)));
    return z * orc_rf(1.0 - z * z, 1.0 - m * z * z, 1.0);
}

/* cn^-1(z|m), including the z<0 continuation; ref: src/sim5elliptic.c:493-514 */
double orc_jacobi_icn(double z, double m)
{
    if ((z > +1.0) && (z < +1.0 + 1e-8)) z = +1.0;
    if ((z < -1.0) && (z > -1.0 - 1e-8)) z = -1.0;
    if ((m > +1.0) && (m < +1.0 + 1e-8)) m = 1.0;
    if ((m < 0.0) && (m > 0.0 - 1e-8)) m = 0.0;

    if (z == 0.0) return orc_elliptic_k(m);
    if (z == 1.0) return 0.0;
    if (m == 0.0) return acos(z);
    if (m == 1.0) return log((1. + sqrt(1. - z)) / z);

    double base = sqrt(1. - z * z) * orc_rf(z * z, 1.0 - m * (1. - z * z), 1.0);
    return (z > 0.0) ? base : 2. / sqrt(1. - m) * orc_elliptic_f_sin(-z, m / (m - 1.)) + base;
}

/* tn^-1(z|m); ref: src/sim5elliptic.c:523-528 */
double orc_jacobi_itn(double z, double m)
{
    if (m == 0.0) return atan(z);
    if (m == 1.0) return log(z + sqrt(1. + z * z));
    return orc_jacobi_isn(sqrt(z * z / (1. + z * z)), m);
}

/* F(acos(c), m); ref: src/sim5elliptic.c:255-271 */
double orc_elliptic_f_cos(double c, double m)
{
    if (m == 1.0) m = 0.99999999;
    if (c == 1.0) return 0.0;
    double whole = 0.0;
    if (c < 0.0) {
        c = -c;
        whole = 2.0 * orc_rf(0.0, 1.0 - m, 1.0);
    }
    double s2 = 1.0 - SQ(c);
    return whole + ((whole == 0.0) ? (+1) : (-1)) * sqrt(s2) * orc_rf(1.0 - s2, 1.0 - s2 * m, 1.0);
}

/* E(acos(c), m) through R_F and R_D; ref: src/sim5elliptic.c:319-337 */
double orc_elliptic_e_cos(double c, double m)
{
    if (m == 1.0) m = 0.99999999;
    if (c == 1.0) return 0.0;
    double whole = 0.0;
    if (c < 0.0) {
        c = -c;
        whole = 2.0 * (orc_rf(0.0, 1.0 - m, 1.0) - m * orc_rd(0.0, 1.0 - m, 1.0) / 3.0);
    }
    double c2 = SQ(c);
    double s = sqrt(1.0 - c2);
    double q = 1.0 - m + c2 * m;
    return whole + ((whole == 0.0) ? (+1) : (-1)) * s * (orc_rf(c2, q, 1.0) - SQ(s * sqrt(m)) * orc_rd(c2, q, 1.0) / 3.0);
}

/* complete Pi(n, m), Mathematica's sign of n; ref: src/sim5elliptic.c:366-378 */
double orc_elliptic_pi_complete(double n, double m)
{
    if (isinf(n)) return 0.0;
    if (m == 1.0) m = 0.99999999;
    if (n == 1.0) n = 0.99999999;
    double q = 1.0 - m;
    return orc_rf(0.0, q, 1.0) + n * orc_rj(0.0, q, 1.0, 1.0 - n) / 3.0;
}

/* Pi(acos(c), n, m); ref: src/sim5elliptic.c:426-450 */
double orc_elliptic_pi_cos(double c, double n, double m)
{
    if (isinf(n)) return 0.0;
    if (c == 1.0) return 0.0;
    if (c == 0.0) return orc_elliptic_pi_complete(n, m);
    if (m == 1.0) m = 0.99999999;
    double whole = 0.0;
    if (c < 0.0) {
        c = -c;
        whole = 2.0 * ((orc_rf(0.0, 1.0 - m, 1.0) + n * orc_rj(0.0, 1.0 - m, 1.0, 1.0 - n) / 3.0));
    }
    double c2 = SQ(c);
    double s = sqrt(1.0 - c2);
    double ns2 = -n * (1.0 - c2);
    double q = 1.0 - (1.0 - c2) * m;
    return whole + ((whole == 0.0) ? (+1) : (-1)) * s * (orc_rf(c2, q, 1.0) - ns2 * orc_rj(c2, q, 1.0, 1.0 + ns2) / 3.0);
}

/* sn, cn, dn by descending Landen (AGM) transformation; ref: src/sim5elliptic.c:536-598 */
void orc_jacobi_sncndn(double u, double m, double *sn, double *cn, double *dn)
{
    if (m == 1.0) m = 0.999999999;
    const double conv = 1.0e-8;
    double a, b, c = 0.0, d = 1.0, emc = 1.0 - m;
    double lvl_a[13], lvl_g[13];
    int top = 0, flipped = 0;

    if (emc == 0.0) {
        *cn = 1.0 / cosh(u);
        *dn = *cn;
        *sn = tanh(u);
        return;
    }
    flipped = (emc < 0.0);
    if (flipped) {
        d = 1.0 - emc;
        emc /= -1.0 / d;
        u *= (d = sqrt(d));
    }
    a = 1.0;
    *dn = 1.0;
    for (int i = 0; i < 13; i++) {
        top = i;
        lvl_a[i] = a;
        lvl_g[i] = (emc = sqrt(emc));
        c = 0.5 * (a + emc);
        if (fabs(a - emc) <= conv * a) break;
        emc *= a;
        a = c;
    }
    u *= c;
    *sn = sin(u);
    *cn = cos(u);
    if (*sn != 0.0) {
        a = (*cn) / (*sn);
        c *= a;
        for (int i = top; i >= 0; i--) {
            b = lvl_a[i];
            a *= c;
            c *= *dn;
            *dn = (lvl_g[i] + a) / (b + a);
            a = c / b;
        }
        a = 1.0 / sqrt(c * c + 1.0);
        *sn = ((*sn) >= 0.0 ? a : -a);
        *cn = c * (*sn);
    }
    if (flipped) {
        a = *dn;
        *dn = *cn;
        *cn = a;
        *sn /= d;
    }
}

/* ref: src/sim5elliptic.c:605-630 */
double orc_jacobi_sn(double u, double m) { double s, c, d; orc_jacobi_sncndn(u, m, &s, &c, &d); return s; }
double orc_jacobi_cn(double u, double m) { double s, c, d; orc_jacobi_sncndn(u, m, &s, &c, &d); return c; }
double orc_jacobi_dn(double u, double m) { double s, c, d; orc_jacobi_sncndn(u, m, &s, &c, &d); return d; }

/* ================================================================================== */
/*  Byrd & Friedman integrals of Jacobi functions and the radial / polar integrals    */
/*  built on them (used by geodesic_position_azm and geodesic_timedelay)              */
/* ================================================================================== */

/* int cn^2 du from the cn of the upper limit; ref: src/sim5elliptic.c:668-673 */
double orc_integral_C2_cos(double cn_u, double m)
{
    return 1. / m * (orc_elliptic_e_cos(cn_u, m) - (1. - m) * orc_elliptic_f_cos(cn_u, m));
}

/* int_0^u cn^2 du; ref: src/sim5elliptic.c:657-664 */
double orc_integral_C2(double u, double m)
{
    double sn, cn, dn;
    orc_jacobi_sncndn(u, m, &sn, &cn, &dn);
    return 1. / m * (orc_elliptic_e_cos(cn, m) - (1. - m) * u);
}

/* int (1 - b sn^2)/(1 - a sn^2) du, B&F 340.01; ref: src/sim5elliptic.c:677-690 */
double orc_integral_Z1(double a, double b, double u, double m)
{
    double sn, cn, dn;
    orc_jacobi_sncndn(u, m, &sn, &cn, &dn);
    return 1. / a * ((a - b) * orc_elliptic_pi_cos(cn, a, m) + b * u);
}

/* int (1 - b sn^2)^2/(1 - a sn^2)^2 du, B&F 340.02; ref: src/sim5elliptic.c:694-715 */
double orc_integral_Z2(double a, double b, double u, double m)
{
    double sn, cn, dn;
    orc_jacobi_sncndn(u, m, &sn, &cn, &dn);
    double V1 = orc_elliptic_pi_cos(cn, a, m);
    double V2 = 0.5 / ((a - 1.) * (m - a)) * (
                    a * orc_elliptic_e_cos(cn, m) + (m - a) * u +
                    (2. * a * m + 2. * a - a * a - 3. * m) * V1 -
                    (a * a * sn * cn * dn) / (1. - a * sn * sn));
    double ab = a - b;
    return 1. / SQ(a) * (SQ(b) * u + 2. * b * ab * V1 + ab * ab * V2);
}

/* int (1 + a cn) du, B&F 341.00; ref: src/sim5elliptic.c:719-729 */
double orc_integral_Rm1(double a, double u, double m)
{
    return u + a / sqrt(m) * acos(orc_jacobi_dn(u, m));
}

/* int (1 + a cn)^2 du, B&F 341.01; ref: src/sim5elliptic.c:733-745 */
double orc_integral_Rm2(double a, double u, double m)
{
    double a2 = SQ(a);
    double sn, cn, dn;
    orc_jacobi_sncndn(u, m, &sn, &cn, &dn);
    return 1 / m * ((m - a2 * (1. - m)) * u + a2 * orc_elliptic_e_cos(cn, m) + 2 * a * sqrt(m) * acos(dn));
}

/* int du/(1 + a cn), B&F 341.03 / 361.54, complex intermediate as on the reference's host path;
   ref: src/sim5elliptic.c:756-792 */
double orc_integral_R1(double a, double u, double m)
{
    double a2 = SQ(a);
    double n = a2 / (a2 - 1.);
    double sn, cn, dn;
    orc_jacobi_sncndn(u, m, &sn, &cn, &dn);
    double mma = (m + (1. - m) * a2) / (1. - a2);
    orc_cplx f1 = (fabs(mma) > 1e-5)
                      ? csqrt((1. / mma) + _Complex_I * 0.0) * catan(csqrt((mma) + _Complex_I * 0.0) * sn / dn)
                      : ((sn / dn) + _Complex_I * 0.0);
    orc_cplx ellpi = orc_elliptic_pi_cos(cn, n, m);
    orc_cplx res = 1. / (1. - a2) * (ellpi + a * f1);
    return creal(res);
}

/* int du/(1 + a cn)^2, B&F 341.04; ref: src/sim5elliptic.c:796-816 */
double orc_integral_R2(double a, double u, double m)
{
    double a2 = SQ(a);
    double mma = (m + (1. - m) * a2);
    double sn, cn, dn;
    orc_jacobi_sncndn(u, m, &sn, &cn, &dn);
    return 1 / (a2 - 1.) / mma * (
               (a2 * (2. * m - 1.) - 2. * m) * orc_integral_R1(a, u, m) +
               2. * m * orc_integral_Rm1(a, u, m) -
               m * orc_integral_Rm2(a, u, m) +
               a * a2 * sn * dn / (1. + a * cn));
}

/* int_a^X dx/sqrt((x-a)(x-b)(x-c)(x-d)), four real roots, B&F 258.00; ref: src/sim5elliptic.c:826-838 */
double orc_integral_R_r0_re(double a, double b, double c, double d, double X)
{
    double m4 = ((b - c) * (a - d)) / ((a - c) * (b - d));
    double sn = sqrt(((b - d) * (X - a)) / ((a - d) * (X - b)));
    return 2.0 / sqrt((a - c) * (b - d)) * orc_jacobi_isn(sn, m4);
}

/* the same up to infinity; ref: src/sim5elliptic.c:842-854 */
double orc_integral_R_r0_re_inf(double a, double b, double c, double d)
{
    double m4 = ((b - c) * (a - d)) / ((a - c) * (b - d));
    double sn = sqrt((b - d) / (a - d));
    return 2.0 / sqrt((a - c) * (b - d)) * orc_jacobi_isn(sn, m4);
}

/* two real roots and a complex pair c, c*, B&F 260.00; ref: src/sim5elliptic.c:858-872 */
double orc_integral_R_r0_cc(double a, double b, orc_cplx c, double X)
{
    double u = creal(c);
    double v2 = SQ(cimag(c));
    double A = sqrt(SQ(a - u) + v2);
    double B = sqrt(SQ(b - u) + v2);
    double m2 = (SQ(A + B) - SQ(a - b)) / (4. * A * B);
    double cn = (X * (A - B) + a * B - b * A) / (X * (A + B) - a * B - b * A);
    return 1. / sqrt(A * B) * orc_jacobi_icn(cn, m2);
}

/* ref: src/sim5elliptic.c:876-889 */
double orc_integral_R_r0_cc_inf(double a, double b, orc_cplx c)
{
    double u = creal(c);
    double v2 = SQ(cimag(c));
    double A = sqrt(SQ(a - u) + v2);
    double B = sqrt(SQ(b - u) + v2);
    double m2 = (SQ(A + B) - SQ(a - b)) / (4. * A * B);
    double cn = (A - B) / (A + B);
    return 1. / sqrt(A * B) * orc_jacobi_icn(cn, m2);
}

/* int_a^X x dx/sqrt(...), four real roots, B&F 258.11; ref: src/sim5elliptic.c:893-905 */
double orc_integral_R_r1_re(double a, double b, double c, double d, double X)
{
    double m2 = ((b - c) * (a - d)) / ((a - c) * (b - d));
    double sn = sqrt(((b - d) * (X - a)) / ((a - d) * (X - b)));
    double u = orc_jacobi_isn(sn, m2);
    double a2 = (a - d) / (b - d);
    double b2 = ((a - d) * b) / (a * (b - d));
    double Z = orc_integral_Z1(a2, b2, u, m2) - orc_integral_Z1(a2, b2, 0, m2);
    return a * 2.0 / sqrt((a - c) * (b - d)) * Z;
}

/* helper of the complex-pair forms: the amplitude u(X) = F(acos(cn(X)), m); ref: src/sim5elliptic.c:924-925 */
static double cc_amplitude(double a, double b, double A, double B, double X, double m)
{
    return orc_elliptic_f_cos((X * (A - B) + a * B - b * A) / (X * (A + B) - a * B - b * A), m);
}

/* int_X1^X2 x dx/sqrt(...), complex pair, B&F 260.03; ref: src/sim5elliptic.c:910-930 */
double orc_integral_R_r1_cc(double a, double b, orc_cplx c, double X1, double X2)
{
    double u = creal(c);
    double v2 = SQ(cimag(c));
    double A = sqrt(SQ(a - u) + v2);
    double B = sqrt(SQ(b - u) + v2);
    double m = (SQ(A + B) - SQ(a - b)) / (4. * A * B);
    double g = 1. / sqrt(A * B);
    double alpha1 = (B * a + b * A) / (B * a - b * A);
    double alpha2 = (B + A) / (B - A);
    double u1 = cc_amplitude(a, b, A, B, X1, m);
    double u2 = cc_amplitude(a, b, A, B, X2, m);
    double t0 = alpha1 * (u2 - u1);
    double t1 = (alpha2 - alpha1) * (orc_integral_R1(alpha2, u2, m) - orc_integral_R1(alpha2, u1, m));
    return (B * a - b * A) / (B + A) * g * (t0 + t1);
}

/* int_a^X x^2 dx/sqrt(...), four real roots; ref: src/sim5elliptic.c:955-967 */
double orc_integral_R_r2_re(double a, double b, double c, double d, double X)
{
    double m2 = ((b - c) * (a - d)) / ((a - c) * (b - d));
    double sn = sqrt(((b - d) * (X - a)) / ((a - d) * (X - b)));
    double u = orc_jacobi_isn(sn, m2);
    double a2 = (a - d) / (b - d);
    double b2 = ((a - d) * b) / (a * (b - d));
    double Z = orc_integral_Z2(a2, b2, u, m2) - orc_integral_Z2(a2, b2, 0, m2);
    return SQ(a) * 2.0 / sqrt((a - c) * (b - d)) * Z;
}

/* int_X1^X2 x^2 dx/sqrt(...), complex pair; ref: src/sim5elliptic.c:972-993 */
double orc_integral_R_r2_cc(double a, double b, orc_cplx c, double X1, double X2)
{
    double u = creal(c);
    double v2 = SQ(cimag(c));
    double A = sqrt(SQ(a - u) + v2);
    double B = sqrt(SQ(b - u) + v2);
    double m = (SQ(A + B) - SQ(a - b)) / (4. * A * B);
    double g = 1. / sqrt(A * B);
    double alpha1 = (B * a + b * A) / (B * a - b * A);
    double alpha2 = (B + A) / (B - A);
    double u1 = cc_amplitude(a, b, A, B, X1, m);
    double u2 = cc_amplitude(a, b, A, B, X2, m);
    double t0 = pow(alpha1, 2.) * (u2 - u1);
    double t1 = 2. * alpha1 * (alpha2 - alpha1) * (orc_integral_R1(alpha2, u2, m) - orc_integral_R1(alpha2, u1, m));
    double t2 = pow(alpha2 - alpha1, 2.) * (orc_integral_R2(alpha2, u2, m) - orc_integral_R2(alpha2, u1, m));
    return pow((B * a - b * A) / (B + A), 2.) * g * (t0 + t1 + t2);
}

/* int_a^X dx/[(x-p) sqrt(...)], four real roots, B&F 258.39; ref: src/sim5elliptic.c:1017-1028 */
double orc_integral_R_rp_re(double a, double b, double c, double d, double p, double X)
{
    double m2 = ((b - c) * (a - d)) / ((a - c) * (b - d));
    double sn = sqrt(((b - d) * (X - a)) / ((a - d) * (X - b)));
    double u1 = orc_jacobi_isn(sn, m2);
    double a2 = (a - d) / (b - d);
    double c2 = ((p - b) * (a - d)) / ((p - a) * (b - d));
    return -2.0 / sqrt((a - c) * (b - d)) / (p - a) * (orc_integral_Z1(c2, a2, u1, m2) - orc_integral_Z1(c2, a2, 0, m2));
}

/* ref: src/sim5elliptic.c:1032-1043 */
double orc_integral_R_rp_re_inf(double a, double b, double c, double d, double p)
{
    double m2 = ((b - c) * (a - d)) / ((a - c) * (b - d));
    double sn = sqrt((b - d) / (a - d));
    double u1 = orc_jacobi_isn(sn, m2);
    double a2 = (a - d) / (b - d);
    double c2 = ((p - b) * (a - d)) / ((p - a) * (b - d));
    return -2.0 / sqrt((a - c) * (b - d)) / (p - a) * (orc_integral_Z1(c2, a2, u1, m2) - orc_integral_Z1(c2, a2, 0, m2));
}

/* int_X1^X2 dx/[(x-p) sqrt(...)], complex pair, B&F 260.04; ref: src/sim5elliptic.c:1048-1078.
   X2 < 0 stands for the upper limit at infinity (ref :1083-1113, cn = (A-B)/(A+B)) */
static double rp_cc2_core(double a, double b, orc_cplx c, double p, double X1, double X2, int to_infinity)
{
    double u = creal(c);
    double v2 = SQ(cimag(c));
    double A = sqrt(SQ(a - u) + v2);
    double B = sqrt(SQ(b - u) + v2);
    double m = (SQ(A + B) - SQ(a - b)) / (4. * A * B);
    double g = 1. / sqrt(A * B);
    double alpha1 = (B * a + b * A - p * A - p * B) / (B * a - b * A + p * A - p * B);
    double alpha2 = (B + A) / (B - A);
    double u1 = cc_amplitude(a, b, A, B, X1, m);
    double u2 = to_infinity ? orc_elliptic_f_cos((A - B) / (A + B), m) : cc_amplitude(a, b, A, B, X2, m);
    double t0 = alpha2 * (u2 - u1);
    double t1 = (alpha1 - alpha2) * (orc_integral_R1(alpha1, u2, m) - orc_integral_R1(alpha1, u1, m));
    return (B - A) * g / (B * a + b * A - p * A - p * B) * (t0 + t1);
}

double orc_integral_R_rp_cc2(double a, double b, orc_cplx c, double p, double X1, double X2)
{
    return rp_cc2_core(a, b, c, p, X1, X2, 0);
}

double orc_integral_R_rp_cc2_inf(double a, double b, orc_cplx c, double p, double X1)
{
    return rp_cc2_core(a, b, c, p, X1, 0.0, 1);
}

/* int_X^b dx/sqrt((a^2+x^2)(b^2-x^2)), B&F 213.00; ref: src/sim5elliptic.c:1122-1129 */
double orc_integral_T_m0(double a2, double b2, double X)
{
    double m = b2 / (a2 + b2);
    return 1. / sqrt(a2 + b2) * orc_jacobi_icn(X / sqrt(b2), m);
}

/* int_X^b x^2 dx/sqrt(...), B&F 213.06; ref: src/sim5elliptic.c:1133-1141 */
double orc_integral_T_m2(double a2, double b2, double X)
{
    double m = b2 / (a2 + b2);
    double cn = X / sqrt(b2);
    return b2 / sqrt(a2 + b2) * (orc_integral_C2_cos(cn, m) - orc_integral_C2(0, m));
}

/* int_X^b dx/[(p - x^2) sqrt(...)], B&F 213.02; ref: src/sim5elliptic.c:1145-1161 */
double orc_integral_T_mp(double a2, double b2, double p, double X)
{
    double m = b2 / (a2 + b2);
    double n = b2 / (b2 - p);
    if (X >= 0.0)
        return 1. / sqrt(a2 + b2) / (p - b2) * orc_elliptic_pi_cos(X / sqrt(b2), n, m);
    else
        return 1. / sqrt(a2 + b2) / (p - b2) * (2. * orc_elliptic_pi_complete(n, m) - orc_elliptic_pi_cos(-X / sqrt(b2), n, m));
}

/* ================================================================================== */
/*  Kerr spacetime                                                                    */
/* ================================================================================== */

/* ref: src/sim5kerr.c:981-984 */
double orc_r_bh(double a) { return 1. + sqrt(1. - SQ(a)); }

/* ref: src/sim5kerr.c:1007-1034 */
double orc_r_mb(double a) { return (2. - a) + 2. * sqrt(1. - a); }
double orc_r_ph(double a) { return 2.0 * (1.0 + cos(2. / 3. * acos(-a))); }

/* ISCO radius (prograde branch only, cbrt); ref: src/sim5kerr.c:994-1004 */
double orc_r_ms(double a)
{
    double z1 = 1. + cbrt(1. - SQ(a)) * (cbrt(1. + a) + cbrt(1. - a));
    double z2 = sqrt(3. * SQ(a) + SQ(z1));
    return 3. + z2 - sqrt((3. - z1) * (3. + z1 + 2. * z2));
}

/* ref: src/sim5kerr.c:31-50 */
void orc_flat_metric(double r, double m, orc_metric *g)
{
    g->a = 0.0; g->r = r; g->m = m;
    g->g00 = -1.0;
    g->g11 = +1.0;
    g->g22 = +r * r;
    g->g33 = +r * r * (1. - m * m);
    g->g03 = 0.0;
}

/* covariant Boyer-Lindquist metric; ref: src/sim5kerr.c:75-101 */
void orc_kerr_metric(double a, double r, double m, orc_metric *g)
{
    double r2 = SQ(r), a2 = SQ(a), m2 = SQ(m);
    double S = r2 + a2 * m2;
    double s2_S = (1.0 - m2) / S;
    g->a = a; g->r = r; g->m = m;
    g->g00 = -1. + 2.0 * r / S;
    g->g11 = S / (r2 - 2. * r + a2);
    g->g22 = S;
    g->g33 = ((a2 + r2) * S + 2. * r * a2 * s2_S * S) * s2_S;
    g->g03 = -2. * a * r * s2_S;
}

/* ref: src/sim5kerr.c:105-132 */
void orc_kerr_metric_contravariant(double a, double r, double m, orc_metric *g)
{
    double r2 = SQ(r), a2 = SQ(a), m2 = SQ(m);
    double S = r2 + a2 * m2;
    double SD = S * (r2 - 2. * r + a2);
    g->a = a; g->r = r; g->m = m;
    g->g00 = -SQ(r2 + a2) / SD + a2 * (1. - m2) / S;
    g->g11 = (r2 - 2. * r + a2) / S;
    g->g22 = 1. / S;
    g->g33 = 1. / S / (1. - m2) - a2 / SD;
    g->g03 = -2. * a * r / SD;
}

/* Kerr-Newman metric (charge Q), covariant and contravariant; ref: src/sim5kerr.c:136-163, 168-194 */
void orc_kerr_newman_metric(double a, double Q, double r, double m, orc_metric *g)
{
    double rQ = SQ(Q), r2 = SQ(r), a2 = SQ(a), m2 = SQ(m);
    double S = r2 + a2 * m2;
    double s2_S = (1.0 - m2) / S;
    g->a = a; g->r = r; g->m = m;
    g->g00 = -1. + (2.0 * r - rQ) / S;
    g->g11 = S / (r2 - 2. * r + a2 + rQ);
    g->g22 = S;
    g->g33 = ((a2 + r2) * S + (2. * r - rQ) * a2 * s2_S * S) * s2_S;
    g->g03 = -a * (2. * r - rQ) * s2_S;
}

void orc_kerr_newman_metric_contravariant(double a, double Q, double r, double m, orc_metric *g)
{
    double rQ = SQ(Q), r2 = SQ(r), a2 = SQ(a), m2 = SQ(m);
    double S = r2 + a2 * m2;
    double SD = S * (r2 - 2. * r + a2 + rQ);
    g->a = a; g->r = r; g->m = m;
    g->g00 = -SQ(r2 + a2) / SD + a2 * (1. - m2) / S;
    g->g11 = (r2 - 2. * r + a2 + rQ) / S;
    g->g22 = 1. / S;
    g->g33 = 1. / S / (1. - m2) - a2 / SD;
    g->g03 = a * (-2. * r + rQ) / SD;
}

/* Kerr-Newman connection, storage as orc_kerr_connection; ref: src/sim5kerr.c:321-397 */
void orc_kerr_newman_connection(double a, double Q, double r, double m, double G[4][4][4])
{
    double rS = 2.0 * r;
    double rQ = SQ(Q);
    double s = sqrt(1. - m * m);
    double cs = s * m;
    double c2 = m * m;
    double s2 = s * s;
    double cc = c2 - s2;
    double CC = 8. * c2 * c2 - 8. * c2 + 1.;
    double a2 = a * a;
    double a4 = a2 * a2;
    double a2cc = a2 * cc;
    double a2c2 = a2 * c2;
    double a2cs = a2 * cs;
    double r2 = r * r;
    double r3 = r2 * r;
    double a2_r2 = a2 + r2;
    double R = pow(a2 + 2. * r2 + a2cc, 2.);
    double D = r2 - 2. * r + a2 + rQ;
    double S = r2 + a2c2;
    double S_1 = 1. / S;
    double S_3 = 1. / (S * S * S);
    double R_1 = 1. / R;
    double m_s = m / s;
    double DR_1 = R_1 / D;
    double DS_1 = S_1 / D;
    double dbl_r2 = 2. * r2;
    memset(G, 0, 64 * sizeof(double));
    G[0][0][1] = 2.0 * 4.0 * (a2_r2) * (r * (r - rQ) - a2c2) * DR_1;
    G[0][0][2] = 2.0 * -4.0 * a2cs * (rS - rQ) * R_1;
    G[0][1][3] = 2.0 * 4.0 * a * s2 * (-a2 * (r2 - r * rQ) - r3 * (3. * r - 2. * rQ) + a2cc * (a2 - r2 + r * rQ)) * DR_1;
    G[0][2][3] = -G[0][0][2] * s2 * a;
    G[1][0][0] = D * (r * (r - rQ) - a2c2) * S_3;
    G[1][0][3] = -2.0 * G[1][0][0] * a * s2;
    G[1][1][1] = (r * (a2 - r + rQ) + a2 * (1. - r) * c2) * DS_1;
    G[1][1][2] = -2.0 * a2cs * S_1;
    G[1][2][2] = -r * D * S_1;
    G[1][3][3] = -D * s2 * (2. * a2c2 * r3 + r2 * r3 + a2 * a2c2 * s2 + a2c2 * a2c2 * r - a2 * r * (r - rQ) * s2) * S_3;
    G[2][0][0] = -(2.0 * r - rQ) * a2cs * S_3;
    G[2][0][3] = 2.0 * -G[2][0][0] * a2_r2 / a;
    G[2][1][1] = +a2cs * DS_1;
    G[2][1][2] = 2.0 * r * S_1;
    G[2][2][2] = -a2cs * S_1;
    G[2][3][3] = -cs * (a2_r2 * S * S + a2 * s2 * (rS - rQ) * (a2_r2 + S)) * S_3;
    G[3][0][1] = 2.0 * a * (r * (r - rQ) - a2c2) * DS_1 * S_1;
    G[3][0][2] = 2.0 * -4.0 * a * (rS - rQ) * m_s * R_1;
    G[3][1][3] = 2.0 * 4.0 * (r3 * (r2 - rS + rQ) + r * a2c2 * a2c2 -
                 a2 * r * (r - rQ) * s2 + a2c2 * r * (dbl_r2 - rS + rQ) + a2c2 * a2 * s2) * DR_1;
    G[3][2][3] = 2.0 * ((3. * a4 + 8. * a2 * r + 8. * a2 * r2 + 8. * r2 * r2 +
                 4. * (dbl_r2 - rS + rQ + a2) * a2cc + a4 * CC) * m_s) * (R_1 / 2.0);
}

/* Minkowski connection, upper-triangle storage with doubled off-diagonals;
 * ref: src/sim5kerr.c:199-229 */
void orc_flat_connection(double r, double m, double G[4][4][4])
{
    double s = sqrt(1. - m * m);
    memset(G, 0, 64 * sizeof(double));
    G[1][2][2] = -r;
    G[1][3][3] = -r * s * s;
    G[2][1][2] = 2.0 * 1. / r;
    G[2][3][3] = -m * s;
    G[3][1][3] = 2.0 * 1. / r;
    G[3][2][3] = 2.0 * m / s;
}

/* Kerr connection (20 non-zero entries, j<=k, off-diagonals pre-doubled);
 * ref: src/sim5kerr.c:233-316 */
void orc_kerr_connection(double a, double r, double m, double G[4][4][4])
{
    double rS = 2.0 * r;
    double s = sqrt(1. - m * m);
    double cs = s * m;
    double c2 = m * m;
    double s2 = s * s;
    double cc = c2 - s2;
    double CC = 8. * c2 * c2 - 8. * c2 + 1.;
    double a2 = a * a;
    double a4 = a2 * a2;
    double a2cc = a2 * cc;
    double a2c2 = a2 * c2;
    double a2cs = a2 * cs;
    double a4CC = a4 * CC;
    double r2 = r * r;
    double r3 = r2 * r;
    double r4 = r2 * r2;
    double a2r2 = a2 * r2;
    double a2_r2 = a2 + r2;
    double R = pow(a2 + 2. * r2 + a2cc, 2.);
    double D = r2 - 2. * r + a2;
    double S = r2 + a2c2;
    double S_1 = 1. / S;
    double S_3 = 1. / (S * S * S);
    double D_1 = 1. / D;
    double R_1 = 1. / R;
    double m_s = m / s;
    double DR_1 = D_1 * R_1;
    double DS_1 = D_1 * S_1;
    double dbl_r2 = 2. * r2;

    memset(G, 0, 64 * sizeof(double));

    G[0][0][1] = 2.0 * 4.0 * (a2_r2) * (r2 - a2c2) * DR_1;
    G[0][0][2] = 2.0 * -4.0 * a2cs * rS * R_1;
    G[0][1][3] = 2.0 * 2.0 * a * s2 * (a4 - 3. * a2r2 - 6. * r4 + a2cc * (a2 - r2)) * DR_1;
    G[0][2][3] = -G[0][0][2] * s2 * a;

    G[1][0][0] = D * (r2 - a2c2) * S_3;
    G[1][0][3] = -2.0 * G[1][0][0] * a * s2;
    G[1][1][1] = (r * (a2 - r) + a2 * (1. - r) * c2) * DS_1;
    G[1][1][2] = -2.0 * a2cs * S_1;
    G[1][2][2] = -r * D * S_1;
    G[1][3][3] = -D * s2 * (2. * a2c2 * r3 + r2 * r3 + a2 * a2c2 * s2 + a2c2 * a2c2 * r - a2r2 * s2) * S_3;

    G[2][0][0] = -2.0 * r * a2cs * S_3;
    G[2][0][3] = 2.0 * -G[2][0][0] * a2_r2 / a;
    G[2][1][1] = +a2cs * DS_1;
    G[2][1][2] = 2.0 * r * S_1;
    G[2][2][2] = -a2cs * S_1;
    G[2][3][3] = -cs * (a2_r2 * S * S + a2 * s2 * rS * (a2_r2 + S)) * S_3;

    G[3][0][1] = 2.0 * a * (r2 - a2c2) * DS_1 * S_1;
    G[3][0][2] = 2.0 * -4.0 * a * rS * m_s * R_1;
    G[3][1][3] = (a4 + 3. * a4 * r - 12. * a2r2 + 8. * a2 * r3 -
                  16. * r4 + 8. * r2 * r3 + 4. * r * (dbl_r2 - r + a2) * a2cc -
                  a4CC * (1. - r)) * DR_1;
    G[3][2][3] = ((3. * a4 + 8. * a2 * r + 8. * a2r2 + 8. * r4 +
                   4. * (dbl_r2 - 2. * r + a2) * a2cc + a4CC) * m_s) * R_1;
}

/* -G^i_jk U^j V^k with the half-weight for the doubled storage; ref: src/sim5kerr.c:422-439 */
void orc_Gamma(double G[4][4][4], double U[4], double V[4], double out[4])
{
    for (int i = 0; i < 4; i++) {
        out[i] = 0.0;
        for (int j = 0; j < 4; j++)
            for (int k = j; k < 4; k++)
                out[i] -= 0.5 * G[i][j][k] * (U[j] * V[k] + U[k] * V[j]);
    }
}

/* ref: src/sim5kerr.c:609-626 (NULL metric = Minkowski in Cartesian-like signature) */
double orc_dotprod(const double u[4], const double v[4], const orc_metric *g)
{
    if (g)
        return u[0] * v[0] * g->g00 + u[1] * v[1] * g->g11 + u[2] * v[2] * g->g22 +
               u[3] * v[3] * g->g33 + u[0] * v[3] * g->g03 + u[3] * v[0] * g->g03;
    return -u[0] * v[0] + u[1] * v[1] + u[2] * v[2] + u[3] * v[3];
}

/* ref: src/sim5kerr.c:553-573 (the factor is re-evaluated per component there;
 * sqrt and division are correctly rounded so the value is the same each time) */
void orc_vector_norm_to(double v[4], double norm, const orc_metric *g)
{
    double N = orc_dotprod(v, v, g);
    v[0] *= sqrt(norm / N);
    v[1] *= sqrt(norm / N);
    v[2] *= sqrt(norm / N);
    v[3] *= sqrt(norm / N);
}

/* ref: src/sim5kerr.c:678-711 */
void orc_tetrad_zamo(const orc_metric *g, orc_tetrad *t)
{
    memset(t->e, 0, sizeof(t->e));
    t->e[0][0] = sqrt(g->g33 / (SQ(g->g03) - g->g33 * g->g00));
    t->e[0][3] = -t->e[0][0] * g->g03 / g->g33;
    t->e[1][1] = 1. / sqrt(g->g11);
    t->e[2][2] = -1. / sqrt(g->g22);
    t->e[3][3] = 1. / sqrt(g->g33);
    t->metric = *g;
}

/* frame of a circular orbiter with angular velocity Omega; ref: src/sim5kerr.c:766-814 */
void orc_tetrad_azimuthal(const orc_metric *g, double Omega, orc_tetrad *t)
{
    if (Omega == 0.0) { orc_tetrad_zamo(g, t); return; }
    double g00 = g->g00, g33 = g->g33, g03 = g->g03;
    double U0 = sqrt(-1.0 / (g00 + 2. * Omega * g03 + SQ(Omega) * g33));
    double U3 = U0 * Omega;
    memset(t->e, 0, sizeof(t->e));
    t->e[0][0] = U0;
    t->e[0][3] = U3;
    t->e[1][1] = sqrt(1. / g->g11);
    t->e[2][2] = -sqrt(1. / g->g22);
    double k1 = (g03 * U3 + g00 * U0);
    double k2 = (g33 * U3 + g03 * U0);
    t->e[3][0] = -(k1 >= 0.0 ? +1.0 : -1.0) * k2 /
                 sqrt((g33 * g00 - g03 * g03) * (g00 * U0 * U0 + g33 * U3 * U3 + 2.0 * g03 * U0 * U3));
    t->e[3][3] = t->e[3][0] * (-k1 / k2);
    t->metric = *g;
}

/* frame comoving with a tilted, radially drifting surface element (Sadowski+2011 App. A);
 * ref: src/sim5kerr.c:818-921 */
void orc_tetrad_surface(const orc_metric *g, double Omega, double V, double dhdr, orc_tetrad *t)
{
    double g00 = g->g00, g11 = g->g11, g22 = g->g22, g33 = g->g33, g03 = g->g03;
    double S0r = 1.0 / sqrt(g11 + g22 * SQ(dhdr));
    double S0h = S0r * dhdr;
    double ur = V / sqrt(1. - V * V) / sqrt(g11);
    double v = (V >= 0.0 ? +1.0 : -1.0) *
               sqrt((SQ(ur / S0r) * (-g00 - 2. * Omega * g03 - SQ(Omega) * g33)) / (1. + SQ(ur / S0r)));

    t->e[0][0] = 1.0;
    t->e[0][1] = v * S0r;
    t->e[0][2] = v * S0h;
    t->e[0][3] = Omega;
    orc_vector_norm_to(t->e[0], -1.0, g);

    t->e[1][0] = (v * t->e[0][0]);
    t->e[1][1] = (v * t->e[0][1] + S0r / t->e[0][0]);
    t->e[1][2] = (v * t->e[0][2] + S0h / t->e[0][0]);
    t->e[1][3] = (v * t->e[0][3]);
    orc_vector_norm_to(t->e[1], 1.0, g);

    t->e[2][0] = 0.0;
    t->e[2][1] = dhdr;
    t->e[2][2] = -1.0;
    t->e[2][3] = 0.0;
    orc_vector_norm_to(t->e[2], 1.0, g);

    t->e[3][0] = -(g03 + g33 * Omega) / (g00 + g03 * Omega);
    t->e[3][1] = 0.0;
    t->e[3][2] = 0.0;
    t->e[3][3] = 1.0;
    orc_vector_norm_to(t->e[3], 1.0, g);

    t->metric = *g;
}

/* coordinate -> local components; ref: src/sim5kerr.c:926-944 */
void orc_bl2on(const double in[4], double out[4], const orc_tetrad *t)
{
    out[0] = -orc_dotprod(t->e[0], in, &t->metric);
    out[1] = +orc_dotprod(t->e[1], in, &t->metric);
    out[2] = +orc_dotprod(t->e[2], in, &t->metric);
    out[3] = +orc_dotprod(t->e[3], in, &t->metric);
}

/* local -> coordinate components; ref: src/sim5kerr.c:948-970 */
void orc_on2bl(const double in[4], double out[4], const orc_tetrad *t)
{
    for (int i = 0; i < 4; i++) {
        out[i] = 0.0;
        for (int j = 0; j < 4; j++) out[i] += in[j] * t->e[j][i];
    }
}

/* ref: src/sim5kerr.c:1037-1047 */
double orc_OmegaK(double r, double a) { return 1. / (a + pow(r, 1.5)); }

/* ref: src/sim5kerr.c:1050-1072 */
double orc_ellK(double r, double a)
{
    return (SQ(r) - 2. * a * sqrt(r) + SQ(a)) / (sqrt(r) * r - 2. * sqrt(r) + a);
}

/* ref: src/sim5kerr.c:1101-1111 */
double orc_Omega_from_ell(double ell, const orc_metric *g)
{
    return -(g->g03 + ell * g->g00) / (g->g33 + ell * g->g03);
}

/* redshift factor for Keplerian equatorial emitters; ref: src/sim5kerr.c:1128-1141 */
double orc_gfactorK(double r, double a, double l)
{
    double Om = 1. / (a + pow(r, 1.5));
    return sqrt(1. - 2. / r * pow(1. - a * Om, 2.) - (r * r + a * a) * SQ(Om)) / (1. - Om * l);
}

/* null 4-momentum from the constants of motion; ref: src/sim5kerr.c:1151-1213 */
void orc_photon_momentum(double a, double r, double m, double l, double q,
                         double r_sign, double m_sign, double k[4])
{
    double a2 = SQ(a), l2 = SQ(l), r2 = SQ(r), m2 = SQ(m);
    double S = r2 + a2 * m2;
    double D = r2 - 2. * r + a2;
    double R = SQ(r2 + a2 - a * l) - D * (SQ(l - a) + q);
    double M = q - l2 * m2 / (1. - m2) + a2 * m2;

    if ((M < 0.0) && (-M < 1e-8)) M = 0.0;
    if ((R < 0.0) && (-R < 1e-8)) R = 0.0;
    if (M < 0.0) { k[0] = k[1] = k[2] = k[3] = NAN; return; }

    k[0] = +1 / S * (-a * (a * (1. - m2) - l) + (r2 + a2) / D * (r2 + a2 - a * l));
    k[1] = +1 / S * sqrt(R);
    k[2] = +1 / S * sqrt(M);
    k[3] = +1 / S * (-a + l / (1. - m2) + a / D * (r2 + a2 - a * l));

    if (r_sign < 0.0) k[1] = -k[1];
    if (m_sign < 0.0) k[2] = -k[2];
}

/* ref: src/sim5kerr.c:1217-1251 */
void orc_photon_motion_constants(double a, double r, double m, const double k[4], double *L, double *Q)
{
    double a2 = SQ(a), r2 = SQ(r);
    double s2 = 1. - m * m;
    double D = r2 - 2. * r + a2;
    double l;
    double nf = k[3] / k[0];
    double nh = SQ(k[2]) / SQ(k[0]);
    *L = l = (-a * a2 + SQ(a2) * nf + nf * SQ(r2) + a * (D - r2) + a2 * nf * (2. * r2 - D * s2)) * s2 /
             (D - a * s2 * (a - a2 * nf + nf * (D - r2)));
    *Q = pow(a * (l - a * s2) + ((a2 + r2) * (a2 - a * l + r2)) / D, 2.0) *
         (nh - (SQ(D * m) * (SQ(l) - a2 * s2)) /
               (-s2 * pow(SQ(a2) - a * a2 * l + SQ(r2) + a * l * (D - r2) + a2 * (2. * r2 - D * s2), 2.0)));
}

/* ref: src/sim5kerr.c:1255-1269 */
double orc_photon_carter_const(const double k[4], const orc_metric *g)
{
    double m2 = SQ(g->m);
    double kt = k[0] * g->g00 + k[3] * g->g03;
    double kh = k[2] * g->g22;
    double kf = k[3] * g->g33 + k[0] * g->g03;
    return SQ(kh) + SQ(kf) * m2 / (1. - m2) - SQ(g->a) * SQ(kt) * m2;
}

/* ================================================================================== */
/*  Null geodesics by elliptic integrals                                              */
/* ================================================================================== */

/* T_int(x) = mK * cn^-1(x / sqrt(m2p) | mm); ref: src/sim5kerr-geod.c:29 */
static double pol_integral(const orc_geodesic *g, double x)
{
    return g->mK * orc_jacobi_icn(x / sqrt(g->m2p), g->mm);
}

/* inverse of the above; ref: src/sim5kerr-geod.c:30 */
static double pol_inverse(const orc_geodesic *g, double T)
{
    return sqrt(g->m2p) * orc_jacobi_cn(T / g->mK, g->mm);
}

/* Reorder the four quartic roots: real ones first, sorted descending, complex ones after
 * in their original order.  "Real" means imaginary part exactly zero.
 * ref: src/sim5polyroots.c:278-325 */
static int order_roots(orc_cplx z[4])
{
    orc_cplx out[4];
    int nreal = 0, k;
    for (int i = 0; i < 4; i++)
        if (cimag(z[i]) == 0.) out[nreal++] = z[i];
    k = nreal;
    for (int i = 0; i < 4; i++)
        if (cimag(z[i]) != 0.) out[k++] = z[i];
    for (int i = 0; i < nreal; i++)
        for (int j = 0; j < nreal - i; j++)
            if (creal(out[i + j]) > creal(out[i])) {
                orc_cplx t = out[i + j]; out[i + j] = out[i]; out[i] = t;
            }
    for (int i = 0; i < 4; i++) z[i] = out[i];
    return nreal;
}

/* Roots of R(r) in closed form (Cadez, Fanton & Calvani 1998), classification of the
 * radial motion and the radial integral from the turning point to infinity.
 * ref: src/sim5kerr-geod.c:986-1104 */
static int radial_roots(orc_geodesic *g, double r0, int *error)
{
    double a = g->a, l = g->l, q = g->q;
    double a2 = SQ(a), l2 = SQ(l);
    double A, B, C, D, E, F, X;

    C = SQ(a - l) + q;
    D = 2. / 3. * (q + l2 - a2);
    E = 9. / 4. * SQ(D) - 12. * a2 * q;
    F = -27. / 4. * (D * D * D) - 108. * a2 * q * D + 108. * SQ(C);
    X = SQ(F) - 4. * (E * E * E);
    if (X >= 0) {
        A = (F > sqrt(X) ? +1 : -1) * 1. / 3. * pow(fabs(F - sqrt(X)) / 2., 1. / 3.) +
            (F > -sqrt(X) ? +1 : -1) * 1. / 3. * pow(fabs(F + sqrt(X)) / 2., 1. / 3.);
    } else {
        double Z = sqrt(pow(F / 54., 2) + pow(sqrt(-X) / 54., 2));
        double z = atan2(sqrt(-X) / 54., F / 54.);
        A = pow(Z, 1. / 3.) * 2. * cos(z / 3.);
    }
    B = sqrt(A + D);
    orc_cplx z[4];
    z[0] = +B / 2. + .5 * csqrt(CMPLX(-A + 2. * D - 4. * C / B, 0.0));
    z[1] = +B / 2. - .5 * csqrt(CMPLX(-A + 2. * D - 4. * C / B, 0.0));
    z[2] = -B / 2. + .5 * csqrt(CMPLX(-A + 2. * D + 4. * C / B, 0.0));
    z[3] = -B / 2. - .5 * csqrt(CMPLX(-A + 2. * D + 4. * C / B, 0.0));
    g->nrr = order_roots(z);
    g->r1 = z[0]; g->r2 = z[1]; g->r3 = z[2]; g->r4 = z[3];

    switch (g->nrr) {
    case 4:
        g->type = ORC_RR;
        if ((r0 < creal(g->r3)) || ((r0 > creal(g->r2)) && (r0 < creal(g->r1)))) {
            if (error) *error = ORC_E_UNKNOWN;
            return 0;
        }
        if (fabs(creal(g->r1) - creal(g->r2)) < 1e-8) {
            g->type = ORC_RR_DBL;
            if (error) *error = ORC_E_RR_DOUBLE;
            return 0;
        }
        if ((r0 >= creal(g->r3)) && (r0 <= creal(g->r2))) g->type = ORC_RR_BH;
        break;
    case 2: g->type = ORC_RC; break;
    case 0: g->type = ORC_CC; break;
    default:
        if (error) *error = ORC_E_UNKNOWN;
        return 0;
    }

    double r1, r2, r3, r4, u, v, mm;
    switch (g->type) {
    case ORC_RR:
        r1 = creal(g->r1); r2 = creal(g->r2); r3 = creal(g->r3); r4 = creal(g->r4);
        mm = ((r2 - r3) * (r1 - r4)) / ((r2 - r4) * (r1 - r3));
        g->rp = r1;
        g->Rpc = 2. / sqrt((r1 - r3) * (r2 - r4)) * orc_jacobi_isn(sqrt((r2 - r4) / (r1 - r4)), mm);
        break;
    case ORC_RR_BH:
        r1 = creal(g->r1); r2 = creal(g->r2); r3 = creal(g->r3); r4 = creal(g->r4);
        mm = ((r2 - r3) * (r1 - r4)) / ((r2 - r4) * (r1 - r3));
        g->rp = r2;
        g->Rpc = 2. / sqrt((r1 - r3) * (r2 - r4)) * orc_elliptic_k(mm);
        break;
    case ORC_RC:
        r1 = creal(g->r1); r2 = creal(g->r2); u = creal(g->r3); v = cimag(g->r3);
        A = sqrt(SQ(r1 - u) + SQ(v));
        B = sqrt(SQ(r2 - u) + SQ(v));
        mm = (SQ(A + B) - SQ(r1 - r2)) / (4. * A * B);
        g->rp = r1;
        g->Rpc = 1. / sqrt(A * B) * orc_jacobi_icn((A - B) / (A + B), mm);
        break;
    case ORC_CC: {
        r1 = creal(g->r1); r2 = creal(g->r3); r3 = cimag(g->r1); r4 = cimag(g->r3);
        A = sqrt(SQ(r1 - r2) + SQ(r3 + r4));
        B = sqrt(SQ(r1 - r2) + SQ(r3 - r4));
        double g1 = sqrt((4. * SQ(r3) - SQ(A - B)) / (SQ(A + B) - 4. * SQ(r3)));
        mm = 4. * A * B / SQ(A + B);
        g->rp = r1 - r3 * g1;
        g->Rpc = 2. / (A + B) * orc_jacobi_itn(-1. / g1, mm);
        break;
    }
    default:
        return 0;
    }
    return 1;
}

/* Roots of the polar potential, written so that small spins do not cancel; the host
 * branch of the reference carries X in 80-bit long double (the sqrt itself is the
 * double one, its argument is narrowed).  ref: src/sim5kerr-geod.c:1110-1184 */
static int polar_roots(orc_geodesic *g, double m, int *error)
{
    double a = g->a, l = g->l, q = g->q;
    double a2 = SQ(a), l2 = SQ(l);

    long double qla = q + l2 - a2;
    long double X = sqrt(SQ(qla) + 4. * q * a2) + qla;
    long double dbla = a2 + a2;
    long double dblq = q + q;
    g->m2m = X / dbla;
    g->m2p = dblq / X;

    if ((g->m2p <= 0.0) || (g->m2p >= 1.0)) {
        if (error) *error = ORC_E_MUPLUS;
        return 0;
    }
    if (q > 0.0) {
        g->mm = g->m2p / (g->m2p + g->m2m);
        if ((g->mm < 0.0) || (g->mm >= 1.0)) { if (error) *error = ORC_E_MM; return 0; }
        if (fabs(m) > sqrt(g->m2p)) { if (error) *error = ORC_E_MU0; return 0; }
        g->mK = 1. / sqrt(a2 * (g->m2p + g->m2m));
    } else if (q < 0.0) {
        g->mm = (g->m2p + g->m2m) / g->m2p;
        if ((g->mm < 0.0) || (g->mm >= 1.0)) { if (error) *error = ORC_E_MM; return 0; }
        if ((fabs(m) > sqrt(g->m2p)) || (fabs(m) < sqrt(-g->m2m))) {
            if (error) *error = ORC_E_MU0;
            return 0;
        }
        g->mK = 1. / sqrt(a2 * g->m2p);
    } else {
        if (error) *error = ORC_E_Q_RANGE;
        return 0;
    }
    return 1;
}

/* geodesic from impact parameters at infinity; ref: src/sim5kerr-geod.c:42-100 */
int orc_geodesic_init_inf(double i, double a, double alpha, double beta, orc_geodesic *g, int *error)
{
    if ((a < 0.0) || (a > 1. - 1e-6)) { if (error) *error = ORC_E_SPIN; return 0; }
    if ((i <= 0.0) || (i >= 1.57079632679)) { if (error) *error = ORC_E_INCL; return 0; }
    if (beta == 0.0) beta = +1e-6;

    g->a = fmax(1e-4, a);
    g->incl = i;
    g->cos_i = cos(i);
    g->alpha = alpha;
    g->beta = beta;
    g->l = -alpha * sin(i);
    g->q = SQ(beta) + SQ(cos(i)) * (SQ(alpha) - SQ(a));      /* the caller's a, not the clamped one */
    if (g->q == 0.0) { if (error) *error = ORC_E_Q_RANGE; return 0; }

    if (!radial_roots(g, DBL_MAX, error)) return 0;
    if (!polar_roots(g, g->cos_i, error)) return 0;

    g->Tpp = 2. * pol_integral(g, 0.0);
    g->Tip = pol_integral(g, g->cos_i);
    if (error) *error = ORC_OK;
    return 1;
}

/* geodesic through a point with a given momentum; ref: src/sim5kerr-geod.c:106-173 */
int orc_geodesic_init_src(double a, double r, double m, double k[4], int ppc, orc_geodesic *g, int *error)
{
    double l, q;
    orc_photon_motion_constants(a, r, m, k, &l, &q);
    g->a = fmax(1e-8, a);
    g->l = l;
    g->q = q;
    g->cos_i = g->alpha = g->beta = NAN;

    if (!radial_roots(g, r, error)) return 0;
    if (!polar_roots(g, m, error)) return 0;

    if (isnan(g->cos_i) && (r > g->rp)) {
        double Tmp = pol_integral(g, m);
        double Tpp = 2. * pol_integral(g, 0.0);
        double T = orc_geodesic_P_int(g, r, ppc);
        double sdm = (k[2] < 0.0) ? +1.0 : -1.0;
        T += (sdm > 0.0) ? Tpp - Tmp : Tmp;
        while (T > Tpp) { T -= Tpp; sdm = -sdm; }
        g->cos_i = -sdm * pol_inverse(g, T);
        g->incl = acos(g->cos_i);
        g->alpha = -g->l / sqrt(1.0 - SQ(g->cos_i));
        g->beta = -sdm * sqrt(g->q - SQ(g->cos_i) * (SQ(g->alpha) - SQ(g->a)));
    }
    g->Tpp = 2. * pol_integral(g, 0.0);
    g->Tip = pol_integral(g, g->cos_i);
    if (error) *error = ORC_OK;
    return 1;
}

/* position integral from infinity down to radius r; ref: src/sim5kerr-geod.c:179-263 */
double orc_geodesic_P_int(const orc_geodesic *g, double r, int ppc)
{
    double r1, r2, r3, r4, u, v, mm, R, A, B;
    if (r == g->rp) return g->Rpc;
    switch (g->type) {
    case ORC_RR:
        r1 = creal(g->r1); r2 = creal(g->r2); r3 = creal(g->r3); r4 = creal(g->r4);
        mm = ((r2 - r3) * (r1 - r4)) / ((r2 - r4) * (r1 - r3));
        R = 2. / sqrt((r1 - r3) * (r2 - r4)) *
            orc_jacobi_isn(sqrt(((r2 - r4) * (r - r1)) / ((r1 - r4) * (r - r2))), mm);
        return (ppc) ? g->Rpc + R : g->Rpc - R;
    case ORC_RR_DBL:
        return NAN;
    case ORC_RR_BH:
        r1 = creal(g->r1); r2 = creal(g->r2); r3 = creal(g->r3); r4 = creal(g->r4);
        mm = ((r2 - r3) * (r1 - r4)) / ((r2 - r4) * (r1 - r3));
        R = 2. / sqrt((r1 - r3) * (r2 - r4)) *
            orc_jacobi_isn(sqrt((r1 - r3) / (r2 - r3) * (r2 - r) / (r1 - r)), mm);
        return (ppc) ? g->Rpc + R : g->Rpc - R;
    case ORC_RC:
        r1 = creal(g->r1); r2 = creal(g->r2); u = creal(g->r3); v = cimag(g->r3);
        A = sqrt(SQ(r1 - u) + SQ(v));
        B = sqrt(SQ(r2 - u) + SQ(v));
        mm = (SQ(A + B) - SQ(r1 - r2)) / (4. * A * B);
        R = 1. / sqrt(A * B) *
            orc_jacobi_icn(((A - B) * r + r1 * B - r2 * A) / ((A + B) * r - r1 * B - r2 * A), mm);
        return g->Rpc - R;
    case ORC_CC: {
        r1 = creal(g->r1); r2 = creal(g->r3); r3 = cimag(g->r1); r4 = cimag(g->r3);
        A = sqrt(SQ(r1 - r2) + SQ(r3 + r4));
        B = sqrt(SQ(r1 - r2) + SQ(r3 - r4));
        double g1 = sqrt((4. * SQ(r3) - SQ(A - B)) / (SQ(A + B) - 4. * SQ(r3)));
        mm = 4. * A * B / SQ(A + B);
        R = 2. / (A + B) * orc_jacobi_itn((r - r1 + r3 * g1) / (r3 + r1 * g1 - g1 * r), mm);
        return g->Rpc - R;
    }
    }
    return NAN;
}

/* r(P); ref: src/sim5kerr-geod.c:291-357 */
double orc_geodesic_position_rad(const orc_geodesic *g, double P)
{
    double r1, r2, r3, r4, u, v;
    if ((P <= 0.0) || (P >= 2. * g->Rpc)) return NAN;
    if (P == g->Rpc) return g->rp;
    switch (g->type) {
    case ORC_RR: {
        r1 = creal(g->r1); r2 = creal(g->r2); r3 = creal(g->r3); r4 = creal(g->r4);
        double m4 = ((r2 - r3) * (r1 - r4)) / ((r2 - r4) * (r1 - r3));
        double x4 = 0.5 * fabs(P - g->Rpc) * sqrt((r2 - r4) * (r1 - r3));
        double sn2 = pow(orc_jacobi_sn(x4, m4), 2.0);
        return (r1 * (r2 - r4) - r2 * (r1 - r4) * sn2) / (r2 - r4 - (r1 - r4) * sn2);
    }
    case ORC_RC: {
        if (P > g->Rpc) return NAN;
        r1 = creal(g->r1); r2 = creal(g->r2); u = creal(g->r3); v = cimag(g->r3);
        double A = sqrt(SQ(r1 - u) + SQ(v));
        double B = sqrt(SQ(r2 - u) + SQ(v));
        double m2 = (SQ(A + B) - SQ(r1 - r2)) / (4. * A * B);
        double cn = orc_jacobi_cn(sqrt(A * B) * (g->Rpc - P), m2);
        return (r2 * A - r1 * B - (r2 * A + r1 * B) * cn) / ((A - B) - (A + B) * cn);
    }
    default:
        return NAN;      /* RR_DBL, RR_BH, CC: not available (:322-352) */
    }
}

/* sign of d(cos theta)/dP at P, and the polar phase it belongs to;
 * ref: src/sim5kerr-geod.c:737-781 (same walk used at :363-407) */
static double polar_phase(const orc_geodesic *g, double P, double *T_out)
{
    double sdm = (g->beta >= 0.0) ? +1.0 : -1.0;
    double T = (sdm > 0.0) ? -(g->Tpp - g->Tip) : -(g->Tip);
    while (P > T + g->Tpp) { T += g->Tpp; sdm = -sdm; }
    if (T_out) *T_out = T;
    return sdm;
}

/* cos(theta)(P); ref: src/sim5kerr-geod.c:363-407 */
double orc_geodesic_position_pol(const orc_geodesic *g, double P)
{
    if (g->type == ORC_RR || g->type == ORC_RC || g->type == ORC_CC) {
        double T, sdm = polar_phase(g, P, &T);
        return -sdm * pol_inverse(g, P - T);
    }
    return NAN;
}

/* ref: src/sim5kerr-geod.c:737-781 */
double orc_geodesic_dm_sign(const orc_geodesic *g, double P)
{
    if (g->type == ORC_RR || g->type == ORC_RC || g->type == ORC_CC) return polar_phase(g, P, 0);
    return NAN;
}

/* photon 4-momentum at P (pointing along increasing P); ref: src/sim5kerr-geod.c:787-840 */
void orc_geodesic_momentum(const orc_geodesic *g, double P, double r, double m, double k[4])
{
    if ((r == 0.0) && (m == 0.0)) {
        r = orc_geodesic_position_rad(g, P);
        m = orc_geodesic_position_pol(g, P);
    }
    if (g->type == ORC_RR || g->type == ORC_RC || g->type == ORC_CC) {
        double dm = orc_geodesic_dm_sign(g, P);
        orc_photon_momentum(g->a, r, m, g->l, g->q, (P < g->Rpc ? -1 : +1), dm, k);
        return;
    }
    if (g->type == ORC_RR_DBL || g->type == ORC_RR_BH) k[0] = k[1] = k[2] = k[3] = NAN;
}

/* clamp with slack; ref: src/sim5math.c:50-58 */
static int clamp_with_slack(double *val, double lo, double hi, double acc)
{
    if (*val < lo - acc) return 0;
    if (*val > hi + acc) return 0;
    if (*val < lo) *val = lo;
    if (*val > hi) *val = hi;
    return 1;
}

/* value of the position integral at the order-th crossing of the equatorial plane;
 * ref: src/sim5kerr-geod.c:846-885 */
double orc_geodesic_find_midplane_crossing(const orc_geodesic *g, int order)
{
    if (g->q <= 0.0) return NAN;
    double u = g->cos_i / sqrt(g->m2p);
    if (!clamp_with_slack(&u, -1.0, +1.0, 1e-4)) return NAN;
    double pos;
    if (g->beta > 0.0)
        pos = g->mK * ((2. * (double)order + 1.) * orc_elliptic_k(g->mm) + orc_jacobi_icn(u, g->mm));
    else if (g->beta < 0.0)
        pos = g->mK * ((2. * (double)order + 1.) * orc_elliptic_k(g->mm) - orc_jacobi_icn(u, g->mm));
    else
        pos = g->mK * ((2. * (double)order + 1.) * orc_elliptic_k(g->mm));
    if (pos > 2. * g->Rpc) pos = NAN;
    return pos;
}

/* advance along the geodesic by a proper-length-like step; ref: src/sim5kerr-geod.c:891-925 */
void orc_geodesic_follow(const orc_geodesic *g, double step, double *P, double *r, double *m, int *status)
{
    const double cap = 5e-2;
    do {
        double truestep = step / fabs(step) * fmin(fabs(step), cap * sqrt(*r));
        (*P) = (*P) + truestep / (SQ(*r) + SQ((g->a) * (*m)));
        (*r) = orc_geodesic_position_rad(g, *P);
        (*m) = orc_geodesic_position_pol(g, *P);
        if ((*r) < 1.01 * orc_r_bh(g->a)) { if (status) *status = 0; return; }
        if ((*P < 0.0) || (*P > 2. * g->Rpc)) { if (status) *status = 0; return; }
        step -= truestep;
    } while (fabs(step) > 1e-5);
    if (status) *status = 1;
}

/* change of azimuth between infinity and the point (r, m) at position integral P;
   ref: src/sim5kerr-geod.c:463-556 */
double orc_geodesic_position_azm(const orc_geodesic *g, double r, double m, double P)
{
    double phi = 0.0;
    int ppc = (g->nrr > 0) && (P > g->Rpc);
    double a2 = SQ(g->a);
    double rp = 1. + sqrt(1. - a2);
    double rm = 1. - sqrt(1. - a2);
    double r1, r2, r3, r4, A, B;

    if (g->type == ORC_RR) {
        r1 = creal(g->r1); r2 = creal(g->r2); r3 = creal(g->r3); r4 = creal(g->r4);
        A = orc_integral_R_rp_re_inf(r1, r2, r3, r4, rp) + (ppc ? +1 : -1) * orc_integral_R_rp_re(r1, r2, r3, r4, rp, r);
        B = orc_integral_R_rp_re_inf(r1, r2, r3, r4, rm) + (ppc ? +1 : -1) * orc_integral_R_rp_re(r1, r2, r3, r4, rm, r);
        phi += 1. / sqrt(1. - a2) * (A * (g->a * rp - g->l * a2 / 2.) - B * (g->a * rm - g->l * a2 / 2.));
    } else if (g->type == ORC_RC) {
        r1 = creal(g->r1); r2 = creal(g->r2);
        A = orc_integral_R_rp_cc2_inf(r1, r2, g->r3, rp, r);
        B = orc_integral_R_rp_cc2_inf(r1, r2, g->r3, rm, r);
        phi += 1. / sqrt(1. - a2) * (A * (g->a * rp - g->l * a2 / 2.) - B * (g->a * rm - g->l * a2 / 2.));
    } else if (g->type == ORC_RR_DBL || g->type == ORC_RR_BH || g->type == ORC_CC) {
        return NAN;
    }

    /* polar part */
    double phi_pp = 2.0 * g->l / g->a * orc_integral_T_mp(g->m2m, g->m2p, 1.0, 0.0);
    double phi_ip = g->l / g->a * orc_integral_T_mp(g->m2m, g->m2p, 1.0, g->cos_i);
    double phi_mp = g->l / g->a * orc_integral_T_mp(g->m2m, g->m2p, 1.0, m);

    double T;
    double sign_dm = (g->beta >= 0.0) ? +1.0 : -1.0;
    if (sign_dm > 0.0) {
        T = -(g->Tpp - g->Tip);
        phi -= phi_pp - phi_ip;
    } else {
        T = -g->Tip;
        phi -= phi_ip;
    }
    if (P >= T + g->Tpp) {                      /* the reference's while-loop leaves after one pass (:545-550) */
        T += g->Tpp;
        phi += phi_pp;
        sign_dm = -sign_dm;
    }
    phi += (sign_dm < 0) ? phi_mp : phi_pp - phi_mp;
    return phi;
}

/* light-travel time between two points of a geodesic (radial part, as the reference: the polar
   part is commented out there); ref: src/sim5kerr-geod.c:560-664 */
double orc_geodesic_timedelay(const orc_geodesic *g, double P1, double r1, double m1, double P2, double r2, double m2)
{
    double time = 0.0;
    if (P1 > P2) {
        double tmp;
        tmp = P2; P2 = P1; P1 = tmp;
        tmp = r2; r2 = r1; r1 = tmp;
        tmp = m2; m2 = m1; m1 = tmp;
    }
    if (r1 == 0) { r1 = orc_geodesic_position_rad(g, P1); m1 = orc_geodesic_position_pol(g, P1); }
    if (r2 == 0) { r2 = orc_geodesic_position_rad(g, P2); m2 = orc_geodesic_position_pol(g, P2); }
    (void)m1; (void)m2;

    double a2 = SQ(g->a);
    double rp = 1. + sqrt(1. - a2);
    double rm = 1. - sqrt(1. - a2);
    double ra = creal(g->r1), rb = creal(g->r2), rc = creal(g->r3), rd = creal(g->r4);
    double R0, R1, R2, RA, RB, A, B, s;

    if (g->type == ORC_RR) {
        s = (((P1 > g->Rpc) && (P2 < g->Rpc)) || ((P1 < g->Rpc) && (P2 > g->Rpc))) ? +1 : -1;
        R0 = orc_integral_R_r0_re(ra, rb, rc, rd, r1) + s * orc_integral_R_r0_re(ra, rb, rc, rd, r2);
        R1 = orc_integral_R_r1_re(ra, rb, rc, rd, r1) + s * orc_integral_R_r1_re(ra, rb, rc, rd, r2);
        R2 = orc_integral_R_r2_re(ra, rb, rc, rd, r1) + s * orc_integral_R_r2_re(ra, rb, rc, rd, r2);
        RA = orc_integral_R_rp_re(ra, rb, rc, rd, rp, r1) + s * orc_integral_R_rp_re(ra, rb, rc, rd, rp, r2);
        RB = orc_integral_R_rp_re(ra, rb, rc, rd, rm, r1) + s * orc_integral_R_rp_re(ra, rb, rc, rd, rm, r2);
    } else if (g->type == ORC_RC) {
        R0 = orc_integral_R_r0_cc(ra, rb, g->r3, r1) - orc_integral_R_r0_cc(ra, rb, g->r3, r2);
        R1 = (r1 < r2) ? orc_integral_R_r1_cc(ra, rb, g->r3, r1, r2) : orc_integral_R_r1_cc(ra, rb, g->r3, r2, r1);
        R2 = (r1 < r2) ? orc_integral_R_r2_cc(ra, rb, g->r3, r1, r2) : orc_integral_R_r2_cc(ra, rb, g->r3, r2, r1);
        RA = (r1 < r2) ? orc_integral_R_rp_cc2(ra, rb, g->r3, rp, r1, r2) : orc_integral_R_rp_cc2(ra, rb, g->r3, rp, r2, r1);
        RB = (r1 < r2) ? orc_integral_R_rp_cc2(ra, rb, g->r3, rm, r1, r2) : orc_integral_R_rp_cc2(ra, rb, g->r3, rm, r2, r1);
    } else if (g->type == ORC_RR_DBL || g->type == ORC_RR_BH || g->type == ORC_CC) {
        return NAN;
    } else {
        return time;
    }
    A = (-g->a * g->l + 4.) * rp - 2. * a2;
    B = (+g->a * g->l - 4.) * rm + 2. * a2;
    time += 4. * fabs(R0) + 2. * fabs(R1) + fabs(R2) + (A * fabs(RA) + B * fabs(RB)) / sqrt(1. - a2);
    return time;
}

/* ================================================================================== */
/*  Novikov-Thorne disk                                                               */
/* ================================================================================== */

/* ref: src/sim5disk-nt.c:91-105 (spin already rounded to float; pow, not cbrt) */
double orc_disk_nt_r_min(const orc_disk_nt *d)
{
    double a = d->spin;
    double sga = (a >= 0.0) ? +1. : -1.;
    double z1 = 1. + pow(1. - a * a, 1. / 3.) * (pow(1. + a, 1. / 3.) + pow(1. - a, 1. / 3.));
    double z2 = sqrt(3. * a * a + z1 * z1);
    double r0 = 3. + z2 - sga * sqrt((3. - z1) * (3. + z1 + 2. * z2));
    return r0 + 1e-3;
}

/* mdot-parametrised set-up only (the luminosity option needs the Simpson/bisection
 * helpers that are outside the hot path); ref: src/sim5disk-nt.c:37-78 */
void orc_disk_nt_setup(orc_disk_nt *d, double M, double a, double mdot, double alpha)
{
    d->mass = M;
    d->spin = a;
    d->rms = orc_disk_nt_r_min(d);
    d->alpha = alpha;
    d->options = 0;
    d->mdot = mdot;
}

/* Page & Thorne (1974) flux, eq. 15n; ref: src/sim5disk-nt.c:110-146 */
double orc_disk_nt_flux(const orc_disk_nt *d, double r)
{
    if (r <= d->rms) return 0.0;
    double a = d->spin;
    double x = sqrt(r);
    double x0 = sqrt(d->rms);
    double x1 = +2. * cos(1. / 3. * acos(a) - M_PI / 3.);
    double x2 = +2. * cos(1. / 3. * acos(a) + M_PI / 3.);
    double x3 = -2. * cos(1. / 3. * acos(a));
    double f0 = x - x0 - 1.5 * a * log(x / x0);
    double f1 = 3. * SQ(x1 - a) / (x1 * (x1 - x2) * (x1 - x3)) * log((x - x1) / (x0 - x1));
    double f2 = 3. * SQ(x2 - a) / (x2 * (x2 - x1) * (x2 - x3)) * log((x - x2) / (x0 - x2));
    double f3 = 3. * SQ(x3 - a) / (x3 * (x3 - x1) * (x3 - x2)) * log((x - x3) / (x0 - x3));
    double F = 1. / (4. * M_PI * r) * 1.5 / (x * x * (x * x * x - 3. * x + 2. * a)) * (f0 - f1 - f2 - f3);
    return 9.1721376255e+28 * F * d->mdot / d->mass;
}

/* ref: src/sim5disk-nt.c:260-266 */
double orc_disk_nt_ell(const orc_disk_nt *d, double r)
{
    double a = d->spin;
    r = fmax(d->rms, r);
    return (r * r - 2. * a * sqrt(r) + a * a) / (sqrt(r) * r - 2. * sqrt(r) + a);
}

/* ref: src/sim5disk-nt.c:193-199 */
double orc_disk_nt_mdot(const orc_disk_nt *d) { return d->mdot; }

/* Column density of the two inner zones; ref: src/sim5disk-nt.c:204-250.  Note 3.*(x1-a)*(x1-a)/... here
 * against 3.*sqr(x1-a)/... in the flux: the products associate differently. */
double orc_disk_nt_sigma(const orc_disk_nt *d, double r)
{
    if (r < d->rms) return 0.0;
    double a = d->spin;
    double x = sqrt(r);
    double x0 = sqrt(d->rms);
    double x1 = +2. * cos(1. / 3. * acos(a) - M_PI / 3.);
    double x2 = +2. * cos(1. / 3. * acos(a) + M_PI / 3.);
    double x3 = -2. * cos(1. / 3. * acos(a));
    double xA = 1. + SQ(a) / SQ(r) + 2. * SQ(a) / (r * r * r);
    double xB = 1. + a / (x * x * x);
    double xC = 1. - 3. / (x * x) + 2. * a / (x * x * x);
    double xD = 1. - 2. / r + SQ(a) / SQ(r);
    double xE = 1. + 4. * SQ(a) / SQ(r) - 4. * SQ(a) / (r * r * r) + 3. * (a * a * a * a) / (r * r * r * r);
    double f0 = x - x0 - 1.5 * a * log(x / x0);
    double f1 = 3. * (x1 - a) * (x1 - a) / (x1 * (x1 - x2) * (x1 - x3)) * log((x - x1) / (x0 - x1));
    double f2 = 3. * (x2 - a) * (x2 - a) / (x2 * (x2 - x1) * (x2 - x3)) * log((x - x2) / (x0 - x2));
    double f3 = 3. * (x3 - a) * (x3 - a) / (x3 * (x3 - x2) * (x3 - x1)) * log((x - x3) / (x0 - x3));
    double xL = (1. + a / (x * x * x)) / sqrt(1. - 3. / (x * x) + 2. * a / (x * x * x)) / x * (f0 - f1 - f2 - f3);
    const double Mdot_Edd = 2.225475942e+18;                       /* ref: src/sim5const.h:49 */
    double xMdot = d->mdot * d->mass * Mdot_Edd / 1e17;
    double r_im = 40. * (pow(d->alpha, 2. / 21.) / pow(d->mass / 3., 2. / 3.) * pow(xMdot, 16. / 20.)) * pow(xA, 20. / 21.) *
                  pow(xB, -36. / 21.) * pow(xD, -8. / 21.) * pow(xE, -10. / 21.) * pow(xL, 16. / 21.);
    double Sigma;
    if (r < r_im)
        Sigma = 20. * (d->mass / 3.) / xMdot / d->alpha * sqrt(r * r * r) * 1. / (xA * xA) * pow(xB, 3.) * sqrt(xC) * xE * 1. / xL;
    else
        Sigma = 5e4 * pow(d->mass / 3., -2. / 5.) * pow(xMdot, 3. / 5.) * pow(d->alpha, -4. / 5.) * pow(r, -3. / 5.) *
                pow(xB, -4. / 5.) * sqrt(xC) * pow(xD, -4. / 5.) * pow(xL, 3. / 5.);
    return Sigma;
}

/* integrand of the luminosity integral over log r; ref: src/sim5disk-nt.c:167-179 */
static double lumi_integrand(const orc_disk_nt *d, double log_r)
{
    double a = d->spin;
    double r = exp(log_r);
    double gtt = -1. + 2. / r;
    double gtf = -2. * a / r;
    /* sqr(disk_nt_bh_spin) on the float static is a FLOAT product (no promotion), then widened */
    double a2f = (double)(d->spin * d->spin);
    double gff = SQ(r) + a2f + 2. * a2f / r;
    double Omega = 1. / (a + pow(r, 1.5));
    double U_t = sqrt(-1.0 / (gtt + 2. * Omega * gtf + SQ(Omega) * gff)) * (gtt + Omega * gtf);
    double F = orc_disk_nt_flux(d, r);
    return 2. * M_PI * r * 2.0 * (-U_t) * F * r;
}

/* Total luminosity in Eddington units: Simpson rule built on the refined trapezoid rule
 * (ref: src/sim5disk-nt.c:151-188; src/sim5integration.c:26-52 stage rule with its running abscissa
 * x += del, :96-133 Simpson with NMAX 23, accuracy 1e-5, at least 4 stages). */
double orc_disk_nt_lumi(const orc_disk_nt *d)
{
    const float disk_rmax = 1e5;
    const double lo = log(d->rms), hi = log(disk_rmax), acc = 1e-5;
    double s = 0.0, st = 0.0, ost = -1.e50, os = -1.e50;
    int n;
    for (n = 1; n <= 23; n++) {
        if (n == 1) {
            st = 0.5 * (hi - lo) * (lumi_integrand(d, hi) + lumi_integrand(d, lo));
        } else {
            int it = 1, j;
            for (j = 1; j < n - 1; j++) it <<= 1;
            double tnm = (double)it, del = (hi - lo) / tnm, x = lo + 0.5 * del, sum = 0.0;
            for (j = 1; j <= it; j++, x += del) sum += lumi_integrand(d, x);
            st = 0.5 * (st + del * sum);
        }
        s = (4. * st - ost) / 3.;
        if (n > 3) {
            if ((fabs(s - os) < acc * fabs(os)) || ((s == 0.) && (os == 0.))) break;
        }
        os = s;
        ost = st;
    }
    const double grav_radius = 1.476716e+05, L_Edd = 1.257142540e+38;   /* ref: src/sim5const.h:32,51 */
    double L = s * SQ(d->mass * grav_radius);
    return L / (L_Edd * d->mass);
}

/* set-up with the reference's options word: bit 0 (DISK_NT_OPTION_LUMINOSITY, ref src/sim5disk-nt.h:17) reads
 * mdot_or_L as a luminosity and finds the accretion rate by bisection on [0, 100] to 1e-6
 * (ref: src/sim5disk-nt.c:37-78, :371-385; src/sim5roots.c:21-63).  The trial rate goes through the float static. */
void orc_disk_nt_setup_opt(orc_disk_nt *d, double M, double a, double mdot_or_L, double alpha, int options)
{
    d->mass = M;
    d->spin = a;
    d->rms = orc_disk_nt_r_min(d);
    d->alpha = alpha;
    d->options = options;
    if (!(options & 1)) { d->mdot = mdot_or_L; return; }
    const double L0 = mdot_or_L, x1 = 0.0, x2 = 100.0, xacc = 1e-6;
    double dx, f, fmid, xmid, rtb;
    long j;
    d->mdot = x2; fmid = L0 - orc_disk_nt_lumi(d);
    d->mdot = x1; f = L0 - orc_disk_nt_lumi(d);
    if ((f * fmid) >= 0.0) { d->mdot = 0.0; return; }
    if (f < 0.0) { rtb = x1; dx = x2 - x1; } else { rtb = x2; dx = x1 - x2; }
    for (j = 0; j < 500; j++) {
        dx = dx * 0.5;
        xmid = rtb + dx;
        d->mdot = xmid; fmid = L0 - orc_disk_nt_lumi(d);
        if (fmid <= 0.0) rtb = xmid;
        if ((fabs(dx) < xacc) || (fmid == 0.0)) break;
    }
    d->mdot = (j >= 500) ? 0.0 : rtb;
}

/* ================================================================================== */
/*  Step-wise integrator (velocity Verlet after Dolence+2009, RK4 fallback)            */
/* ================================================================================== */

#define REL_DIFF(a, b) (fabs((b) - (a)) / (fabs(b) + 1e-40))
#define RT_MAX_ERR 1e-2
#define RT_TINY 1e-40

/* ref: src/sim5raytrace.c:44-94 */
void orc_raytrace_prepare(double bh_spin, double x[4], double k[4], double precision, int options,
                          orc_raytrace_data *rtd)
{
    orc_metric g;
    double G[4][4][4];
    rtd->opt_gr = !((options & 1) == 1);
    rtd->step_epsilon = sqrt(precision) / 10.;
    if (rtd->opt_gr) {
        orc_kerr_metric(bh_spin, x[1], x[2], &g);
        orc_kerr_connection(bh_spin, x[1], x[2], G);
    } else {
        orc_flat_metric(x[1], x[2], &g);
        orc_flat_connection(x[1], x[2], G);
    }
    rtd->bh_spin = bh_spin;
    rtd->E = k[0] * g.g00 + k[3] * g.g03;
    rtd->Q = orc_photon_carter_const(k, &g);
    rtd->pass = 0;
    rtd->refines = 0;
    rtd->kt = rtd->E;
    rtd->error = 0.0;
    orc_Gamma(G, k, k, rtd->dk);
}

/* -G^j_ab k^a k^b over the stored upper triangle; ref: src/sim5raytrace.c:151-156 */
static double accel(double G[4][4][4], int j, const double k[4])
{
    double s = 0.0;
    for (int a = 0; a < 4; a++)
        for (int b = a; b < 4; b++) s -= G[j][a][b] * k[a] * k[b];
    return s;
}

/* classical RK4 step on (x,k) with theta as the angle; ref: src/sim5raytrace.c:251-323 */
static void rk4_step(double x[4], double k[4], double dl, orc_raytrace_data *rtd)
{
    orc_metric g;
    double G[4][4][4], xp[4];
    double k1[4], d1[4], k2[4], d2[4], k3[4], d3[4], k4[4], d4[4];
    double h = 0.5 * dl;
    double kt0 = rtd->kt;
    int i;

    x[2] = acos(x[2]);
#define CONN(rr, th) do { if (rtd->opt_gr) orc_kerr_connection(rtd->bh_spin, rr, cos(th), G); \
                          else orc_flat_connection(rr, cos(th), G); } while (0)
    for (i = 0; i < 4; i++) xp[i] = x[i];
    CONN(xp[1], xp[2]);
    for (i = 0; i < 4; i++) k1[i] = k[i];
    orc_Gamma(G, k1, k1, d1);

    for (i = 0; i < 4; i++) xp[i] = x[i] + k1[i] * h;
    CONN(xp[1], xp[2]);
    for (i = 0; i < 4; i++) k2[i] = k[i] + d1[i] * h;
    orc_Gamma(G, k2, k2, d2);

    for (i = 0; i < 4; i++) xp[i] = x[i] + k2[i] * h;
    CONN(xp[1], xp[2]);
    for (i = 0; i < 4; i++) k3[i] = k[i] + d2[i] * h;
    orc_Gamma(G, k3, k3, d3);

    for (i = 0; i < 4; i++) xp[i] = x[i] + k3[i] * dl;
    CONN(xp[1], xp[2]);
    for (i = 0; i < 4; i++) k4[i] = k[i] + d3[i] * dl;
    orc_Gamma(G, k4, k4, d4);
#undef CONN

    for (i = 0; i < 4; i++) {
        x[i] += dl / 6. * (k1[i] + 2. * k2[i] + 2. * k3[i] + k4[i]);
        k[i] += dl / 6. * (d1[i] + 2. * d2[i] + 2. * d3[i] + d4[i]);
    }
    x[2] = cos(x[2]);

    if (rtd->opt_gr) orc_kerr_connection(rtd->bh_spin, x[1], x[2], G);
    else orc_flat_connection(x[1], x[2], G);
    orc_Gamma(G, k, k, rtd->dk);

    orc_kerr_metric(rtd->bh_spin, x[1], x[2], &g);     /* Kerr metric even when flat (:302) */
    double kt1 = k[0] * g.g00 + k[3] * g.g03;
    rtd->error = REL_DIFF(kt1, kt0);
}

/* one adaptive step; the two error accumulators are single precision as in the
 * reference (:140, src/sim5raytrace.h:42); ref: src/sim5raytrace.c:109-245 */
void orc_raytrace(double x[4], double k[4], double *step, orc_raytrace_data *rtd)
{
    orc_metric g;
    double G[4][4][4];
    double *dk = rtd->dk;
    double x0[4], k0[4], xp[4], kp[4], kq[4];
    double kk, kt;
    float kerr;
    int i;

    for (i = 0; i < 4; i++) { x0[i] = x[i]; k0[i] = k[i]; }

    double stepsize = rtd->step_epsilon /
        (fabs(dk[0]) / (fabs(k[0]) + RT_TINY) + fabs(dk[1]) / (fabs(k[1]) + RT_TINY) +
         fabs(dk[2]) / (fabs(k[2]) + RT_TINY) + fabs(dk[3]) / (fabs(k[3]) + RT_TINY) + RT_TINY);
    double dl = fmin(*step, stepsize);
    if (dl < 1e-3) dl = 1e-3;

    rtd->pass++;

    double half_dl = 0.5 * dl;
    double half_dl2 = 0.5 * dl * dl;
    xp[0] = x[0] + k[0] * dl + dk[0] * half_dl2;
    xp[1] = x[1] + k[1] * dl + dk[1] * half_dl2;
    xp[2] = cos(acos(x[2]) + (k[2] * dl + dk[2] * half_dl2));
    xp[3] = x[3] + k[3] * dl + dk[3] * half_dl2;

    for (i = 0; i < 4; i++) k[i] += dk[i] * half_dl;

    if (rtd->opt_gr) {
        orc_kerr_metric(rtd->bh_spin, xp[1], xp[2], &g);
        orc_kerr_connection(rtd->bh_spin, xp[1], xp[2], G);
    } else {
        orc_flat_metric(xp[1], xp[2], &g);
        orc_flat_connection(xp[1], xp[2], G);
    }

    for (i = 0; i < 4; i++) kp[i] = k[i] + dk[i] * half_dl;

    int iter = 0;
    do {
        kerr = 0.0;
        for (i = 0; i < 4; i++) kq[i] = kp[i];
        for (i = 0; i < 4; i++) {
            kp[i] = k[i] + accel(G, i, kq) * half_dl;
            kerr += REL_DIFF(kp[i], kq[i]);
        }
        iter++;
    } while (kerr > RT_MAX_ERR * 1e-3 && iter < 3);

    kt = kp[0] * g.g00 + kp[3] * g.g03;
    kk = fabs(orc_dotprod(kp, kp, &g));
    rtd->error = fmax(REL_DIFF(kt, rtd->kt), kk);
    if ((kerr > RT_MAX_ERR * 1e-2) || (rtd->error > RT_MAX_ERR * 1e-2)) {
        for (i = 0; i < 4; i++) { x[i] = x0[i]; k[i] = k0[i]; }
        rk4_step(x, k, dl, rtd);
        *step = dl;
        return;
    }

    for (i = 0; i < 4; i++) {
        x[i] = xp[i];
        k[i] = kp[i];
        dk[i] = accel(G, i, kp);
    }
    rtd->kt = kt;
    *step = dl;
}

/* relative drift of Carter's constant; ref: src/sim5raytrace.c:328-343 */
double orc_raytrace_error(double x[4], double k[4], orc_raytrace_data *rtd)
{
    orc_metric g;
    if (rtd->opt_gr) orc_kerr_metric(rtd->bh_spin, x[1], x[2], &g);
    else orc_flat_metric(x[1], x[2], &g);
    return REL_DIFF(rtd->Q, orc_photon_carter_const(k, &g));
}

/* ================================================================================== */
/*  Polarization (Walker-Penrose) and radiation                                       */
/* ================================================================================== */

/* kappa = (A1 - i A2)(r - i a cos theta), Connors, Piran & Stark 1980;
 * ref: src/sim5polarization.c:145-158 */
orc_cplx orc_polarization_constant(const double k[4], const double f[4], const orc_metric *g)
{
    double a = g->a, m = g->m, r = g->r;
    double A1 = (k[0] * f[1] - k[1] * f[0]) + a * (1. - m * m) * (k[1] * f[3] - k[3] * f[1]);
    double A2 = sqrt(1. - m * m) * ((r * r + a * a) * (k[3] * f[2] - k[2] * f[3]) - a * (k[0] * f[2] - k[2] * f[0]));
    double wp1 = +r * A1 - a * m * A2;
    double wp2 = -r * A2 - a * m * A1;
    return wp1 + _Complex_I * wp2;
}

/* polarization vector with f^t = 0 from (k, kappa); ref: src/sim5polarization.c:55-105 */
void orc_polarization_vector(const double k[4], orc_cplx wp, const orc_metric *g, double f[4])
{
    double a = g->a, m = g->m, r = g->r;
    double s = sqrt(1.0 - m * m);
    double ra2 = r * r + a * a;
    double r2 = r * r;
    double a2 = a * a;
    double s2 = 1.0 - m * m;
    if (s < 1e-12) {
        s = 1e-12;
        s2 = 1e-24;
        m = 1.0 - 0.5 * s;
    }
    double A1 = (+r * creal(wp) - a * m * cimag(wp)) / (r * r + a * a * m * m);
    double A2 = (-r * cimag(wp) - a * m * creal(wp)) / (r * r + a * a * m * m);

    f[0] = 0.0;
    f[3] = (
             + g->g11 * A1 * k[1] * (s * r2 * k[3] + s * a2 * k[3] - s * a * k[0])
             + g->g22 * A2 * k[2] * (k[0] - a * s2 * k[3])
           ) / (
             + SQ(k[0]) * g->g33 * (s * k[3] * a)
             + SQ(k[0]) * g->g03 * (s * k[0] * a - s * r2 * k[3] - s * a2 * k[3] - a2 * s * s2 * k[3])
             + SQ(k[1]) * g->g11 * a * s * s2 * (+r2 * k[3] + a2 * k[3] - a * k[0])
             + SQ(k[2]) * g->g22 * (a2 * a * s * s2 * k[3] + r2 * a * s * s2 * k[3] - s * r2 * k[0] - s * a2 * k[0])
             + SQ(k[3]) * g->g33 * s * (k[3] * a * s2 * r2 + k[3] * a2 * a * s2 - k[0] * r2 - k[0] * a2 - a2 * s2 * k[0])
             + SQ(k[3]) * g->g03 * a * s * s2 * (r2 * k[0] + a2 * k[0])
           );
    f[1] = (A1 - a * s * s * k[1] * f[3]) / (k[0] - a * s * s * k[3]);
    f[2] = (A2 + s * k[2] * f[3] * ra2) / (s * k[3] * ra2 - s * a * k[0]);
    orc_vector_norm_to(f, 1.0, g);
}

/* ref: src/sim5polarization.c:249-258 */
orc_cplx orc_polarization_constant_infinity(double a, double alpha, double beta, double incl)
{
    double gamma = -alpha - a * sin(incl);
    double K1 = -gamma;
    double K2 = -beta;
    return K1 + _Complex_I * K2;
}

/* rotation of the polarization angle between the emitter and infinity;
 * ref: src/sim5polarization.c:272-285 */
double orc_polarization_angle_rotation(double a, double inc, double alpha, double beta, orc_cplx kappa)
{
    double k1 = creal(kappa), k2 = cimag(kappa);
    double S = -alpha - a * sin(inc);
    double T = +beta;
    double X = (-S * k2 - T * k1) / (S * S + T * T);
    double Y = (-S * k1 + T * k2) / (S * S + T * T);
    return atan2(Y, X);
}

/* Planck specific intensity with colour hardening and limb darkening; constants are the
 * CGS values of ref: src/sim5const.h:30-41,86-87; ref: src/sim5radiation.c:27-49 */
double orc_blackbody_Iv(double T, double hardf, double cos_mu, double E)
{
    const double h = 6.626069e-27, c = 2.997925e+10, kB = 1.380650e-16;
    const double kev2freq = 2.417990e+17, freq2kev = 4.135667e-18;
    if (T <= 0.0) return 0.0;
    double limbf = (cos_mu >= 0.0) ? 0.5 + 0.75 * cos_mu : 1.0;
    double freq = kev2freq * E;
    return limbf * 2.0 * h * (freq * freq * freq) / SQ(c) / (hardf * hardf * hardf * hardf) /
           expm1((h * freq) / (kB * hardf * T)) * (1. / freq2kev);
}

/* ================================================================================== */
/*  One pixel of the thin-disk image: the body of the caller's loop                    */
/*  ref: examples/04-disk-image-eqplane/disk-image.c:60-104                            */
/* ================================================================================== */
void orc_disk_pixel(const orc_disk_nt *d, double inc, double a, double rms,
                    double alpha, double beta, orc_pixel *px)
{
    orc_geodesic gd;
    int err = 0;
    px->cls = ORC_PX_ERROR; px->gtype = -1; px->err = 0;
    px->r = NAN; px->g = 0.0; px->flux = 0.0; px->image_f = 0.0f; px->image_g = 0.0f;

    orc_geodesic_init_inf(inc, a, alpha, beta, &gd, &err);
    px->err = err;
    if (err) return;
    px->gtype = gd.type;

    for (int order = 0; order < 2; order++) {
        double P = orc_geodesic_find_midplane_crossing(&gd, order);
        if (isnan(P)) { px->cls = order ? ORC_PX_NAN1 : ORC_PX_NAN0; return; }
        double r = orc_geodesic_position_rad(&gd, P);
        if (r >= rms) {
            double g = orc_gfactorK(r, a, gd.l);
            double f = orc_disk_nt_flux(d, r);
            px->cls = order ? ORC_PX_HIT1 : ORC_PX_HIT0;
            px->r = r; px->g = g; px->flux = f;
            px->image_f = f * pow(g, 4.);
            px->image_g = g;
            return;
        }
    }
    px->cls = ORC_PX_MISS;
}
