// s5_trig.hpp -- sin, cos, acos for the bounded arguments of this path, fast variant.
//
// The device libm's f64 sin/cos carry a Payne-Hanek reduction for huge arguments: never taken here, but
// it costs registers (the RK4 fallback of the step-wise integrator needed 210 VGPRs mostly because of
// four inlined cos) and instructions.  Every angle on this path is a polar angle, an AGM amplitude or a
// third of an atan2: |x| < ~1e3.  For those a two-constant Cody-Waite reduction by pi/2 with FMAs is exact
// to < 1 ulp of the reduced argument, followed by the classical minimax kernels on [-pi/4, pi/4]
// (coefficients: fdlibm k_sin.c / k_cos.c / e_acos.c, Sun Microsystems, freely distributable).
// ~35 instructions for sin and cos together, ~45 for acos; errors < 1 ulp on the stated ranges.
// The strict variant keeps the device libm.
#pragma once
#include "s5_math.hpp"

namespace S5NS {

#if S5_F_LIBM

// sin(y), cos(y) for |y| <= pi/4 (+ a little), y = yh + yl
S5_DEV double kernel_sin(double y, double yl)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = y * y;
    const double v = z * y;
    const double r = hfmac(z, hfmac(z, hfmac(z, hfma(z, S6, S5), S4), S3), S2);
    return y - ((z * (0.5 * yl - v * r) - yl) - v * S1);
}

S5_DEV double kernel_cos(double y, double yl)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = y * y;
    const double r = z * hfmac(z, hfmac(z, hfmac(z, hfmac(z, hfma(z, C6, C5), C4), C3), C2), C1);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - y * yl));
}

// |x| up to ~1e5: n = nearest integer to x / (pi/2), reduced argument as a head/tail pair
S5_DEV int reduce_pio2(double x, double& yh, double& yl)
{
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632679489655800e+00;     // 0x1.921fb54442d18p+0
    const double pio2_lo = 6.12323399573676603587e-17;     // 0x1.1a62633145c07p-54
    const double fn = __builtin_rint(x * two_over_pi);
    const double r = __builtin_fma(-fn, pio2_hi, x);       // exact when it cancels
    const double w = fn * pio2_lo;
    yh = r - w;
    yl = (r - yh) - w;
    return (int)fn;
}

S5_DEV void msincos(double x, double& s, double& c)
{
    double yh, yl;
    const int n = reduce_pio2(x, yh, yl);
    const double ks = kernel_sin(yh, yl), kc = kernel_cos(yh, yl);
    const bool swap = (n & 1) != 0;
    const double ss = swap ? kc : ks;
    const double cc = swap ? ks : kc;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cc : cc;
}

S5_DEV double mcos(double x) { double s, c; msincos(x, s, c); return c; }
S5_DEV double msin(double x) { double s, c; msincos(x, s, c); return s; }

// sin and cos from a table of SC_N nodes per turn (host-made in long double, capi_core.hip: tab[2 i] = sin(2 pi i / SC_N),
// tab[2 i + 1] = cos): the angle is split into the nearest node and a remainder |d| <= pi / SC_N = 0.0123, for which three
// series terms each are exact to rounding (next terms: d^8 / 9! = 1e-21, d^8 / 8! = 1e-20 relative), and the two are joined
// by the angle-addition formulas -- ~21 instructions and one 16-byte load where the minimax kernels with their Cody-Waite
// reduction and quadrant selects take ~40.  Any finite |x| < 2^20 (the node index wraps; the remainder is formed with the
// SIGNED node number against a two-part pi / 128 whose head has 13 trailing zero bits, so n * head is exact).  The image
// kernels' fast variant uses it for the one sincos a crossing needs (s5_thindisk.hpp); accuracy 1-2 ulp.
constexpr int SC_N = 256;
S5_DEV void msincos_tab(const double* __restrict__ tab, double x, double& s, double& c)
{
    const double inv_step = 40.74366543152521;               // SC_N / (2 pi)
    const double step_hi = 0.02454369260615863, step_lo = 1.1630542938260349e-14;
    const double fn = __builtin_rint(x * inv_step);
    const int i = ((int)fn) & (SC_N - 1);
    double d = __builtin_fma(-fn, step_hi, x);
    d = __builtin_fma(-fn, step_lo, d);
    const double z = d * d;
    const double ps = __builtin_fma(z, __builtin_fma(z, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0);      // (sin d / d - 1) / z
    const double pc = __builtin_fma(z, __builtin_fma(z, -1.0 / 720.0, 1.0 / 24.0), -0.5);             // (cos d - 1) / z
    const double sd = __builtin_fma(d * z, ps, d);
    const double cm = z * pc;                                                                            // cos d - 1
    const double S = tab[2 * i], C = tab[2 * i + 1];          // (one 16-byte load)
    // sin(a + d) = S + (S (cos d - 1) + C sin d),  cos(a + d) = C + (C (cos d - 1) - S sin d)
    s = S + __builtin_fma(S, cm, C * sd);
    c = C + __builtin_fma(C, cm, -(S * sd));
}

// The same for an angle that is almost always small (the change of the polar angle over one integrator step): for
// |x| <= pi/4 the reduction above gives n = 0 and a zero tail, so the two kernels are evaluated on x directly -- the values
// msincos returns, without the reduction, the tail terms and the quadrant selects (18 instructions instead of ~40); the
// lanes beyond take msincos, their wave with them.
S5_DEV void msincos_small(double x, double& s, double& c)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double v = z * x;
    const double rs = hfmac(z, hfmac(z, hfmac(z, hfma(z, S6, S5), S4), S3), S2);
    s = x + (z * (v * rs) + v * S1);
    const double rc = z * hfmac(z, hfmac(z, hfmac(z, hfmac(z, hfma(z, C6, C5), C4), C3), C2), C1);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    c = w + (((1.0 - w) - hz) + z * rc);
    const bool big = !(fabs(x) <= 0.785);
    if (wave_any(big)) {
        if (big) msincos(x, s, c);
    }
}

// acos(x), |x| <= 1 (NaN outside), after fdlibm e_acos.c
S5_DEV double macos(double x)
{
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    const double pi = 3.14159265358979311600e+00;
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05;
    const double qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    const double ax = fabs(x);
    if (!(ax <= 1.0)) return NAN;
    if (ax < 0.5) {
        const double z = x * x;
        const double p = z * hfmac(z, hfmac(z, hfmac(z, hfmac(z, hfma(z, pS5, pS4), pS3), pS2), pS1), pS0);
        const double q = hfmac(z, hfmac(z, hfmac(z, hfma(z, qS4, qS3), qS2), qS1), 1.0);
        const double r = mdiv(p, q);
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    const double z = (1.0 - ax) * 0.5;
    const double p = z * hfmac(z, hfmac(z, hfmac(z, hfmac(z, hfma(z, pS5, pS4), pS3), pS2), pS1), pS0);
    const double q = hfmac(z, hfmac(z, hfmac(z, hfma(z, qS4, qS3), qS2), qS1), 1.0);
    const double sq_ = msqrt(z);
    const double r = mdiv(p, q);
    if (x < 0.0) {
        const double w = r * sq_ - pio2_lo;
        return pi - 2.0 * (sq_ + w);
    }
    // head of sqrt(z) with the low 32 bits cleared, for an exact correction term
    const double df = __longlong_as_double(__double_as_longlong(sq_) & 0xffffffff00000000ll);
    const double cc = mdiv(z - df * df, sq_ + df);
    const double w = r * sq_ + cc;
    return 2.0 * (df + w);
}

// atan2(y, x) for finite arguments, not both zero (no NaN / infinity / signed-zero bookkeeping): the ratio
// of the smaller to the larger magnitude is folded below tan(pi/8) with the same single division
// (atan r = pi/4 + atan((r-1)/(r+1))), then r + r s P(s), s = r^2, P of degree 10 (own Chebyshev fit of
// (atan(sqrt s)/sqrt s - 1)/s on [0, tan^2(pi/8)], max relative error 1.2e-16), then the octant is undone.
// ~45 instructions where the device libm's general atan2 takes ~120.
S5_DEV double matan2(double y, double x)
{
    const double A0 = -0.3333333333333333, A1 = 0.19999999999995563, A2 = -0.14285714284681905,
                 A3 = 0.11111111016277783, A4 = -0.09090904608874864, A5 = 0.07692183692273913,
                 A6 = -0.06664516112329982, A7 = 0.0585817293772659, A8 = -0.050855066749560494,
                 A9 = 0.03923170243065068, A10 = -0.019175342136636544;
    const double pio4_hi = 7.85398163397448278999e-01, pio4_lo = 3.06161699786838301793e-17;
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    const double pi_hi = 3.14159265358979311600e+00, pi_lo = 1.22464679914735320717e-16;
    const double ax = fabs(x), ay = fabs(y);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    const bool fold = mn > 0.41421356237309503 * mx;
    const double num = fold ? mn - mx : mn;
    const double den = fold ? mn + mx : mx;
    const double r = mdiv(num, den);
    const double s2 = r * r;
    const double P = hfmac(s2, hfmac(s2, hfmac(s2, hfmac(s2, hfmac(s2, hfmac(s2, hfmac(s2, hfmac(s2, hfmac(s2,
                     hfma(s2, A10, A9), A8), A7), A6), A5), A4), A3), A2), A1), A0);
    double t = hfma(r * s2, P, fold ? pio4_lo : 0.0) + r;       // atan(mn/mx) - (fold ? pi/4 head : 0)
    t = fold ? pio4_hi + t : t;
    if (ay > ax) t = pio2_hi - (t - pio2_lo);
    if (x < 0.0) t = pi_hi - (t - pi_lo);
    return (y < 0.0) ? -t : t;
}

// cos(z/3) for z in [0, pi] (the angle of the trigonometric cubic solution): the argument never leaves
// [0, pi/3], so one polynomial in z^2 replaces reduction + both kernels (own Chebyshev fit, max relative error
// 2.7e-16 = 1.2 ulp including evaluation rounding)
S5_DEV double mcos_third(double z)
{
    const double c1 = -0.05555555555555555, c2 = 0.0005144032921810684, c3 = -1.9051973784476465e-06,
                 c4 = 3.7801535284439776e-09, c5 = -4.6668561610329095e-12, c6 = 3.9283232935466075e-15,
                 c7 = -2.3976517600295036e-18, c8 = 1.084136668630678e-21;
    const double u = z * z;
    return hfmac(u, hfmac(u, hfmac(u, hfmac(u, hfmac(u, hfmac(u, hfmac(u, hfma(u, c8, c7), c6), c5), c4), c3), c2), c1), 1.0);
}

// exp(x) for |x| <= 700 (no overflow / denormal / NaN handling): x = k ln2 + r, |r| <= ln2/2, Taylor series of
// degree 12 in r (remainder 0.347^13/13! = 1.7e-16 relative), scaled by 2^k with v_ldexp_f64.  ~22 instructions
// where the device libm's exp takes ~45.
S5_DEV double mexp(double x)
{
    const double log2e = 1.44269504088896338700e+00;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double fk = __builtin_rint(x * log2e);
    const double r = __builtin_fma(-fk, ln2_lo, __builtin_fma(-fk, ln2_hi, x));
    double p = hfma(r, 1.0 / 479001600.0, 1.0 / 39916800.0);
    p = hfmac(r, p, 1.0 / 3628800.0); p = hfmac(r, p, 1.0 / 362880.0); p = hfmac(r, p, 1.0 / 40320.0);
    p = hfmac(r, p, 1.0 / 5040.0); p = hfmac(r, p, 1.0 / 720.0); p = hfmac(r, p, 1.0 / 120.0);
    p = hfmac(r, p, 1.0 / 24.0); p = hfmac(r, p, 1.0 / 6.0); p = hfmac(r, p, 0.5);
    p = hfmac(r, p, 1.0); p = hfmac(r, p, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)fk);
}

// log(x) for positive, finite, normal x (after fdlibm e_log.c): x = 2^k (1+f), sqrt(2)/2 <= 1+f < sqrt(2),
// s = f/(2+f), log(1+f) = f - f^2/2 + s (f^2/2 + R(s^2)); < 1 ulp.  The exponent/mantissa split uses
// the v_frexp instructions instead of integer surgery.
S5_DEV double mlog(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    double m = __builtin_amdgcn_frexp_mant(x);                 // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    k = low ? k - 1 : k;
    const double f = m - 1.0;
    const double s = mdiv(f, 2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * hfmac(w, hfma(w, Lg6, Lg4), Lg2);
    const double t2 = z * hfmac(w, hfmac(w, hfma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double res = dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    return (x > 0.0) ? res : ((x == 0.0) ? -INFINITY : NAN);
}

#else

S5_DEV double mlog(double x) { return log(x); }
S5_DEV double mexp(double x) { return exp(x); }
S5_DEV void msincos(double x, double& s, double& c) { s = sin(x); c = cos(x); }
S5_DEV void msincos_small(double x, double& s, double& c) { s = sin(x); c = cos(x); }
S5_DEV double mcos(double x) { return cos(x); }
S5_DEV double msin(double x) { return sin(x); }
S5_DEV double macos(double x) { return acos(x); }
S5_DEV double matan2(double y, double x) { return atan2(y, x); }
S5_DEV double mcos_third(double z) { return cos(z / 3.); }

#endif

} // namespace S5NS
