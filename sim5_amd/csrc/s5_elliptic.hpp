// s5_elliptic.hpp -- Carlson symmetric integrals and the Jacobi functions built on them,
// as gfx950 device code (FP64 VALU; one lane = one ray).
//
// Algorithms restated from the reference (ref: /root/reference/src/sim5elliptic.c):
//   carlson_rf  <- rf :19-52        carlson_rd <- rd :59-98     carlson_rc <- rc :105-137
//   carlson_rj  <- rj :145-206      ell_K <- elliptic_k :218    ell_F_sin <- elliptic_f_sin :274
//   inv_sn/inv_cn/inv_tn <- jacobi_isn/icn/itn :481-528         sncndn <- jacobi_sncndn :536-598
//
// Design notes (CDNA4):
//  * every loop has a trip-count cap so that a wave always drains, whatever the input;
//  * loops leave as soon as no lane of the wave needs another pass (wave vote), lanes that are
//    done keep their value under predication -- rays of one wave are neighbours on the image
//    plane, so their trip counts differ by at most one or two;
//  * the Landen ladder of sncndn is fully unrolled (rung index compile-time) and its rungs are kept by a storage
//    policy: registers for the generic per-ray routines (LadderRegs), LDS for the whole-image and surface kernels
//    (LadderLds / LadderLdsAt); never scratch.  The routine comes in two halves, climb (modulus only) and descent
//    (argument), so a caller with a fixed modulus climbs once (GeodTrack, s5_geod.hpp).
#pragma once
#include "s5_trig.hpp"

namespace S5NS {

#if S5_F_RF7
// ---------------------------------------------------------------------------------------
// R_F(x,y,z), fast variant.  Same duplication theorem; what changes against the reference's
// Numerical-Recipes form (ERRTOL 3e-4, 5th-order series, three divisions per pass):
//  * Carlson's (1995) stopping rule on the initial spread, max|A0 - x_i| 4^-n < tol |A_n|, which needs
//    no division inside the loop; the scaled deviations X, Y, Z are formed once at the end;
//  * the series is carried to 9th order, error ~ 0.01 |X|^10, so tol = 0.03 reaches double precision:
//    ~3 passes where the reference needs ~6.5;
//  * lanes stop individually (their result is a function of their own arguments only, whatever the
//    neighbours in the wave are); the wave leaves the loop when its last lane has converged.
// ---------------------------------------------------------------------------------------
// CHECKED = false: the caller guarantees x, y, z > 0 for every lane whose result it uses (other lanes may carry
// anything: NaN stops a lane's loop at once, its result is discarded by the caller)
// ROOT_X: the caller knows sqrt(x) exactly (x was formed as its square) and z = 1: the first pass then needs ONE square root
// instead of three (sqrt_x is ignored otherwise)
template <bool CHECKED, bool ROOT_X = false>
S5_DEV double carlson_rf_impl(double x, double y, double z, double sqrt_x = 0.0)
{
    S5_FPC_RF
    const double tol = 0.03, third = 1.0 / 3.0;
    bool bad = false;
    if (CHECKED) {
        bad = !(x >= 0.0) || !(y >= 0.0) || !(z >= 0.0);               // sqrt of a negative / NaN -> NaN
        x = fmax(x, 1e-300); y = fmax(y, 1e-300); z = fmax(z, 1e-300);  // at most one argument may be 0
    }
    const double A0 = third * (x + y + z);
    const double dx0 = A0 - x, dy0 = A0 - y;
    const double dev = max3abs(dx0, dy0, A0 - z);
    // The loop carries X_n = 4^n x_n (same for y, z, A): then x_{n+1} = (x_n + lambda_n)/4 becomes
    // X_{n+1} = X_n + Lambda_n with Lambda_n = sqrt(X_n Y_n) + sqrt(X_n Z_n) + sqrt(Y_n Z_n) -- no scaling
    // by 1/4 inside the loop, and since powers of two are exact the values are those of the textbook form.
    double A = A0, pw = 1.0;                        // pw = 2^n
    bool live = dev >= tol * A;
    if (ROOT_X) {
        if (S5_ANY(live)) {
            if (live) {
                const double sy = sqrt_pos(y);
                const double lam = sqrt_x * (sy + 1.0) + sy;
                x += lam;
                y += lam;
                z += lam;
                A += lam;
                pw += pw;
            }
        }
    }
    for (int pass = 0; pass < 32; ++pass) {
        // the wave's vote on a flag compared HERE costs nothing beyond the comparison (s5_math.hpp S5_ANY); a lane that
        // stopped keeps its A, so asking again gives its answer again.  (Without a vote, as a plain divergent loop, the
        // kernels need 8 more vector registers -- the polarized pair kernel then spills -- for the same time.)
        live = dev >= tol * A;
        if (!wave_any(live)) break;
        if (live) {
            const double sx = sqrt_pos(x), sy = sqrt_pos(y), sz = sqrt_pos(z);
            const double lam = sx * (sy + sz) + sy * sz;
            x += lam;
            y += lam;
            z += lam;
            A += lam;
            pw += pw;
        }
    }
    // 1/(4^n A_n) and its square root from one reciprocal square root
    const double rsA = rsqrt_pos(A);
    const double rA = rsA * rsA;
    const double X = dx0 * rA, Y = dy0 * rA, Z = -(X + Y);
    const double E2 = X * Y - Z * Z, E3 = X * Y * Z;
    // coefficient of E2^j E3^k: (-1)^j (1/2)_(j+k) / (j! k! (2(2j+3k)+1))  (DLMF 19.19.7 with E1 = 0), all terms
    // of total degree <= 9 in the deviations
    const double s2 = hfmac(E2, hfmac(E2, hfma(E2, 35.0 / 2176.0, -5.0 / 208.0), 1.0 / 24.0), -0.1);          // pure E2
    const double s3 = hfmac(E3, hfma(E3, 5.0 / 304.0, 3.0 / 104.0), 1.0 / 14.0);                            // pure E3
    const double sx = hfma(E2, hfma(E2, -35.0 / 608.0, 1.0 / 16.0), hfma(E3, -15.0 / 272.0, -3.0 / 44.0)); // x E2 E3
    const double ser = hfma(E2, hfma(E3, sx, s2), hfmac(E3, s3, 1.0));
    const double res = ser * pw * rsA;              // A_n^-1/2 = 2^n (4^n A_n)^-1/2
    return (CHECKED && bad) ? NAN : res;
}
// R_F(sx^2, y, 1) for 0 < sx, y (callers as carlson_rf_positive)
S5_DEV double carlson_rf_root_x(double sx, double x, double y) { return carlson_rf_impl<false, true>(x, y, 1.0, sx); }
S5_DEV double carlson_rf(double x, double y, double z) { return carlson_rf_impl<true>(x, y, z); }
S5_DEV double carlson_rf_positive(double x, double y, double z) { return carlson_rf_impl<false>(x, y, z); }
#else
// ---------------------------------------------------------------------------------------
// R_F(x,y,z) by the duplication theorem, tolerance 3e-4 and 5th-order series as the reference.
// ---------------------------------------------------------------------------------------
S5_DEV double carlson_rf(double x, double y, double z)
{
    const double tol = 0.0003, third = 1.0 / 3.0;
    double dx = 0.0, dy = 0.0, dz = 0.0, mu = 1.0;
    bool live = true;
    for (int pass = 0; pass < 48; ++pass) {
        if (live) {
            double sx = sqrt(x), sy = sqrt(y), sz = sqrt(z);
            double lam = sx * (sy + sz) + sy * sz;
            x = 0.25 * (x + lam);
            y = 0.25 * (y + lam);
            z = 0.25 * (z + lam);
            mu = third * (x + y + z);
            dx = (mu - x) / mu;
            dy = (mu - y) / mu;
            dz = (mu - z) / mu;
            live = max3abs(dx, dy, dz) > tol;      // false for NaN: the lane stops, as in C
        }
        if (!wave_any(live)) break;
    }
    double e2 = dx * dy - dz * dz;
    double e3 = dx * dy * dz;
    return (1.0 + ((1.0 / 24.0) * e2 - 0.1 - (3.0 / 44.0) * e3) * e2 + (1.0 / 14.0) * e3) / sqrt(mu);
}
S5_DEV double carlson_rf_positive(double x, double y, double z) { return carlson_rf(x, y, z); }

#endif

// R_C(x,y), Cauchy principal value for y < 0
S5_DEV double carlson_rc(double x, double y)
{
    const double tol = 0.0003, third = 1.0 / 3.0;
    double xt, yt, w, mu = 1.0, s = 0.0;
    if (y > 0.0) { xt = x; yt = y; w = 1.0; }
    else { xt = x - y; yt = -y; w = sqrt(x) / sqrt(xt); }
    bool live = true;
    for (int pass = 0; pass < 48; ++pass) {
        if (live) {
            double lam = 2.0 * sqrt(xt) * sqrt(yt) + yt;
            xt = 0.25 * (xt + lam);
            yt = 0.25 * (yt + lam);
            mu = third * (xt + yt + yt);
            s = (yt - mu) / mu;
            live = fabs(s) > tol;
        }
        if (!wave_any(live)) break;
    }
    return w * (1.0 + s * s * (0.3 + s * ((1.0 / 7.0) + s * (0.375 + s * (9.0 / 22.0))))) / sqrt(mu);
}

// R_D(x,y,z)
S5_DEV double carlson_rd(double x, double y, double z)
{
    const double tol = 0.0003;
    const double c1 = 3.0 / 14.0, c2 = 1.0 / 6.0, c3 = 9.0 / 22.0, c4 = 3.0 / 26.0;
    const double c5 = 0.25 * c3, c6 = 1.5 * c4;
    double acc = 0.0, w = 1.0, dx = 0.0, dy = 0.0, dz = 0.0, mu = 1.0;
    bool live = true;
    for (int pass = 0; pass < 48; ++pass) {
        if (live) {
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
            live = max3abs(dx, dy, dz) > tol;
        }
        if (!wave_any(live)) break;
    }
    double ea = dx * dy, eb = dz * dz, ec = ea - eb, ed = ea - 6.0 * eb, ee = ed + ec + ec;
    return 3.0 * acc + w * (1.0 + ed * (-c1 + c5 * ed - c6 * dz * ee)
        + dz * (c2 * ee + dz * (-c3 * ec + dz * c4 * ea))) / (mu * sqrt(mu));
}

// R_J(x,y,z,p); returns 0 for arguments outside the validity box, as the reference does
S5_DEV double carlson_rj(double x, double y, double z, double p)
{
    const double tol = 0.0003;
    const double tiny = 2.2325156320215171e-103;   // pow(5*DBL_MIN, 1/3)
    const double big  = 7.8546894664816594e+101;   // 0.3*pow(0.1*DBL_MAX, 1/3)
    const double c1 = 3.0 / 14.0, c2 = 1.0 / 3.0, c3 = 3.0 / 22.0, c4 = 3.0 / 26.0;
    const double c5 = 0.75 * c3, c6 = 1.5 * c4, c7 = 0.5 * c2, c8 = c3 + c3;
    if ((fmin(fmin(x, y), z) < 0.0) || (fmin(fmin(x + y, x + z), fmin(y + z, fabs(p))) < tiny) ||
        (fmax(fmax(x, y), fmax(z, fabs(p))) > big))
        return 0.0;
    double a = 0.0, b = 0.0, rcx = 0.0, acc = 0.0, w = 1.0;
    double xt, yt, zt, pt, dx = 0.0, dy = 0.0, dz = 0.0, dp = 0.0, mu = 1.0;
    if (p > 0.0) { xt = x; yt = y; zt = z; pt = p; }
    else {
        xt = fmin(fmin(x, y), z);
        zt = fmax(fmax(x, y), z);
        yt = x + y + z - xt - zt;
        a = 1.0 / (yt - p);
        b = a * (zt - yt) * (yt - xt);
        pt = yt + b;
        double rho = xt * zt / yt;
        double tau = p * pt / yt;
        rcx = carlson_rc(rho, tau);
    }
    bool live = true;
    for (int pass = 0; pass < 48; ++pass) {
        if (!live) break;                      // lane-level exit: this routine is off the image path
        double sx = sqrt(xt), sy = sqrt(yt), sz = sqrt(zt);
        double lam = sx * (sy + sz) + sy * sz;
        double al = sq(pt * (sx + sy + sz) + sx * sy * sz);
        double be = pt * sq(pt + lam);
        acc += w * carlson_rc(al, be);
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
        live = fmax(fmax(fabs(dx), fabs(dy)), fmax(fabs(dz), fabs(dp))) > tol;
    }
    double ea = dx * (dy + dz) + dy * dz;
    double eb = dx * dy * dz;
    double ec = dp * dp;
    double ed = ea - 3.0 * ec;
    double ee = eb + 2.0 * dp * (ea - ec);
    double ans = 3.0 * acc + w * (1.0 + ed * (-c1 + c5 * ed - c6 * ee) + eb * (c7 + dp * (-c8 + dp * c4))
        + dp * ea * (c2 - dp * c3) - c2 * dp * ec) / (mu * sqrt(mu));
    if (p <= 0.0) ans = a * (b * ans + 3.0 * (rcx - carlson_rf(xt, yt, zt)));
    return ans;
}

// ---------------------------------------------------------------------------------------
// Legendre / Jacobi
// ---------------------------------------------------------------------------------------
#if S5_F_AGMK
// K(m) by the arithmetic-geometric mean, K = pi / (2 AGM(1, sqrt(1-m))): ~5 square roots where
// R_F(0, 1-m, 1) takes ~20.  Quadratic convergence: once |a-b| <= 2e-8 a the next mean is exact to
// double precision, so the loop stops there and uses (a+b)/2.
S5_DEV double ell_K(double m)
{
    if (m == 1.0) m = 1.0 - 1e-8;
    double a = 1.0, b = msqrt(1.0 - m);
    bool live = fabs(a - b) > 2e-8 * a;
    for (int pass = 0; pass < 16; ++pass) {
        if (!wave_any(live)) break;
        if (live) {
            const double an = 0.5 * (a + b);
            b = sqrt_pos(a * b);
            a = an;
            live = fabs(a - b) > 2e-8 * a;
        }
    }
    return mdiv(3.14159265358979323846, a + b);
}
#else
S5_DEV double ell_K(double m)
{
    if (m == 1.0) m = 1.0 - 1e-8;
    return carlson_rf(0.0, 1.0 - m, 1.0);
}
#endif

S5_DEV double ell_F_sin(double s, double m)
{
    if (m == 1.0) m = 0.99999999;
    if (s == 0.0) return 0.0;
    double s2 = s * s;
    return s * carlson_rf(1. - s2, 1.0 - s2 * m, 1.0);
}

S5_DEV double inv_sn(double z, double m)
{
    if (fabs(m - 0.0) < 1e-8) return asin(z);
    if (fabs(m - 1.0) < 1e-8) return log(msqrt(mdiv(1. + z, 1. - z)));
    return z * carlson_rf(1.0 - z * z, 1.0 - m * z * z, 1.0);
}

S5_DEV double inv_cn(double z, double m)
{
    if ((z > +1.0) && (z < +1.0 + 1e-8)) z = +1.0;
    if ((z < -1.0) && (z > -1.0 - 1e-8)) z = -1.0;
    if ((m > +1.0) && (m < +1.0 + 1e-8)) m = 1.0;
    if ((m < 0.0) && (m > 0.0 - 1e-8)) m = 0.0;

    if (z == 0.0) return ell_K(m);
    if (z == 1.0) return 0.0;
    if (m == 0.0) return acos(z);
    if (m == 1.0) return log(mdiv(1. + msqrt(1. - z), z));

    double base = msqrt(1. - z * z) * carlson_rf(z * z, 1.0 - m * (1. - z * z), 1.0);
    if (z > 0.0) return base;
    return mdiv(2., msqrt(1. - m)) * ell_F_sin(-z, mdiv(m, m - 1.)) + base;
}

S5_DEV double inv_tn(double z, double m)
{
    if (m == 0.0) return atan(z);
    if (m == 1.0) return log(z + msqrt(1. + z * z));
    return inv_sn(msqrt(mdiv(z * z, 1. + z * z)), m);
}

// sn, cn, dn by the descending Landen ladder (ref: src/sim5elliptic.c:580-642).  The rungs (a_i, g_i) that
// the climb produces and the descent consumes are kept by a storage policy:
//   LadderRegs  in VGPRs (generic per-ray routines; any launch shape)
//   LadderLds   in LDS, element k of a lane at base[k * 256] (256-thread workgroups only).  The whole-image
//               kernels use this one: the register arrays cost 40 VGPRs and, worse, whole-array v_mov chains
//               at every control-flow join around the inlined call; in LDS a rung is two ds_write_b64 and two
//               ds_read_b64 with immediate offsets, lanes 8 B apart (conflict-free), 2 x LADDER_RUNGS x 2 KB per workgroup
//               (32 KB fast -- 34 KB in the image kernels, which keep one more row: ladder_descend_squares -- 52 KB strict).
// 13 rungs as in the reference; the AGM of a double-precision modulus (1 - m >= 1.1e-16, the m == 1 clamp included)
// reaches 1e-8 at rung index 7 at the latest (scanned over 1e5 moduli up to 1 - 2^-53), so the fast variant keeps 8.
constexpr int LADDER_RUNGS = S5_FAST ? 8 : 13;

struct LadderRegs {
    double a[LADDER_RUNGS], g[LADDER_RUNGS];
    S5_DEV void put(int i, double av, double gv) { a[i] = av; g[i] = gv; }
    S5_DEV double get_a(int i) const { return a[i]; }
    S5_DEV double get_g(int i) const { return g[i]; }
};

struct LadderLds {
    double* base;                               // this lane's column of the workgroup's ladder block
    S5_DEV void put(int i, double av, double gv) { base[(2 * i) * 256] = av; base[(2 * i + 1) * 256] = gv; }
    S5_DEV void put_a(int i, double av) { base[(2 * i) * 256] = av; }         // (i = LADDER_RUNGS: the block's extra row)
    S5_DEV double get_a(int i) const { return base[(2 * i) * 256]; }
    S5_DEV double get_g(int i) const { return base[(2 * i + 1) * 256]; }
};

// The routine in two halves.  The climb (the AGM rungs) depends on the modulus only; the descent takes the
// argument u.  A caller that evaluates sn/cn/dn many times for ONE modulus (a geodesic's r(P) and mu(P): the
// moduli are constants of the ray) climbs once and keeps the rungs; sncndn_with below is climb + descent, so
// both uses run the same operations in the same order.
struct LadderState {
    double c;        // the last arithmetic mean, multiplies the argument
    double d;        // scale of the flipped (m > 1) branch
    int top;         // index of the last rung
    bool flipped;
    bool degenerate; // 1 - m == 0 after the clamp: unreachable, kept for parity
    bool incomplete; // the AGM had not converged when the NR rungs were used up (a caller with a short ladder checks)
};

// STORE_NEXT: the arithmetic mean ABOVE the last rung, a_{top+1} = (a_top + g_top) / 2 = st.c, is stored as well (slot
// top + 1 of the a column; the storage must have NR + 1 of them): ladder_descend_squares reads a_{i+1} at every rung
template <class Ladder, int NR = LADDER_RUNGS, bool STORE_NEXT = false>
S5_DEV void ladder_climb(Ladder& lad, double m, LadderState& st)
{
    S5_FPC_LADDER
    if (m == 1.0) m = 0.999999999;
    const double conv = 1.0e-8;
    double emc = 1.0 - m;
    st.d = 1.0; st.c = 0.0; st.top = NR - 1; st.flipped = false; st.incomplete = false;
    st.degenerate = (emc == 0.0);
    if (st.degenerate) return;
    st.flipped = emc < 0.0;
    if (st.flipped) {
        double d = 1.0 - emc;
        emc /= -1.0 / d;
        st.d = msqrt(d);
    }
    double a = 1.0, c = 0.0;
    int top = NR - 1;
    bool climbing = true;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        if (climbing) {
            emc = sqrt_pos(emc);                // 0 < 1 - m <= 1, and products of positive means after that
            lad.put(i, a, emc);
            c = 0.5 * (a + emc);
            if (fabs(a - emc) <= conv * a) { climbing = false; top = i; if constexpr (STORE_NEXT) lad.put_a(i + 1, c); }
            else { emc *= a; a = c; }
        }
        if (!S5_ANY(climbing)) break;          // rungs above are never read (i <= top below)
    }
    st.c = c; st.top = top; st.incomplete = climbing;
}

#if S5_FAST
// The descent of the regular case (modulus in (0, 1), rungs complete) carried as FRACTIONS, without the final square root
// and division: with s0 = sin(u c) != 0 and rho^2 = C^2 + ga^2,
//   sn = sign(s0) |ga| / rho,   cn = sign(s0) sign(ga) C / rho,   dn = N / D   (N and D have the same sign).
// The recurrence  a <- c a;  c <- dn c;  dn <- (g_i + a)/(b_i + a);  a <- c/b_i  of the reference (ref: src/sim5elliptic.c:
// 598-606) with a = A/al, c = C/ga, dn = N/D:  dn' = (g_i ga al + C A)/(b_i ga al + C A),  c' = N C/(D ga),  a'' = c'/b_i
// -- seven multiplications per rung and no division; cot(u c) enters as cos/sin without being divided.  Only the ratios
// matter and C^2 + ga^2 is roughly cubed by a rung (it starts in [1/4, 1]), so every second rung the pair (C, ga) is
// rescaled by the exact power of two that brings ga's exponent back to zero.
template <class Ladder, int NR = LADDER_RUNGS>
S5_DEV void ladder_descend_fractions(const Ladder& lad, const LadderState& st, const double s0, const double c0,
                                     double& C, double& ga, double& N, double& D)      // (s0, c0) = sincos(u st.c)
{
    S5_FPC_LADDER
    const int top = st.top;
    const double c = st.c;
    double A = c0, al = s0;
    C = c * c0; ga = s0; N = 1.0; D = 1.0;
#pragma unroll
    for (int i = NR - 1; i >= 0; --i) {
        if (S5_ANY(i <= top)) {             // rungs no lane of the wave reached are skipped (s5_math.hpp S5_ANY)
            if (i <= top) {
                const double b = lad.get_a(i), g = lad.get_g(i);
                const double t1 = C * A, t2 = ga * al;
                const double Nn = g * t2 + t1, Dn = b * t2 + t1;
                C = N * C;
                ga = D * ga;
                if ((i & 1) == 0) {
                    const int e = -__builtin_amdgcn_frexp_exp(fabs(C) + fabs(ga));
                    C = __builtin_amdgcn_ldexp(C, e);
                    ga = __builtin_amdgcn_ldexp(ga, e);
                }
                A = C;
                al = ga * b;
                N = Nn; D = Dn;
            }
        }
    }
}


// The same descent with the carried pair (A, al) eliminated.  After a rung A = C and al = ga b_i, so the next rung's products
// are t1 = C^2 and t2 = ga^2 a_i -- and the FIRST rung a lane takes fits the same form with a_{top+1} = c, the mean the climb
// ended on: t1 / t2 = (c c0)^2 / (s0^2 c) = c cot^2, the reference's a <- c a.  Every rung then reads a_{i+1} beside its own
// (a_i, g_i) (ladder_climb<..., STORE_NEXT = true> stored the extra one) and updates C, ga, N, D in place: seven
// multiplications, no register copies at the join of the predicated block (the form above: seven and three copies of a
// double per rung).  The pair (C, ga) is rescaled by a power of two after rung 5 (lanes that entered at rungs 5-7: moduli
// within ~1e-3 of 1) and after the LAST rung: with (e_C, e_N) the binary exponents of the pairs (C, ga) and (N, D), a rung
// maps (e_C, e_N) to (e_C + e_N, 2 e_C) -- they roughly double -- and the start is (<= 4, 0) (C^2 + ga^2 lies between c^2
// and 1, and the last mean c = pi / 2K(m) >= 0.08 for a double-precision modulus).  Five rungs without rescaling (top = 4,
// the common case) end at (84, 88); three rungs, a rescaling and five more (top = 7) at (264, 240) with squares of 2^-240
// inside the last rung: the products the addition theorem forms from the rescaled pair and (N, D) stay above 2^-600.
template <class Ladder, int NR = LADDER_RUNGS>
S5_DEV void ladder_descend_squares(const Ladder& lad, const LadderState& st, const double s0, const double c0,
                                   double& C, double& ga, double& N, double& D)      // (s0, c0) = sincos(u st.c)
{
    S5_FPC_LADDER
    static_assert(NR <= 8, "rescaling points are laid out for at most 8 rungs");
    const int top = st.top;
    C = st.c * c0; ga = s0; N = 1.0; D = 1.0;
#pragma unroll
    for (int i = NR - 1; i >= 0; --i) {
        if (S5_ANY(i <= top)) {             // rungs no lane of the wave reached are skipped (s5_math.hpp S5_ANY)
            if (i <= top) {
                const double an = lad.get_a(i + 1), b = lad.get_a(i), g = lad.get_g(i);
                const double t1 = C * C, t2 = (ga * ga) * an;
                C = N * C;
                ga = D * ga;
                N = g * t2 + t1;
                D = b * t2 + t1;
                if (i == 5 || i == 0) {
                    const int e = -__builtin_amdgcn_frexp_exp(fabs(C) + fabs(ga));
                    C = __builtin_amdgcn_ldexp(C, e);
                    ga = __builtin_amdgcn_ldexp(ga, e);
                }
            }
        }
    }
}
#endif

template <class Ladder, int NR = LADDER_RUNGS>
S5_DEV void ladder_descend(const Ladder& lad, const LadderState& st, double u, double& sn, double& cn, double& dn)
{
    S5_FPC_LADDER
    if (st.degenerate) {
        cn = 1.0 / cosh(u);
        dn = cn;
        sn = tanh(u);
        return;
    }
    if (st.flipped) u *= st.d;
    const int top = st.top;
    double a, c = st.c;
    u *= c;
    double s0, c0;
    msincos(u, s0, c0);
    sn = s0; cn = c0; dn = 1.0;
    if (s0 != 0.0) {
#if S5_FAST
        // the descent carried as fractions (ladder_descend_fractions above), leaving through ONE reciprocal square root
        double C, ga, N, D;
        ladder_descend_fractions<Ladder, NR>(lad, st, s0, c0, C, ga, N, D);
        const double rs = rsqrt_pos(C * C + ga * ga);
        a = fabs(ga) * rs;
        sn = (s0 >= 0.0 ? a : -a);
        cn = (ga >= 0.0 ? C : -C) * rs;
        if (!(s0 >= 0.0)) cn = -cn;
        dn = mdiv(N, D);
        c = 0.0;
#else
        a = mdiv(c0, s0);
        c *= a;
#pragma unroll
        for (int i = NR - 1; i >= 0; --i) {
            if (wave_any(i <= top)) {          // rungs no lane of the wave reached are skipped
                if (i <= top) {
                    const double b = lad.get_a(i);
                    a *= c;
                    c *= dn;
                    dn = mdiv(lad.get_g(i) + a, b + a);
                    a = mdiv(c, b);
                }
            }
        }
        a = mdiv(1.0, msqrt(c * c + 1.0));
        sn = (s0 >= 0.0 ? a : -a);
        cn = c * sn;
#endif
    }
    if (st.flipped) {
        a = dn;
        dn = cn;
        cn = a;
        sn = mdiv(sn, st.d);
    }
}

template <class Ladder>
S5_DEV void sncndn_with(Ladder& lad, double u, double m, double& sn, double& cn, double& dn)
{
    LadderState st;
    ladder_climb(lad, m, st);
    ladder_descend(lad, st, u, sn, cn, dn);
}

// Rungs a double-precision modulus can need: with 0 <= m < 1 the AGM reaches |a - g| <= 1e-8 a at rung index 7 at
// the latest (m = 1 - 2^-53 and the m == 1 clamp included; scanned in tests/tools, same count for both variants), so a
// ladder kept for a known-valid modulus stores 8 rungs instead of the reference's array of 13.
constexpr int LADDER_RUNGS_VALID = 8;

// rungs of one lane in LDS with a caller-chosen lane stride (workgroup size): element k at base[k * STRIDE]
template <int STRIDE>
struct LadderLdsAt {
    double* base;
    S5_DEV void put(int i, double av, double gv) { base[(2 * i) * STRIDE] = av; base[(2 * i + 1) * STRIDE] = gv; }
    S5_DEV double get_a(int i) const { return base[(2 * i) * STRIDE]; }
    S5_DEV double get_g(int i) const { return base[(2 * i + 1) * STRIDE]; }
};

S5_DEV void sncndn(double u, double m, double& sn, double& cn, double& dn)
{
    LadderRegs lad;
    sncndn_with(lad, u, m, sn, cn, dn);
}

// LDS-backed form for kernels launched with 256-thread one-dimensional workgroups
S5_DEV void sncndn_lds(double u, double m, double& sn, double& cn, double& dn)
{
    __shared__ double s_ladder[2 * LADDER_RUNGS * 256];
    LadderLds lad{&s_ladder[threadIdx.x]};
    sncndn_with(lad, u, m, sn, cn, dn);
}

// A kernel launched with 256-thread one-dimensional workgroups may define S5_LADDER_IN_LDS before including the
// headers: the generic routines (position_rad, position_pol, ...) then keep the ladder rungs in LDS as well.
#define S5_SNCNDN sncndn
S5_DEV double jac_sn(double u, double m) { double s, c, d; S5_SNCNDN(u, m, s, c, d); return s; }
S5_DEV double jac_cn(double u, double m) { double s, c, d; S5_SNCNDN(u, m, s, c, d); return c; }
S5_DEV double jac_dn(double u, double m) { double s, c, d; S5_SNCNDN(u, m, s, c, d); return d; }

} // namespace S5NS
