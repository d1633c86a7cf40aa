// s5_kerr.hpp -- Kerr / Minkowski metric, connection, tetrads and photon kinematics as gfx950
// device code.  Restated from the reference (ref: /root/reference/src/sim5kerr.c); the line
// ranges are given at each routine.
//
// Data layout: the connection is kept as its 20 structurally non-zero entries in registers
// (the reference fills a dense 4x4x4 array, 512 B, of which 44 entries are always zero).  Sums
// over the connection visit the non-zero entries in the reference's (j<=k) order, so skipping the
// zeros does not change any rounding.
#pragma once
#include "s5_elliptic.hpp"

namespace S5NS {

struct Metric { double a, r, m, g00, g11, g22, g33, g03; };   // = sim5gpu_metric (64 B)
struct Tetrad { double e[4][4]; Metric metric; };               // = sim5gpu_tetrad (192 B)

// G^t_{tr} G^t_{t th} G^t_{r ph} G^t_{th ph} | G^r_{..} x6 | G^th_{..} x6 | G^ph_{..} x4
// (off-diagonal entries carry the factor 2 of the symmetric pair, as in the reference)
struct Conn {
    double t01, t02, t13, t23;
    double r00, r03, r11, r12, r22, r33;
    double h00, h03, h11, h12, h22, h33;
    double p01, p02, p13, p23;
};

S5_DEV double r_horizon(double a) { return 1. + sqrt(1. - a * a); }           // ref :981

S5_DEV double r_isco(double a)                                                 // ref :994-1004
{
    double z1 = 1. + cbrt(1. - a * a) * (cbrt(1. + a) + cbrt(1. - a));
    double z2 = sqrt(3. * (a * a) + z1 * z1);          // (3.*sqr(a) of the reference: 3 (a a))
    return 3. + z2 - sqrt((3. - z1) * (3. + z1 + 2. * z2));
}

S5_DEV void flat_metric(double r, double m, Metric& g)                         // ref :31-50
{
    g.a = 0.0; g.r = r; g.m = m;
    g.g00 = -1.0; g.g11 = +1.0; g.g22 = +r * r; g.g33 = +r * r * (1. - m * m); g.g03 = 0.0;
}

S5_DEV void kerr_metric(double a, double r, double m, Metric& g)               // ref :75-101
{
    double r2 = r * r, a2 = a * a, m2 = m * m;
    double S = r2 + a2 * m2;
    double s2_S = mdiv(1.0 - m2, S);
    g.a = a; g.r = r; g.m = m;
    g.g00 = -1. + mdiv(2.0 * r, S);
    g.g11 = mdiv(S, r2 - 2. * r + a2);
    g.g22 = S;
    g.g33 = ((a2 + r2) * S + 2. * r * a2 * s2_S * S) * s2_S;
    g.g03 = -2. * a * r * s2_S;
}

S5_DEV void kerr_metric_contravariant(double a, double r, double m, Metric& g) // ref :105-132
{
    double r2 = r * r, a2 = a * a, m2 = m * m;
    double S = r2 + a2 * m2;
    double SD = S * (r2 - 2. * r + a2);
    g.a = a; g.r = r; g.m = m;
    g.g00 = -sq(r2 + a2) / SD + a2 * (1. - m2) / S;
    g.g11 = (r2 - 2. * r + a2) / S;
    g.g22 = 1. / S;
    g.g33 = 1. / S / (1. - m2) - a2 / SD;
    g.g03 = -2. * a * r / SD;
}

S5_DEV void flat_connection(double r, double m, Conn& G)                       // ref :199-229
{
    double s = sqrt(1. - m * m);
    G.t01 = G.t02 = G.t13 = G.t23 = 0.0;
    G.r00 = G.r03 = G.r11 = G.r12 = 0.0;
    G.r22 = -r;
    G.r33 = -r * s * s;
    G.h00 = G.h03 = G.h11 = G.h22 = 0.0;
    G.h12 = 2.0 * 1. / r;
    G.h33 = -m * s;
    G.p01 = G.p02 = 0.0;
    G.p13 = 2.0 * 1. / r;
    G.p23 = 2.0 * m / s;
}

S5_DEV void kerr_connection(double a, double r, double m, Conn& G)             // ref :233-316
{
    double rS = 2.0 * r;
#if S5_FAST
    // sin(theta) and 1/sin(theta) from one rsq seed (on the axis: 0 and inf, as sqrt and the division give)
    const double om = 1. - m * m;
    double s, inv_s;
    sqrt_rsqrt_pos(om, s, inv_s);
    if (om == 0.0) { s = 0.0; inv_s = INFINITY; }
#else
    double s = msqrt(1. - m * m);
#endif
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
    double Rq = a2 + 2. * r2 + a2cc;
    double R = Rq * Rq;
    double D = r2 - 2. * r + a2;
    double S = r2 + a2c2;
#if S5_FAST
    // the three reciprocals from one: 1/(S D R), then products (S, R > 0; D > 0 outside the horizon)
    const double SD = S * D;
    const double inv_all = mrcp(SD * R);
    const double R_1 = SD * inv_all;
    const double D_1 = (S * R) * inv_all;
    const double S_1 = (D * R) * inv_all;
    const double S_3 = S_1 * S_1 * S_1;
    const double m_s = m * inv_s;
#else
    double S_1 = mdiv(1., S);
    double S_3 = mdiv(1., S * S * S);
    double D_1 = mdiv(1., D);
    double R_1 = mdiv(1., R);
    double m_s = mdiv(m, s);
#endif
    double DR_1 = D_1 * R_1;
    double DS_1 = D_1 * S_1;
    double dbl_r2 = 2. * r2;

    G.t01 = 2.0 * 4.0 * (a2_r2) * (r2 - a2c2) * DR_1;
    G.t02 = 2.0 * -4.0 * a2cs * rS * R_1;
    G.t13 = 2.0 * 2.0 * a * s2 * (a4 - 3. * a2r2 - 6. * r4 + a2cc * (a2 - r2)) * DR_1;
    G.t23 = -G.t02 * s2 * a;

    G.r00 = D * (r2 - a2c2) * S_3;
    G.r03 = -2.0 * G.r00 * a * s2;
    G.r11 = (r * (a2 - r) + a2 * (1. - r) * c2) * DS_1;
    G.r12 = -2.0 * a2cs * S_1;
    G.r22 = -r * D * S_1;
    G.r33 = -D * s2 * (2. * a2c2 * r3 + r2 * r3 + a2 * a2c2 * s2 + a2c2 * a2c2 * r - a2r2 * s2) * S_3;

    G.h00 = -2.0 * r * a2cs * S_3;
    G.h03 = mdiv(2.0 * -G.h00 * a2_r2, a);
    G.h11 = +a2cs * DS_1;
    G.h12 = 2.0 * r * S_1;
    G.h22 = -a2cs * S_1;
    G.h33 = -cs * (a2_r2 * S * S + a2 * s2 * rS * (a2_r2 + S)) * S_3;

    G.p01 = 2.0 * a * (r2 - a2c2) * DS_1 * S_1;
    G.p02 = 2.0 * -4.0 * a * rS * m_s * R_1;
    G.p13 = (a4 + 3. * a4 * r - 12. * a2r2 + 8. * a2 * r3 -
             16. * r4 + 8. * r2 * r3 + 4. * r * (dbl_r2 - r + a2) * a2cc -
             a4CC * (1. - r)) * DR_1;
    G.p23 = ((3. * a4 + 8. * a2 * r + 8. * a2r2 + 8. * r4 +
              4. * (dbl_r2 - 2. * r + a2) * a2cc + a4CC) * m_s) * R_1;
}

#if S5_FAST
// The connection in COMPACT form (fast variant of the step-wise integrator; round 6).  The reference writes the twenty entries as
// polynomials in cos 2 theta and cos 4 theta over R = (a^2 + 2 r^2 + a^2 cos 2 theta)^2 (ref src/sim5kerr.c:255-312; kept
// operation for operation in kerr_connection above: the strict variant's and the batch API's).  R is 4 Sigma^2, and every entry
// collapses to a few products of Sigma = r^2 + a^2 c^2, Delta = r^2 - 2 r + a^2, w = r^2 - a^2 c^2, A = r^2 + a^2:
//   t01 = 2 A w / (Sigma^2 Delta)                  t02 = -4 a^2 r s c / Sigma^2
//   t13 = 2 a s^2 [a^2 c^2 (a^2 - r^2) - r^2 (a^2 + 3 r^2)] / (Sigma^2 Delta)            t23 = -t02 a s^2
//   r00 = Delta w / Sigma^3     r03 = -2 a s^2 r00     r11 = (r a^2 s^2 - w) / (Sigma Delta)     r12 = -2 a^2 s c / Sigma
//   r22 = -r Delta / Sigma      r33 = -Delta s^2 [r Sigma^2 - a^2 s^2 w] / Sigma^3
//   h00 = -2 a^2 r s c / Sigma^3      h03 = 4 a r A s c / Sigma^3      h11 = a^2 s c / (Sigma Delta)      h12 = 2 r / Sigma
//   h22 = -a^2 s c / Sigma      h33 = -s c [A Sigma^2 + 2 a^2 r s^2 (A + Sigma)] / Sigma^3
//   p01 = 2 a w / (Sigma^2 Delta)     p02 = -4 a r cot / Sigma^2
//   p13 = 2 [Sigma (r Sigma + a^2 s^2) - 2 r^2 A] / (Sigma^2 Delta)      p23 = 2 cot (Sigma^2 + 2 a^2 r s^2) / Sigma^2
// (off-diagonal entries doubled, as the reference stores them: ref :237-243).  Each was checked against the Christoffel symbols
// of the Kerr metric symbolically and against the reference's expressions in long double on 20 000 random points
// (tests/tools/kerr_connection_compact.py).  ~105 vector instructions with the metric against ~190: the march kernel evaluates
// a connection 2.4 times per raytrace() call.  ONE reciprocal, 1 / (Sigma Delta), serves everything, the metric included.
template <bool WITH_METRIC>
S5_DEV void kerr_connection_compact(double a, double r, double m, Metric* g, Conn& G)
{
    const double c2 = m * m;
    const double s2 = 1. - c2;
    double s, inv_s;
    sqrt_rsqrt_pos(s2, s, inv_s);
    if (s2 == 0.0) { s = 0.0; inv_s = INFINITY; }
    const double cs = s * m;
    const double m_s = m * inv_s;
    const double a2 = a * a, r2 = r * r;
    const double A2 = a2 + r2;
    const double a2c2 = a2 * c2, a2s2 = a2 * s2;
    const double S = r2 + a2c2;
    const double D = A2 - 2. * r;
    const double w = r2 - a2c2;
    const double inv = mrcp(S * D);                 // 1 / (Sigma Delta)
    const double S_1 = D * inv;
    const double S_2 = S_1 * S_1;
    const double S_3 = S_2 * S_1;
    const double DS2 = inv * S_1;                   // 1 / (Sigma^2 Delta)
    const double SS = S * S;
    const double two_r = r + r;
    const double as2 = a * s2;
    const double rcs = r * cs;
    const double a2cs = a2 * cs;

    if (WITH_METRIC) {
        g->a = a; g->r = r; g->m = m;
        g->g00 = -1. + two_r * S_1;
        g->g11 = SS * inv;                          // Sigma / Delta
        g->g22 = S;
        g->g03 = -(two_r * as2) * S_1;
        g->g33 = s2 * (A2 - a * g->g03);
    }

    G.t01 = 2. * ((A2 * w) * DS2);
    G.t02 = -4. * (a2 * (rcs * S_2));
    G.t13 = (2. * as2) * ((a2c2 * (a2 - r2) - r2 * (a2 + 3. * r2)) * DS2);
    G.t23 = -G.t02 * as2;

    G.r00 = (D * w) * S_3;
    G.r03 = -2. * (as2 * G.r00);
    G.r11 = (r * a2s2 - w) * inv;
    G.r12 = -2. * (a2cs * S_1);
    G.r22 = -(r * D) * S_1;
    G.r33 = -(D * s2) * ((r * SS - a2s2 * w) * S_3);

    G.h00 = -2. * ((a2 * rcs) * S_3);
    G.h03 = 4. * ((a * rcs) * (A2 * S_3));
    G.h11 = a2cs * inv;
    G.h12 = two_r * S_1;
    G.h22 = -(a2cs * S_1);
    G.h33 = -cs * ((A2 * SS + (a2s2 * two_r) * (A2 + S)) * S_3);

    G.p01 = 2. * ((a * w) * DS2);
    G.p02 = -4. * ((a * r) * (m_s * S_2));
    G.p13 = 2. * ((S * (r * S + a2s2) - 2. * (r2 * A2)) * DS2);
    G.p23 = 2. * (m_s * ((SS + a2s2 * two_r) * S_2));
}

// metric and connection at one point from one set of sub-expressions and ONE reciprocal
S5_DEV void kerr_metric_connection(double a, double r, double m, Metric& g, Conn& G)
{
    kerr_connection_compact<true>(a, r, m, &g, G);
}
#endif

// expand to the dense [4][4][4] array of the SIM5 API (batch kerr_connection only)
S5_DEV void conn_to_dense(const Conn& G, double* out)
{
    for (int i = 0; i < 64; ++i) out[i] = 0.0;
#define S5_AT(i, j, k) out[(i) * 16 + (j) * 4 + (k)]
    S5_AT(0,0,1) = G.t01; S5_AT(0,0,2) = G.t02; S5_AT(0,1,3) = G.t13; S5_AT(0,2,3) = G.t23;
    S5_AT(1,0,0) = G.r00; S5_AT(1,0,3) = G.r03; S5_AT(1,1,1) = G.r11; S5_AT(1,1,2) = G.r12;
    S5_AT(1,2,2) = G.r22; S5_AT(1,3,3) = G.r33;
    S5_AT(2,0,0) = G.h00; S5_AT(2,0,3) = G.h03; S5_AT(2,1,1) = G.h11; S5_AT(2,1,2) = G.h12;
    S5_AT(2,2,2) = G.h22; S5_AT(2,3,3) = G.h33;
    S5_AT(3,0,1) = G.p01; S5_AT(3,0,2) = G.p02; S5_AT(3,1,3) = G.p13; S5_AT(3,2,3) = G.p23;
#undef S5_AT
}

// d k^j / d lambda = -G^j_ab k^a k^b over the stored triangle (ref src/sim5raytrace.c:151-156)
S5_DEV void geodesic_accel(const Conn& G, const double k[4], double out[4])
{
    double s;
    s = 0.0; s -= G.t01 * k[0] * k[1]; s -= G.t02 * k[0] * k[2]; s -= G.t13 * k[1] * k[3]; s -= G.t23 * k[2] * k[3];
    out[0] = s;
    s = 0.0; s -= G.r00 * k[0] * k[0]; s -= G.r03 * k[0] * k[3]; s -= G.r11 * k[1] * k[1];
    s -= G.r12 * k[1] * k[2]; s -= G.r22 * k[2] * k[2]; s -= G.r33 * k[3] * k[3];
    out[1] = s;
    s = 0.0; s -= G.h00 * k[0] * k[0]; s -= G.h03 * k[0] * k[3]; s -= G.h11 * k[1] * k[1];
    s -= G.h12 * k[1] * k[2]; s -= G.h22 * k[2] * k[2]; s -= G.h33 * k[3] * k[3];
    out[2] = s;
    s = 0.0; s -= G.p01 * k[0] * k[1]; s -= G.p02 * k[0] * k[2]; s -= G.p13 * k[1] * k[3]; s -= G.p23 * k[2] * k[3];
    out[3] = s;
}

// transport_rhs(G, k, k) without the symmetrisation: 0.5 G (k_j k_k + k_k k_j) = G (k_j k_k) exactly
// (doubling and halving are exact), so this is the same number as the reference's Gamma(G, k, k) for half
// the multiplications.  NOT the same rounding as geodesic_accel, which forms (G k_a) k_b.
#define S5_GK(Gjk, j, k) s -= (Gjk) * (K[j] * K[k])
S5_DEV void transport_self(const Conn& G, const double K[4], double out[4])
{
    double s;
    s = 0.0; S5_GK(G.t01, 0, 1); S5_GK(G.t02, 0, 2); S5_GK(G.t13, 1, 3); S5_GK(G.t23, 2, 3); out[0] = s;
    s = 0.0; S5_GK(G.r00, 0, 0); S5_GK(G.r03, 0, 3); S5_GK(G.r11, 1, 1); S5_GK(G.r12, 1, 2);
    S5_GK(G.r22, 2, 2); S5_GK(G.r33, 3, 3); out[1] = s;
    s = 0.0; S5_GK(G.h00, 0, 0); S5_GK(G.h03, 0, 3); S5_GK(G.h11, 1, 1); S5_GK(G.h12, 1, 2);
    S5_GK(G.h22, 2, 2); S5_GK(G.h33, 3, 3); out[2] = s;
    s = 0.0; S5_GK(G.p01, 0, 1); S5_GK(G.p02, 0, 2); S5_GK(G.p13, 1, 3); S5_GK(G.p23, 2, 3); out[3] = s;
}
#undef S5_GK

// -G^i_jk U^j V^k with the half weight for the doubled storage (ref :422-439)
#define S5_GT(Gjk, j, k) s -= 0.5 * (Gjk) * (U[j] * V[k] + U[k] * V[j])
S5_DEV void transport_rhs(const Conn& G, const double U[4], const double V[4], double out[4])
{
    double s;
    s = 0.0; S5_GT(G.t01, 0, 1); S5_GT(G.t02, 0, 2); S5_GT(G.t13, 1, 3); S5_GT(G.t23, 2, 3); out[0] = s;
    s = 0.0; S5_GT(G.r00, 0, 0); S5_GT(G.r03, 0, 3); S5_GT(G.r11, 1, 1); S5_GT(G.r12, 1, 2);
    S5_GT(G.r22, 2, 2); S5_GT(G.r33, 3, 3); out[1] = s;
    s = 0.0; S5_GT(G.h00, 0, 0); S5_GT(G.h03, 0, 3); S5_GT(G.h11, 1, 1); S5_GT(G.h12, 1, 2);
    S5_GT(G.h22, 2, 2); S5_GT(G.h33, 3, 3); out[2] = s;
    s = 0.0; S5_GT(G.p01, 0, 1); S5_GT(G.p02, 0, 2); S5_GT(G.p13, 1, 3); S5_GT(G.p23, 2, 3); out[3] = s;
}
#undef S5_GT

S5_DEV double dot(const double u[4], const double v[4], const Metric& g)        // ref :609-626
{
    return u[0] * v[0] * g.g00 + u[1] * v[1] * g.g11 + u[2] * v[2] * g.g22 +
           u[3] * v[3] * g.g33 + u[0] * v[3] * g.g03 + u[3] * v[0] * g.g03;
}

S5_DEV void normalize_to(double v[4], double norm, const Metric& g)             // ref :553-573
{
    double f = msqrt(mdiv(norm, dot(v, v, g)));
    v[0] *= f; v[1] *= f; v[2] *= f; v[3] *= f;
}

S5_DEV void clear_tetrad(Tetrad& t)
{
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) t.e[i][j] = 0.0;
}

S5_DEV void tetrad_zamo(const Metric& g, Tetrad& t)                             // ref :678-711
{
    clear_tetrad(t);
    t.e[0][0] = msqrt(mdiv(g.g33, sq(g.g03) - g.g33 * g.g00));
    t.e[0][3] = mdiv(-t.e[0][0] * g.g03, g.g33);
    t.e[1][1] = mdiv(1., msqrt(g.g11));
    t.e[2][2] = mdiv(-1., msqrt(g.g22));
    t.e[3][3] = mdiv(1., msqrt(g.g33));
    t.metric = g;
}

S5_DEV void tetrad_azimuthal(const Metric& g, double Omega, Tetrad& t)          // ref :766-814
{
    if (Omega == 0.0) { tetrad_zamo(g, t); return; }
    double g00 = g.g00, g33 = g.g33, g03 = g.g03;
    double U0 = msqrt(mdiv(-1.0, g00 + 2. * Omega * g03 + sq(Omega) * g33));
    double U3 = U0 * Omega;
    clear_tetrad(t);
    t.e[0][0] = U0;
    t.e[0][3] = U3;
    t.e[1][1] = msqrt(mdiv(1., g.g11));
    t.e[2][2] = -msqrt(mdiv(1., g.g22));
    double k1 = (g03 * U3 + g00 * U0);
    double k2 = (g33 * U3 + g03 * U0);
    t.e[3][0] = mdiv(-(k1 >= 0.0 ? +1.0 : -1.0) * k2,
                     msqrt((g33 * g00 - g03 * g03) * (g00 * U0 * U0 + g33 * U3 * U3 + 2.0 * g03 * U0 * U3)));
    t.e[3][3] = t.e[3][0] * mdiv(-k1, k2);
    t.metric = g;
}

S5_DEV void tetrad_surface(const Metric& g, double Omega, double V, double dhdr, Tetrad& t) // ref :818-921
{
    double g00 = g.g00, g11 = g.g11, g22 = g.g22, g33 = g.g33, g03 = g.g03;
    double S0r = mdiv(1.0, msqrt(g11 + g22 * sq(dhdr)));
    double S0h = S0r * dhdr;
    double ur = mdiv(mdiv(V, msqrt(1. - V * V)), msqrt(g11));
    double v = (V >= 0.0 ? +1.0 : -1.0) *
               msqrt(mdiv(sq(mdiv(ur, S0r)) * (-g00 - 2. * Omega * g03 - sq(Omega) * g33), 1. + sq(mdiv(ur, S0r))));
    t.e[0][0] = 1.0; t.e[0][1] = v * S0r; t.e[0][2] = v * S0h; t.e[0][3] = Omega;
    normalize_to(t.e[0], -1.0, g);
    t.e[1][0] = (v * t.e[0][0]);
    t.e[1][1] = (v * t.e[0][1] + mdiv(S0r, t.e[0][0]));
    t.e[1][2] = (v * t.e[0][2] + mdiv(S0h, t.e[0][0]));
    t.e[1][3] = (v * t.e[0][3]);
    normalize_to(t.e[1], 1.0, g);
    t.e[2][0] = 0.0; t.e[2][1] = dhdr; t.e[2][2] = -1.0; t.e[2][3] = 0.0;
    normalize_to(t.e[2], 1.0, g);
    t.e[3][0] = mdiv(-(g03 + g33 * Omega), g00 + g03 * Omega);
    t.e[3][1] = 0.0; t.e[3][2] = 0.0; t.e[3][3] = 1.0;
    normalize_to(t.e[3], 1.0, g);
    t.metric = g;
}

S5_DEV void bl2on(const double in[4], double out[4], const Tetrad& t)           // ref :926-944
{
    out[0] = -dot(t.e[0], in, t.metric);
    out[1] = +dot(t.e[1], in, t.metric);
    out[2] = +dot(t.e[2], in, t.metric);
    out[3] = +dot(t.e[3], in, t.metric);
}

S5_DEV void on2bl(const double in[4], double out[4], const Tetrad& t)           // ref :948-970
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += in[j] * t.e[j][i];
        out[i] = s;
    }
}

// r^1.5 as r*sqrt(r): two correctly rounded operations, within 1 ulp of pow(r,1.5)
S5_DEV double omega_kepler(double r, double a) { return mdiv(1., a + r * msqrt(r)); }       // ref :1037-1047

S5_DEV double ell_kepler(double r, double a)                                              // ref :1050-1072
{
    return (r * r - 2. * a * sqrt(r) + a * a) / (sqrt(r) * r - 2. * sqrt(r) + a);
}

S5_DEV double omega_from_ell(double ell, const Metric& g)                                 // ref :1101-1111
{
    return mdiv(-(g.g03 + ell * g.g00), g.g33 + ell * g.g03);
}

#if S5_FAST
// with x = sqrt(r) supplied by the caller.  With Omega = 1/d, d = a + r x, the reference's radicand
// 1 - (2/r)(1 - a Omega)^2 - (r^2 + a^2) Omega^2 is (d^2 - 2 r^2 - r^2 - a^2)/d^2 = r (r^2 - 3 r + 2 a x)/d^2 and 1 - Omega l is
// (d - l)/d:  g = x sqrt(r^2 - 3 r + 2 a x) / (r x + a - l) -- one square root and one division, no reciprocal for Omega.
// Same conditioning as the reference's form (both lose 2-3 digits where the radicand cancels, at the marginally stable orbit
// of a fast hole: 4e-13 against the exact value over 2e4 random (a, r, l), either way); NaN below the photon orbit as there.
S5_DEV double gfactor_kepler_x(double r, double x, double a, double l)
{
    S5_FPC_GFLUX
    return mdiv(x * msqrt(r * r - 3. * r + 2. * a * x), r * x + (a - l));
}
#endif

S5_DEV double gfactor_kepler(double r, double a, double l)                                // ref :1128-1141
{
#if S5_FAST
    // Omega_K = 1/(a + r^1.5) and 2/r from one reciprocal
    const double den = a + r * sqrt_pos(r);
    const double t = mrcp(den * r);
    const double Om = r * t;
    const double w = 1. - a * Om;
    return mdiv(msqrt(1. - (2. * den * t) * (w * w) - (r * r + a * a) * (Om * Om)), 1. - Om * l);
#else
    double Om = mdiv(1., a + r * msqrt(r));
    double w = 1. - a * Om;
    return mdiv(msqrt(1. - mdiv(2., r) * (w * w) - (r * r + a * a) * (Om * Om)), 1. - Om * l);
#endif
}

S5_DEV void photon_momentum(double a, double r, double m, double l, double q,
                            double r_sign, double m_sign, double k[4])                    // ref :1151-1213
{
    double a2 = a * a, l2 = l * l, r2 = r * r, m2 = m * m;
    double S = r2 + a2 * m2;
    double D = r2 - 2. * r + a2;
    double R = sq(r2 + a2 - a * l) - D * (sq(l - a) + q);
    double M = q - mdiv(l2 * m2, 1. - m2) + a2 * m2;
    if ((M < 0.0) && (-M < 1e-8)) M = 0.0;
    if ((R < 0.0) && (-R < 1e-8)) R = 0.0;
    if (M < 0.0) { k[0] = k[1] = k[2] = k[3] = NAN; return; }
    k[0] = mdiv(+1, S) * (-a * (a * (1. - m2) - l) + mdiv(r2 + a2, D) * (r2 + a2 - a * l));
    k[1] = mdiv(+1, S) * msqrt(R);
    k[2] = mdiv(+1, S) * msqrt(M);
    k[3] = mdiv(+1, S) * (-a + mdiv(l, 1. - m2) + mdiv(a, D) * (r2 + a2 - a * l));
    if (r_sign < 0.0) k[1] = -k[1];
    if (m_sign < 0.0) k[2] = -k[2];
}

S5_DEV void photon_motion_constants(double a, double r, double m, const double k[4],
                                    double& L, double& Q)                                 // ref :1217-1251
{
    double a2 = a * a, r2 = r * r;
    double s2 = 1. - m * m;
    double D = r2 - 2. * r + a2;
    double nf = k[3] / k[0];
    double nh = sq(k[2]) / sq(k[0]);
    double l = (-a * a2 + sq(a2) * nf + nf * sq(r2) + a * (D - r2) + a2 * nf * (2. * r2 - D * s2)) * s2 /
               (D - a * s2 * (a - a2 * nf + nf * (D - r2)));
    L = l;
    double t1 = a * (l - a * s2) + ((a2 + r2) * (a2 - a * l + r2)) / D;
    double t2 = sq(a2) - a * a2 * l + sq(r2) + a * l * (D - r2) + a2 * (2. * r2 - D * s2);
    Q = (t1 * t1) * (nh - (sq(D * m) * (sq(l) - a2 * s2)) / (-s2 * (t2 * t2)));
}

S5_DEV double carter_constant(const double k[4], const Metric& g)                         // ref :1255-1269
{
    double m2 = sq(g.m);
    double kt = k[0] * g.g00 + k[3] * g.g03;
    double kh = k[2] * g.g22;
    double kf = k[3] * g.g33 + k[0] * g.g03;
    return sq(kh) + sq(kf) * m2 / (1. - m2) - sq(g.a) * sq(kt) * m2;
}

} // namespace S5NS
