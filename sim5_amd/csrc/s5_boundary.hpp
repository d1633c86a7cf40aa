// s5_boundary.hpp -- the rest of the public SIM5 prototypes of the headers SURVEY.md 8(b) cites, as gfx950 device
// code: the routines a SIM5 caller of the hot path may link that the image / march kernels themselves do not inline.
// Restated from the reference in its operation order (strict arithmetic: these serve the batch entry points only);
// the line ranges are given at each routine.
//   ref: src/sim5kerr.c (metric helpers, Gamma, vectors, tetrad_general/_radial, epicyclic frequencies, four-velocities),
//        src/sim5kerr-geod.c:413-457, src/sim5elliptic.c:226-474 (Legendre integrals by angle / sine),
//        src/sim5radiation.c:53-99.
#pragma once
#include "s5_azimuth.hpp"

namespace S5NS {

S5_DEV void flat_metric_contravariant(double r, double m, Metric& g)                     // ref src/sim5kerr.c:54-71
{
    g.a = 0.0; g.r = r; g.m = m;
    g.g00 = -1.0; g.g11 = +1.0; g.g22 = +1. / (r * r); g.g33 = +1. / (r * r) / (1. - m * m); g.g03 = 0.0;
}

// Kerr-Newman metric, covariant and contravariant, and connection (charge Q): the Kerr closed forms with 2r -> 2r - Q^2 and
// Delta -> Delta + Q^2 (ref src/sim5kerr.c:136-194, 321-397); same storage of the connection as kerr_connection
// (20 entries, j <= k, off-diagonals pre-doubled: s5_kerr.hpp Conn)
S5_DEV void kerr_newman_metric(double a, double Q, double r, double m, Metric& g)       // ref src/sim5kerr.c:136-163
{
    const double rQ = sq(Q), r2 = sq(r), a2 = sq(a), m2 = sq(m);
    const double S = r2 + a2 * m2;
    const double s2_S = (1.0 - m2) / S;
    g.a = a; g.r = r; g.m = m;
    g.g00 = -1. + (2.0 * r - rQ) / S;
    g.g11 = S / (r2 - 2. * r + a2 + rQ);
    g.g22 = S;
    g.g33 = ((a2 + r2) * S + (2. * r - rQ) * a2 * s2_S * S) * s2_S;
    g.g03 = -a * (2. * r - rQ) * s2_S;
}

S5_DEV void kerr_newman_metric_contravariant(double a, double Q, double r, double m, Metric& g)   // ref :168-194
{
    const double rQ = sq(Q), r2 = sq(r), a2 = sq(a), m2 = sq(m);
    const double S = r2 + a2 * m2;
    const double SD = S * (r2 - 2. * r + a2 + rQ);
    g.a = a; g.r = r; g.m = m;
    g.g00 = -sq(r2 + a2) / SD + a2 * (1. - m2) / S;
    g.g11 = (r2 - 2. * r + a2 + rQ) / S;
    g.g22 = 1. / S;
    g.g33 = 1. / S / (1. - m2) - a2 / SD;
    g.g03 = a * (-2. * r + rQ) / SD;
}

S5_DEV void kerr_newman_connection(double a, double Q, double r, double m, Conn& G)     // ref :321-397
{
    const double rS = 2.0 * r;
    const double rQ = sq(Q);
    const double s = sqrt(1. - m * m);
    const double cs = s * m;
    const double c2 = m * m;
    const double s2 = s * s;
    const double cc = c2 - s2;
    const double CC = 8. * c2 * c2 - 8. * c2 + 1.;
    const double a2 = a * a;
    const double a4 = a2 * a2;
    const double a2cc = a2 * cc;
    const double a2c2 = a2 * c2;
    const double a2cs = a2 * cs;
    const double r2 = r * r;
    const double r3 = r2 * r;
    const double a2_r2 = a2 + r2;
    const double R = pow(a2 + 2. * r2 + a2cc, 2.);
    const double D = r2 - 2. * r + a2 + rQ;
    const double S = r2 + a2c2;
    const double S_1 = 1. / S;
    const double S_3 = 1. / (S * S * S);
    const double R_1 = 1. / R;
    const double m_s = m / s;
    const double DR_1 = R_1 / D;
    const double DS_1 = S_1 / D;
    const double dbl_r2 = 2. * r2;

    G.t01 = 2.0 * 4.0 * (a2_r2) * (r * (r - rQ) - a2c2) * DR_1;
    G.t02 = 2.0 * -4.0 * a2cs * (rS - rQ) * R_1;
    G.t13 = 2.0 * 4.0 * a * s2 * (-a2 * (r2 - r * rQ) - r3 * (3. * r - 2. * rQ) + a2cc * (a2 - r2 + r * rQ)) * DR_1;
    G.t23 = -G.t02 * s2 * a;

    G.r00 = D * (r * (r - rQ) - a2c2) * S_3;
    G.r03 = -2.0 * G.r00 * a * s2;
    G.r11 = (r * (a2 - r + rQ) + a2 * (1. - r) * c2) * DS_1;
    G.r12 = -2.0 * a2cs * S_1;
    G.r22 = -r * D * S_1;
    G.r33 = -D * s2 * (2. * a2c2 * r3 + r2 * r3 + a2 * a2c2 * s2 + a2c2 * a2c2 * r - a2 * r * (r - rQ) * s2) * S_3;

    G.h00 = -(2.0 * r - rQ) * a2cs * S_3;
    G.h03 = 2.0 * -G.h00 * a2_r2 / a;
    G.h11 = +a2cs * DS_1;
    G.h12 = 2.0 * r * S_1;
    G.h22 = -a2cs * S_1;
    G.h33 = -cs * (a2_r2 * S * S + a2 * s2 * (rS - rQ) * (a2_r2 + S)) * S_3;

    G.p01 = 2.0 * a * (r * (r - rQ) - a2c2) * DS_1 * S_1;
    G.p02 = 2.0 * -4.0 * a * (rS - rQ) * m_s * R_1;
    G.p13 = 2.0 * 4.0 * (r3 * (r2 - rS + rQ) + r * a2c2 * a2c2 -
            a2 * r * (r - rQ) * s2 + a2c2 * r * (dbl_r2 - rS + rQ) + a2c2 * a2 * s2) * DR_1;
    G.p23 = 2.0 * ((3. * a4 + 8. * a2 * r + 8. * a2 * r2 + 8. * r2 * r2 +
            4. * (dbl_r2 - rS + rQ + a2) * a2cc + a4 * CC) * m_s) * (R_1 / 2.0);
}

// -G^i_(jk) U^j V^k for a DENSE connection handed in by the caller (any values, not only kerr_connection's): the
// reference's loop over j <= k with the pair symmetrised (ref :422-440)
S5_DEV void gamma_dense(const double* G, const double U[4], const double V[4], double out[4])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = j; k < 4; ++k) s -= 0.5 * G[16 * i + 4 * j + k] * (U[j] * V[k] + U[k] * V[j]);
        out[i] = s;
    }
}

// metric == nullptr: Minkowski, as everywhere in the reference's vector helpers
S5_DEV double dot_or_flat(const double u[4], const double v[4], const Metric* g)         // ref :609-626
{
    return g ? dot(u, v, *g) : (-u[0] * v[0] + u[1] * v[1] + u[2] * v[2] + u[3] * v[3]);
}

S5_DEV void vector_covariant(const double v[4], double out[4], const Metric* g)          // ref :477-500
{
    if (g) {
        out[0] = v[0] * g->g00 + v[3] * g->g03;
        out[1] = v[1] * g->g11;
        out[2] = v[2] * g->g22;
        out[3] = v[3] * g->g33 + v[0] * g->g03;
    } else {
        out[0] = -v[0]; out[1] = +v[1]; out[2] = +v[2]; out[3] = +v[3];
    }
}

S5_DEV double vector_norm(const double v[4], const Metric* g) { return sqrt(dot_or_flat(v, v, g)); }       // ref :504-516
S5_DEV double vector_3norm(const double v[4]) { return sqrt(v[1] * v[1] + v[2] * v[2] + v[3] * v[3]); }   // ref :521-532

// time component set to V0, the spatial part rescaled so that the vector stays null (ref :577-605)
S5_DEV void vector_norm_to_null(double v[4], double V0, const Metric* g)
{
    double alpha;
    if (g) {
        const double a = v[1] * v[1] * g->g11 + v[2] * v[2] * g->g22 + v[3] * v[3] * g->g33;
        const double b = V0 * v[3] * g->g03;
        const double c = V0 * V0 * g->g00;
        alpha = fmax(-b / a + sqrt(b * b - a * c) / a, -b / a - sqrt(b * b - a * c) / a);
    } else {
        const double a = v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        const double c = -V0 * V0;
        alpha = sqrt(-c / a);
    }
    v[0] = V0; v[1] *= alpha; v[2] *= alpha; v[3] *= alpha;
}

// tetrad of an observer with a general four-velocity (ref :630-674).  e[2][1] carries N1 where the pattern of the
// other rows suggests N2: the reference's expression, kept.
S5_DEV void tetrad_general(const Metric& g, const double U[4], Tetrad& t)
{
    double u[4];
    const double D = sq(g.r) - 2 * (g.r) + sq(g.a);
    vector_covariant(U, u, &g);
    const double N1 = sqrt(-g.g11 * (u[0] * U[0] + u[3] * U[3]) * (1. + u[2] * U[2]));
    const double N2 = sqrt(+g.g22 * (1. + u[2] * U[2]));
    const double N3 = sqrt(-(u[0] * U[0] + u[3] * U[3]) * D * (1. - sq(g.m)));
    t.e[0][0] = U[0]; t.e[0][1] = U[1]; t.e[0][2] = U[2]; t.e[0][3] = U[3];
    t.e[1][0] = u[1] * U[0] / N1;
    t.e[1][1] = -(u[0] * U[0] + u[3] * U[3]) / N1;
    t.e[1][2] = 0.0;
    t.e[1][3] = u[1] * U[3] / N1;
    t.e[2][0] = u[2] * U[0] / N2;
    t.e[2][1] = u[2] * U[0] / N1;
    t.e[2][2] = (1. + u[2] * U[2]) / N2;
    t.e[2][3] = u[2] * U[3] / N2;
    t.e[3][0] = -u[0] / N3;
    t.e[3][1] = 0.0;
    t.e[3][2] = 0.0;
    t.e[3][3] = +u[3] / N3;
    t.metric = g;
}

// tetrad of an observer moving in the radial direction only (ref :715-762)
S5_DEV void tetrad_radial(const Metric& g, double v_r, Tetrad& t)
{
    if (v_r == 0.0) { tetrad_zamo(g, t); return; }             // ref :732
    const double g00 = g.g00, g11 = g.g11;
    const double U0 = sqrt((-1. - sq(v_r) * g11) / g00);
    const double U1 = v_r;
    clear_tetrad(t);
    t.e[0][0] = U0;
    t.e[0][1] = U1;
    const double UG = U0 * U0 * g00 + U1 * U1 * g11;
    t.e[1][0] = -U1 * sqrt(UG * g11 * g00) * U0 / (g11 * UG) * g11 / (U0 * g00);
    t.e[1][1] = sqrt(UG * g11 * g00) * U0 / (g11 * UG);
    t.e[2][2] = -1. / sqrt(g.g22);
    t.e[3][3] = 1. / sqrt(g.g33);
    t.metric = g;
}

// the reference's OmegaK: pow(r, 1.5) (ref :1037-1047)
S5_DEV double omega_kepler_pow(double r, double a) { return 1. / (a + pow(r, 1.5)); }

S5_DEV double omega_r(double r, double a)                                                 // ref :1076-1085
{
    return omega_kepler_pow(r, a) * sqrt(1. - 6. / r + 8. * a / sqrt(r * r * r) - 3. * a * a / sq(r));
}

S5_DEV double omega_z(double r, double a)                                                 // ref :1088-1098
{
    return omega_kepler_pow(r, a) * sqrt(1. - 4. * a / sqrt(r * r * r) + 3. * a * a / sq(r));
}

S5_DEV double ell_from_omega(double Omega, const Metric& g)                               // ref :1114-1124
{
    return -(g.g03 + g.g33 * Omega) / (g.g00 + g.g03 * Omega);
}

S5_DEV void fourvelocity_zamo(const Metric& g, double U[4])                               // ref :1278-1291
{
    U[0] = sqrt(g.g33 / (sq(g.g03) - g.g33 * g.g00));
    U[1] = 0.0; U[2] = 0.0;
    U[3] = -U[0] * g.g03 / g.g33;
}

S5_DEV void fourvelocity_azimuthal(double Omega, const Metric& g, double U[4])            // ref :1295-1309
{
    U[0] = sqrt(-1.0 / (g.g00 + 2. * Omega * g.g03 + sq(Omega) * g.g33));
    U[1] = 0.0; U[2] = 0.0;
    U[3] = U[0] * Omega;
}

S5_DEV void fourvelocity_radial(double vr, const Metric& g, double U[4])                  // ref :1313-1327
{
    U[0] = sqrt((-1.0 - sq(vr) * g.g11) / g.g00);
    U[1] = vr; U[2] = 0.0; U[3] = 0.0;
}

S5_DEV double fourvelocity_norm(double U1, double U2, double U3, const Metric& g)         // ref :1331-1338
{
    const double D = sq(g.g03 * U3) - g.g00 * g.g11 * sq(U1) - g.g00 * g.g22 * sq(U2) - g.g00 * g.g33 * sq(U3) - g.g00;
    return (-g.g03 * U3 - sqrt(D)) / g.g00;
}

S5_DEV void fourvelocity(double U1, double U2, double U3, const Metric& g, double U[4])   // ref :1342-1353
{
    const double N = fourvelocity_norm(U1, U2, U3, g);
    U[0] = 1. / N; U[1] = U1 / N; U[2] = U2 / N; U[3] = U3 / N;
}

// sign of k^theta at position integral P (ref src/sim5kerr-geod.c:413-457): the bookkeeping of geodesic_dm_sign
// with the opposite sign returned; NaN for RR_DBL / RR_BH
S5_DEV double position_pol_sign_k_theta(const Geod& g, double P)
{
    if (!escapes(g)) return NAN;
    double T;
    return (polar_phase(g, P, T) < 0) ? +1 : -1;               // dk[2] = -d(m), ref :441
}

// F(phi, m) for any real phi (ref src/sim5elliptic.c:235-252)
S5_DEV double ell_F(double phi, double m)
{
    if (m == 1.0) m = 0.99999999;
    if (phi == 0.0) return 0.0;
    int k = 0;
    for (int guard = 0; guard < 100000 && fabs(phi) > M_PI / 2.; ++guard) { (phi > 0) ? k++ : k--; phi += (phi > 0) ? -M_PI : +M_PI; }
    const double s2 = pow(sin(phi), 2);
    double f = (phi > 0 ? +1 : -1) * sqrt(s2) * carlson_rf(1 - s2, 1.0 - s2 * m, 1.0);
    if (k != 0) f += 2. * k * ell_K(m);
    return f;
}

// E(asin(s), m), 0 <= s <= 1 (ref :339-357)
S5_DEV double ell_E_sin(double s, double m)
{
    if (m == 1.0) m = 0.99999999;
    if (s == 0.0) return 0.0;
    const double s2 = s * s;
    const double c2 = 1.0 - s2;
    const double q = 1.0 - s2 * m;
    return s * (carlson_rf(c2, q, 1.0) - sq(s * sqrt(m)) * carlson_rd(c2, q, 1.0) / 3.0);
}

// Pi(asin(s), n, m), 0 <= s <= 1 (ref :453-474)
S5_DEV double ell_Pi_sin(double s, double n, double m)
{
    const double s2 = s * s;
    if (isinf(n)) return 0.0;
    if (m == 1.0) m = 0.99999999;
    if (s == 0.0) return 0.0;
    if (s == 1.0) return ell_Pi_complete(n, m);
    const double c2 = 1.0 - s2;
    const double ns2 = -n * s2;
    const double q = 1.0 - s2 * m;
    return s * (carlson_rf(c2, q, 1.0) - ns2 * carlson_rj(c2, q, 1.0, 1.0 + ns2) / 3.0);
}

// Pi(phi, n, m) for any real phi, complex for the hyperbolic case n > 1 (ref :382-423); out = {re, im}
S5_DEV void ell_Pi(double phi, double n, double m, double out[2])
{
    out[0] = 0.0; out[1] = 0.0;
    if (isinf(n)) return;
    if (m == 1.0) m = 0.99999999;
    if (phi == 0.0) return;
    const bool hyperbolic = (n > 1.0);
    double p = 0, dn = 0, la = 0;
    if (hyperbolic) {
        p = sqrt((n - 1.) * (1. - m / n));
        dn = sqrt(1. - m * sin(phi) * sin(phi));
        la = (dn + p * tan(phi)) / (dn - p * tan(phi));
        n = m / n;
    }
    int k = 0;
    for (int guard = 0; guard < 100000 && fabs(phi) > M_PI / 2.; ++guard) { (phi > 0) ? k++ : k--; phi += (phi > 0) ? -M_PI : +M_PI; }
    const double s = sin(phi);
    const double c2 = 1.0 - s * s;
    const double q = 1.0 - s * s * m;
    const double ns2 = -n * s * s;
    double re = s * (carlson_rf(c2, q, 1.0) - ns2 * carlson_rj(c2, q, 1.0, 1.0 + ns2) / 3.0), im = 0.0;
    if ((k != 0) && (!hyperbolic)) re += 2. * k * ell_Pi_complete(n, m);
    if (hyperbolic) {
        re = -re + ell_F(phi, m) + log(fabs(la)) / (2 * p);
        im = (la < 0) ? M_PI / (2 * p) : 0.0;
    }
    out[0] = re; out[1] = im;
}

// black-body specific intensity, array form: the two factors are formed once per spectrum and each energy is
// BB1 E^3 / expm1(BB2 E) (ref src/sim5radiation.c:53-78; blackbody_Iv, :27-49, is the per-energy form with another
// operation order).  The caller computes (BB1, BB2) per spectrum with blackbody_factors().
S5_DEV void blackbody_factors(double T, double hardf, double cos_mu, double& BB1, double& BB2)
{
    const double planck_h = 6.626069e-27, speed_of_light = 2.997925e+10, boltzmann_k = 1.380650e-16, kev2freq = 2.417990e+17;
    const double limbf = (cos_mu >= 0.0) ? 0.5 + 0.75 * cos_mu : 1.0;
    BB1 = limbf * 2.0 * planck_h / sq(speed_of_light) / (hardf * hardf * hardf * hardf) * (kev2freq * kev2freq * kev2freq * kev2freq);
    BB2 = (planck_h * kev2freq) / (boltzmann_k * hardf * T);
}

S5_DEV double blackbody_photons(double T, double hardf, double cos_mu, double E)          // ref :83-92
{
    const double kev2erg = 1.602177e-09;
    return blackbody_Iv(T, hardf, cos_mu, E) / (E * kev2erg);
}

S5_DEV double blackbody_photons_total(double T, double hardf)                             // ref :96-114
{
    const double planck_h = 6.626069e-27, boltzmann_k = 1.380650e-16, speed_of_light2 = 8.987554e+20;
    return M_PI * 4.808227612 * (T * T * T) * (boltzmann_k * boltzmann_k * boltzmann_k) / (planck_h * planck_h * planck_h) /
           speed_of_light2 / hardf;
}

} // namespace S5NS
