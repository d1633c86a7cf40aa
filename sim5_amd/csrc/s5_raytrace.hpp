// s5_raytrace.hpp -- step-wise null-geodesic integrator (velocity Verlet after Dolence+2009 with an
// RK4 fallback), gfx950 device code.  Restated from the reference
// (ref: /root/reference/src/sim5raytrace.c:44-94 prepare, :109-245 step, :251-323 RK4, :328-343 error).
//
// The integrator state is the reference's raytrace_data (144 B) kept in registers; the connection
// is the 20-entry register form of s5_kerr.hpp.  The two error accumulators are single precision
// exactly as in the reference (`float k_frac_error`, `float error`): their thresholds decide the
// corrector trip count and the RK4 fallback, hence the step sequence.
#pragma once
#include "s5_kerr.hpp"

namespace S5NS {

// byte-identical to sim5gpu_raytrace_data / the reference's struct raytrace_data
struct RayState {
    int opt_gr, opt_pol;
    double step_epsilon;
    double bh_spin, E, Q;
    double WP[2];
    int pass, refines;
    double dk[4], df[4];
    double kt;
    float error;
};
static_assert(sizeof(RayState) == 144, "raytrace_data must keep the SIM5 layout");

S5_DEV double rel_diff(double a, double b) { return fabs(b - a) / (fabs(b) + 1e-40); }   // ref :31

S5_DEV void rt_metric(const RayState& s, double r, double m, Metric& g)
{
    if (s.opt_gr) kerr_metric(s.bh_spin, r, m, g); else flat_metric(r, m, g);
}
S5_DEV void rt_connection(const RayState& s, double r, double m, Conn& G)
{
    if (s.opt_gr) kerr_connection(s.bh_spin, r, m, G); else flat_connection(r, m, G);
}

S5_DEV void raytrace_prepare(double bh_spin, const double x[4], const double k[4], double precision,
                             int options, RayState& s)                        // ref :44-94
{
    s.opt_gr = !((options & 1) == 1);
    s.step_epsilon = sqrt(precision) / 10.;
    s.bh_spin = bh_spin;
    Metric g;
    Conn G;
    rt_metric(s, x[1], x[2], g);
    rt_connection(s, x[1], x[2], G);
    s.E = k[0] * g.g00 + k[3] * g.g03;
    s.Q = carter_constant(k, g);
    s.pass = 0;
    s.refines = 0;
    s.kt = s.E;
    s.error = 0.0f;
    transport_rhs(G, k, k, s.dk);
}

// classical RK4 on (x,k), theta as the angle (ref :251-323)
S5_DEV void rk4_step(double x[4], double k[4], double dl, RayState& s)
{
    Conn G;
    double xp[4], k1[4], d1[4], k2[4], d2[4], k3[4], d3[4], k4[4], d4[4];
    const double h = 0.5 * dl;
    const double kt0 = s.kt;
    x[2] = acos(x[2]);
#pragma unroll
    for (int i = 0; i < 4; ++i) { xp[i] = x[i]; k1[i] = k[i]; }
    rt_connection(s, xp[1], cos(xp[2]), G);
    transport_rhs(G, k1, k1, d1);
#pragma unroll
    for (int i = 0; i < 4; ++i) { xp[i] = x[i] + k1[i] * h; k2[i] = k[i] + d1[i] * h; }
    rt_connection(s, xp[1], cos(xp[2]), G);
    transport_rhs(G, k2, k2, d2);
#pragma unroll
    for (int i = 0; i < 4; ++i) { xp[i] = x[i] + k2[i] * h; k3[i] = k[i] + d2[i] * h; }
    rt_connection(s, xp[1], cos(xp[2]), G);
    transport_rhs(G, k3, k3, d3);
#pragma unroll
    for (int i = 0; i < 4; ++i) { xp[i] = x[i] + k3[i] * dl; k4[i] = k[i] + d3[i] * dl; }
    rt_connection(s, xp[1], cos(xp[2]), G);
    transport_rhs(G, k4, k4, d4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[i] += dl / 6. * (k1[i] + 2. * k2[i] + 2. * k3[i] + k4[i]);
        k[i] += dl / 6. * (d1[i] + 2. * d2[i] + 2. * d3[i] + d4[i]);
    }
    x[2] = cos(x[2]);
    rt_connection(s, x[1], x[2], G);
    transport_rhs(G, k, k, s.dk);
    Metric g;
    kerr_metric(s.bh_spin, x[1], x[2], g);          // Kerr metric also in flat mode, ref :302
    const double kt1 = k[0] * g.g00 + k[3] * g.g03;
    s.error = (float)rel_diff(kt1, kt0);
}

// one adaptive step (ref :109-245); `step` in: cap on the step, out: step taken
S5_DEV void raytrace_step(double x[4], double k[4], double& step, RayState& s)
{
    const double tiny = 1e-40;
    double x0[4], k0[4], xp[4], kp[4], kq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { x0[i] = x[i]; k0[i] = k[i]; }
    const double* dk = s.dk;

    const double stepsize = s.step_epsilon /
        (fabs(dk[0]) / (fabs(k[0]) + tiny) + fabs(dk[1]) / (fabs(k[1]) + tiny) +
         fabs(dk[2]) / (fabs(k[2]) + tiny) + fabs(dk[3]) / (fabs(k[3]) + tiny) + tiny);
    double dl = fmin(step, stepsize);
    if (dl < 1e-3) dl = 1e-3;
    s.pass++;

    const double half_dl = 0.5 * dl;
    const double half_dl2 = 0.5 * dl * dl;
    xp[0] = x[0] + k[0] * dl + dk[0] * half_dl2;
    xp[1] = x[1] + k[1] * dl + dk[1] * half_dl2;
    xp[2] = cos(acos(x[2]) + (k[2] * dl + dk[2] * half_dl2));
    xp[3] = x[3] + k[3] * dl + dk[3] * half_dl2;
#pragma unroll
    for (int i = 0; i < 4; ++i) k[i] += dk[i] * half_dl;

    Metric g;
    Conn G;
    rt_metric(s, xp[1], xp[2], g);
    rt_connection(s, xp[1], xp[2], G);
#pragma unroll
    for (int i = 0; i < 4; ++i) kp[i] = k[i] + dk[i] * half_dl;

    float kerr = 0.0f;
    bool again = true;
#pragma unroll
    for (int iter = 0; iter < 3; ++iter) {
        if (again) {
            kerr = 0.0f;
            double acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) kq[i] = kp[i];
            geodesic_accel(G, kq, acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                kp[i] = k[i] + acc[i] * half_dl;
                kerr = (float)((double)kerr + rel_diff(kp[i], kq[i]));
            }
            again = (double)kerr > 1e-2 * 1e-3;
        }
    }

    const double kt = kp[0] * g.g00 + kp[3] * g.g03;
    const double kk = fabs(dot(kp, kp, g));
    s.error = (float)fmax(rel_diff(kt, s.kt), kk);
    if (((double)kerr > 1e-2 * 1e-2) || ((double)s.error > 1e-2 * 1e-2)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = x0[i]; k[i] = k0[i]; }
        rk4_step(x, k, dl, s);
        step = dl;
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[i] = xp[i]; k[i] = kp[i]; }
    geodesic_accel(G, kp, s.dk);
    s.kt = kt;
    step = dl;
}

S5_DEV double raytrace_error(const double x[4], const double k[4], const RayState& s)   // ref :328-343
{
    Metric g;
    rt_metric(s, x[1], x[2], g);
    return rel_diff(s.Q, carter_constant(k, g));
}

} // namespace S5NS
