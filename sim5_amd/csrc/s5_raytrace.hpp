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

// Scheduling fence between the phases of a step: without it the compiler overlaps metric, connection
// and corrector arithmetic for ILP and runs out of registers (spills to scratch).
#define S5_FENCE() __builtin_amdgcn_sched_barrier(0)

S5_DEV double rel_diff(double a, double b) { return mdiv(fabs(b - a), fabs(b) + 1e-40); }   // ref :31

S5_DEV void rt_metric(const RayState& s, double r, double m, Metric& g)
{
    if (s.opt_gr) kerr_metric(s.bh_spin, r, m, g); else flat_metric(r, m, g);
}
S5_DEV void rt_connection(const RayState& s, double r, double m, Conn& G)
{
#if S5_FAST
    if (s.opt_gr) kerr_connection_compact<false>(s.bh_spin, r, m, nullptr, G); else flat_connection(r, m, G);
#else
    if (s.opt_gr) kerr_connection(s.bh_spin, r, m, G); else flat_connection(r, m, G);
#endif
}

S5_DEV void raytrace_prepare(double bh_spin, const double x[4], const double k[4], double precision,
                             int options, RayState& s)                        // ref :44-94
{
    s.opt_gr = !((options & 1) == 1);
    s.step_epsilon = S5_DIVC(msqrt(precision), 10.);
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
    transport_self(G, k, s.dk);
}

// Phase marks of one raytrace() call.  The step routines take a policy object and call ph.mark(PH_*) where one phase of the
// call ends; the default policy is empty (nothing is emitted).  The march kernel's instrumented build (-DS5_TORUS_DEBUG,
// k_torus.hip) passes a cycle counter: profiles/r06_torus_call_phases.json is the table of where a lone ray's ~15 000 cycles go.
enum : int { PH_STEPSIZE = 0, PH_PREDICT, PH_CONNECTION, PH_CORR1, PH_CORR2, PH_CORR3, PH_CHECK, PH_ACCEL,
             PH_RK4_HEAD, PH_RK4_STAGE1, PH_RK4_STAGE2, PH_RK4_STAGE3, PH_RK4_TAIL, PH_LOAD, PH_STORE_TRANSFER, PH_QUEUES, PH_N };
struct NoPhases { S5_DEV void mark(int) const {} };

// classical RK4 on (x,k), theta as the angle (ref :251-323).  The four stages run as one rolled loop
// (stage offsets 0, h, h, dl and weights 1, 2, 2, 1 selected by the wave-uniform stage index): one copy of
// the connection code, and only the latest stage plus the running sums are live -- 8+8 doubles instead of
// 32.  The sums are accumulated in the reference's order ((k1 + 2 k2) + 2 k3) + k4 and the stage-0
// operations with a zero offset / unit weight are exact, so every rounding is the reference's.
// LATE_TPHI (march kernel, fast variant): t and phi do not enter the connection, so a caller that keeps the ray's state in
// memory passes a functor that returns x[0] / x[3] when asked and the routine asks only at the very end (x[0], x[3] on entry are
// ignored): four registers less through the three stages.  Same expressions, same operands: the same numbers.
struct NoLateFetch { S5_DEV double operator()(int) const { return 0.0; } };
template <bool LATE_TPHI = false, class Fetch = NoLateFetch, class Phases = NoPhases>
S5_DEV void rk4_step(double x[4], double k[4], double dl, RayState& s, Metric& g, const Fetch& late = Fetch(), const Phases& ph = Phases())
{
    Conn G;
    double xp[4], ki[4], di[4], sx[4], sk[4];
    const double h = 0.5 * dl;
    const double kt0 = s.kt;
#if S5_FAST
    // The reference integrates the polar ANGLE: acos at the start, cos of every stage angle and of the result
    // (ref :269-298).  With m = cos(theta), sn = sin(theta) = sqrt(1 - m^2) (theta in [0, pi]) every cosine it needs is
    // cos(theta + d) = m cos d - sn sin d for a small offset d: one bounded sincos per stage, no acos.
    const double m0 = x[2];
    const double sn0 = msqrt(1. - m0 * m0);
    x[2] = 0.0;                                   // x[2] and xp[2] now hold the OFFSET from theta
#else
    x[2] = macos(x[2]);
#endif
#if S5_FAST
    // Stage 0 apart from the loop: its offsets are zero and its acceleration is the one the previous step left in s.dk, so
    // it is four additions of zero and two sums -- written out with the operations the rolled loop performed for stage == 0
    // (x + 0 * 0, k + 0 * 0, 0 + 1 * k: every rounding as before), and s.dk is dead from here on instead of being carried,
    // eight registers, through the three stages that evaluate a connection.  The position offsets of t and phi are never
    // formed: the connection does not depend on them.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ki[i] = k[i] + 0.0 * 0.0;
        di[i] = s.dk[i];
        sx[i] = 0.0 + 1.0 * ki[i];
        sk[i] = 0.0 + 1.0 * di[i];
    }
    xp[0] = xp[3] = 0.0;
    ph.mark(PH_RK4_HEAD);
#pragma unroll 1
    for (int stage = 1; stage < 4; ++stage) {
        const double off = (stage == 3) ? dl : 0.5 * dl;           // (0.5 dl is exact: formed here, not carried)
        const double wgt = (stage == 3) ? 1.0 : 2.0;
        xp[1] = x[1] + ki[1] * off; xp[2] = x[2] + ki[2] * off;
#pragma unroll
        for (int i = 0; i < 4; ++i) ki[i] = k[i] + di[i] * off;
        {
            double sd, cd;
            msincos_small(xp[2], sd, cd);
            rt_connection(s, xp[1], m0 * cd - sn0 * sd, G);
            transport_self(G, ki, di);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { sx[i] = sx[i] + wgt * ki[i]; sk[i] = sk[i] + wgt * di[i]; }
        ph.mark(PH_RK4_HEAD + stage);
    }
#else
#pragma unroll
    for (int i = 0; i < 4; ++i) { ki[i] = 0.0; di[i] = 0.0; sx[i] = 0.0; sk[i] = 0.0; }
#pragma unroll 1
    for (int stage = 0; stage < 4; ++stage) {
        const double off = (stage == 0) ? 0.0 : (stage == 3) ? dl : h;
        const double wgt = (stage == 1 || stage == 2) ? 2.0 : 1.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { xp[i] = x[i] + ki[i] * off; ki[i] = k[i] + di[i] * off; }
        rt_connection(s, xp[1], mcos(xp[2]), G);
        transport_self(G, ki, di);
#pragma unroll
        for (int i = 0; i < 4; ++i) { sx[i] = sx[i] + wgt * ki[i]; sk[i] = sk[i] + wgt * di[i]; }
    }
#endif
    if (LATE_TPHI) { x[0] = late(0); x[3] = late(3); }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[i] += S5_DIVC(dl, 6.) * sx[i];
        k[i] += S5_DIVC(dl, 6.) * sk[i];
    }
    // g: the KERR metric at the new point, also in flat mode (ref :305); returned for the caller's transfer step
#if S5_FAST
    {
        double sd, cd;
        msincos_small(x[2], sd, cd);
        x[2] = m0 * cd - sn0 * sd;
    }
    if (s.opt_gr) kerr_metric_connection(s.bh_spin, x[1], x[2], g, G);
    else { flat_connection(x[1], x[2], G); kerr_metric(s.bh_spin, x[1], x[2], g); }
    transport_self(G, k, s.dk);
#else
    x[2] = mcos(x[2]);
    rt_connection(s, x[1], x[2], G);
    transport_self(G, k, s.dk);
    kerr_metric(s.bh_spin, x[1], x[2], g);
#endif
    const double kt1 = k[0] * g.g00 + k[3] * g.g03;
    s.error = (float)rel_diff(kt1, kt0);
    ph.mark(PH_RK4_TAIL);
}

// Verlet part of one adaptive step (ref :109-245 up to the precision check :217-220).  Returns true if
// the step was accepted (x, k, s advanced); false if the reference would now fall back to RK4: x and k
// are then unchanged (restored) and `dl` holds the step size the fallback must use.  s.pass is counted
// either way, as in the reference.
// size of the next step (ref :164-166): a function of k and dk only, so a caller that defers the RK4
// fallback can recompute exactly the value the failed Verlet attempt used
S5_DEV double next_step_size(const double k[4], double step_cap, const RayState& s)
{
    const double tiny = 1e-40;
    const double* dk = s.dk;
#if S5_FAST
    // eps / (sum_i |dk_i| / (|k_i| + tiny) + tiny) over the common denominator: one division instead of five.
    // The product of the four denominators cannot leave the normal range for a null vector (k^t is O(1)); if it
    // ever did, the reference's form below is taken.
    const double d0 = fabs(k[0]) + tiny, d1 = fabs(k[1]) + tiny, d2 = fabs(k[2]) + tiny, d3 = fabs(k[3]) + tiny;
    const double d01 = d0 * d1, d23 = d2 * d3;
    const double den = d01 * d23;
    const double num = (fabs(dk[0]) * d1 + fabs(dk[1]) * d0) * d23 + (fabs(dk[2]) * d3 + fabs(dk[3]) * d2) * d01;
    double stepsize = mdiv(s.step_epsilon * den, num + tiny * den);
    if (!(den > 1e-250) || !(den < 1e250))
        stepsize = mdiv(s.step_epsilon,
            mdiv(fabs(dk[0]), d0) + mdiv(fabs(dk[1]), d1) + mdiv(fabs(dk[2]), d2) + mdiv(fabs(dk[3]), d3) + tiny);
#else
    const double stepsize = mdiv(s.step_epsilon,
        mdiv(fabs(dk[0]), fabs(k[0]) + tiny) + mdiv(fabs(dk[1]), fabs(k[1]) + tiny) +
        mdiv(fabs(dk[2]), fabs(k[2]) + tiny) + mdiv(fabs(dk[3]), fabs(k[3]) + tiny) + tiny);
#endif
    double dl = fmin(step_cap, stepsize);
    if (dl < 1e-3) dl = 1e-3;
    return dl;
}

template <class Phases = NoPhases>
S5_DEV bool verlet_attempt(double x[4], double k[4], double step_cap, double& dl, RayState& s, Metric& g, const Phases& ph = Phases())
{
    double xp[4], kh[4], kp[4], kq[4];
    const double* dk = s.dk;

    dl = next_step_size(k, step_cap, s);
    s.pass++;
    ph.mark(PH_STEPSIZE);

    const double half_dl = 0.5 * dl;
    const double half_dl2 = 0.5 * dl * dl;
    xp[0] = x[0] + k[0] * dl + dk[0] * half_dl2;
    xp[1] = x[1] + k[1] * dl + dk[1] * half_dl2;
#if S5_FAST
    {   // cos(acos(m) + d) = m cos d - sqrt(1 - m^2) sin d: one bounded sincos and a square root instead of acos
        // and cos (the reference's form, ref :177, stays in the strict variant); same step counts on the C4 job
        const double d = k[2] * dl + dk[2] * half_dl2;
        double sd, cd;
        msincos_small(d, sd, cd);
        xp[2] = x[2] * cd - msqrt(1. - x[2] * x[2]) * sd;
    }
#else
    xp[2] = mcos(macos(x[2]) + (k[2] * dl + dk[2] * half_dl2));
#endif
    xp[3] = x[3] + k[3] * dl + dk[3] * half_dl2;
#pragma unroll
    for (int i = 0; i < 4; ++i) kh[i] = k[i] + dk[i] * half_dl;          // the reference updates k in place
    ph.mark(PH_PREDICT);

    S5_FENCE();
    Conn G;
#if S5_FAST
    if (s.opt_gr) kerr_metric_connection(s.bh_spin, xp[1], xp[2], g, G);
    else { flat_metric(xp[1], xp[2], g); flat_connection(xp[1], xp[2], G); }
#else
    rt_metric(s, xp[1], xp[2], g);
    S5_FENCE();
    rt_connection(s, xp[1], xp[2], G);
#endif
    S5_FENCE();
#pragma unroll
    for (int i = 0; i < 4; ++i) kp[i] = kh[i] + dk[i] * half_dl;
    ph.mark(PH_CONNECTION);

#if S5_FAST
    // The corrector's convergence measure k_frac_error = sum_i |kp_i - kq_i| / (|kq_i| + 1e-40), accumulated in
    // float (ref :199-210), is only ever COMPARED (with 1e-5 to iterate again, with 1e-4 to reject the step,
    // ref :213,220).  The comparisons are made on the exact rational N/D > T as N > T D (no division); the float
    // accumulation differs from the exact sum by < 3e-7 relative, so whenever N/D is within 1e-6 of a threshold
    // the reference's own float sequence is evaluated and decides (`near`, rare).  Decisions are identical.
    bool again = true, reject = false;
#pragma unroll
    for (int iter = 0; iter < 3; ++iter) {
        if (again) {
            double acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) kq[i] = kp[i];
            transport_self(G, kq, acc);              // G (k_a k_b): the ten products once instead of (G k_a) k_b per entry
#pragma unroll
            for (int i = 0; i < 4; ++i) kp[i] = kh[i] + acc[i] * half_dl;
            const double tiny = 1e-40;
            const double d0 = fabs(kq[0]) + tiny, d1 = fabs(kq[1]) + tiny, d2 = fabs(kq[2]) + tiny, d3 = fabs(kq[3]) + tiny;
            const double n0 = fabs(kp[0] - kq[0]), n1 = fabs(kp[1] - kq[1]), n2 = fabs(kp[2] - kq[2]), n3 = fabs(kp[3] - kq[3]);
            const double d01 = d0 * d1, d23 = d2 * d3;
            const double D = d01 * d23;
            const double N = (n0 * d1 + n1 * d0) * d23 + (n2 * d3 + n3 * d2) * d01;
            const double T1 = 1e-2 * 1e-3, T2 = 1e-2 * 1e-2, slack = 1e-6;
            again = N > (T1 * (1. + slack)) * D;
            reject = N > (T2 * (1. + slack)) * D;
            const bool sure = (again || !(N >= (T1 * (1. - slack)) * D)) && (reject || !(N >= (T2 * (1. - slack)) * D)) &&
                              (D > 1e-250) && (D < 1e250) && (N == N);
            if (!sure) {
                float kerr = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) kerr = (float)((double)kerr + rel_diff(kp[i], kq[i]));
                again = (double)kerr > T1;
                reject = (double)kerr > T2;
            }
        }
        S5_FENCE();
        ph.mark(PH_CORR1 + iter);
    }
    const double kt = kp[0] * g.g00 + kp[3] * g.g03;
    const double kk = fabs(dot(kp, kp, g));
    s.error = (float)fmax(rel_diff(kt, s.kt), kk);
    ph.mark(PH_CHECK);
    if (reject || ((double)s.error > 1e-2 * 1e-2)) return false;
#else
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
                kp[i] = kh[i] + acc[i] * half_dl;
                kerr = (float)((double)kerr + rel_diff(kp[i], kq[i]));
            }
            again = (double)kerr > 1e-2 * 1e-3;
        }
        S5_FENCE();
    }

    const double kt = kp[0] * g.g00 + kp[3] * g.g03;
    const double kk = fabs(dot(kp, kp, g));
    s.error = (float)fmax(rel_diff(kt, s.kt), kk);
    if (((double)kerr > 1e-2 * 1e-2) || ((double)s.error > 1e-2 * 1e-2)) return false;
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[i] = xp[i]; k[i] = kp[i]; }
#if S5_FAST
    transport_self(G, kp, s.dk);
#else
    geodesic_accel(G, kp, s.dk);
#endif
    s.kt = kt;
    ph.mark(PH_ACCEL);
    return true;
}

// one adaptive step (ref :109-245); `step` in: cap on the step, out: step taken
S5_DEV void raytrace_step(double x[4], double k[4], double& step, RayState& s)
{
    double dl;
    Metric g;
    if (!verlet_attempt(x, k, step, dl, s, g)) rk4_step(x, k, dl, s, g);
    step = dl;
}

S5_DEV double raytrace_error(const double x[4], const double k[4], const RayState& s)   // ref :328-343
{
    Metric g;
    rt_metric(s, x[1], x[2], g);
    return rel_diff(s.Q, carter_constant(k, g));
}

} // namespace S5NS
