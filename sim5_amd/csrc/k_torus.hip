// k_torus.hip -- step-wise (Verlet) ray tracer with radiative transfer through an optically thin
// torus, gfx950.
//
// Two kernels:
//  (A) torus_start_kernel: one lane per pixel sets up the ray where it enters the integration
//      domain: geodesic_init_inf -> geodesic_P_int(r0, before pericentre) -> geodesic_position_pol
//      -> geodesic_momentum -> raytrace_prepare (ref src/sim5kerr-geod.c:42,179,363,787;
//      src/sim5raytrace.c:44).  The state goes to HBM as a structure of arrays (one 8-B column
//      per quantity, so every load/store below is a coalesced 512-B wave access).
//  (B) torus_pool_kernel: persistent waves advance rays with raytrace() (ref src/sim5raytrace.c:109-245),
//      accumulating the transfer integral after each accepted step; each workgroup keeps a pool of 320 rays in
//      LDS from which its waves run the Verlet half and the RK4 half of raytrace() as separate full-width
//      batches (see the comment at the kernel).
//
// Transfer model (the reference has no transfer integrator nor torus; this is
// this project's definition, stated in DESIGN.md): fluid on circular orbits with constant specific
// angular momentum ell; density rho = exp(-((R-R_t)^2 + z^2)/(2 w^2)), R = r sin(theta),
// z = r cos(theta) (shape 0), or rho = 1 inside a sphere of radius w (shape 1, analytic test case);
// with k normalised to k_t = -1 at infinity the local photon energy is -k.U = 1/g, the proper length
// of a step of affine size dl is dl/g, and per accepted step
//      dtau = absorb0 * rho * dl / g,     dI = g^4 * emis0 * rho * exp(-tau) * dl / g .
#include <mutex>
#include "s5_disk.hpp"
#include "s5_raytrace.hpp"
#include "k_torus.hpp"

namespace S5NS {

using namespace s5abi;

enum : int { COL_X0 = 0, COL_X1, COL_X2, COL_X3, COL_K0, COL_K1, COL_K2, COL_K3,
             COL_DK0, COL_DK1, COL_DK2, COL_DK3, COL_KT, COL_Q, NCOL };

struct RayCols {                 // start state, structure of arrays: column c of ray i is d[c * cap + i]
    double* d;
    size_t cap;
};

// START_AGAIN (ok[] value, fast job): the start of this ray is formed again by the STRICT variant's kernel (redo = 1).
// The reference forms k^r = sqrt(R)/S and k^theta = sqrt(M)/S with R and M as differences (src/sim5kerr.c:1176-1177); where a
// ray starts AT a turning point they cancel -- a ray of the round-6 campaign had M = 5.6e-11 from terms of 88: one unit in the
// last place of mu moves sqrt(M) by 1e-3 -- and what comes out is the rounding pattern of whoever formed mu.  The march then
// carries the difference on: two rays of 190 615 random ones ended 2e-5 .. 3e-5 from the CPU loop in the fast variant, at identical
// call counts and well-conditioned from the checker's own start state on (tests/tools/fuzz_torus_case.py, torus_ray_steps.py:
// the whole difference is in k^theta after the first call).  The strict variant has the reference's mu and M bit for bit; so
// the fast job starts the few rays whose R or M came out below 1e-6 of its terms from the strict variant's bits.
constexpr int START_AGAIN = 2;

__global__ __launch_bounds__(256, 2)
void torus_start_kernel(TorusParams p, RayCols st, int* __restrict__ ok, int* __restrict__ order,
                        unsigned long long* __restrict__ counters, const int redo)
{
    double* __restrict__ cols = st.d;
    const size_t i_raw = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i_raw < p.nrays;              // (no early exit: the workgroup meets at a barrier below)
    const size_t i = valid ? i_raw : p.nrays - 1;    // lanes past the end redo the last ray and store nothing
    bool take = valid;                               // lanes that store a start state
    if (redo) {                                      // second pass of the fast job: the marked rays only, no ordering
        take = valid && (ok[i] == START_AGAIN);
        if (!wave_any(take)) return;
    }
    const int ix = (int)(i % (size_t)p.nx);
    const int iy = p.y0 + (int)(i / (size_t)p.nx);
    const double alpha = (((double)(ix) + .5) / (double)(p.nx) - 0.5) * 2.0 * p.rmax;
    const double beta = (((double)(iy) + .5) / (double)(p.ny) - 0.5) * 2.0 * p.rmax *
                        ((double)p.ny / (double)p.nx);
    Geod gd;
    GeodCache cache;
    int err = 0;
    int good = 0;
    double x[4] = { 0.0, p.r0, 0.0, 0.0 }, k[4] = { 0.0, 0.0, 0.0, 0.0 };
    RayState s;
    s.dk[0] = s.dk[1] = s.dk[2] = s.dk[3] = 0.0; s.kt = 0.0; s.Q = 0.0;
    if (redo && !take) {
        // (a lane of the second pass whose ray is not marked: nothing to form)
    } else
    if (p.options & 1) {
        // RTOPT_FLAT: there is no geodesic_init_inf for Minkowski space; the ray is the straight line
        // that reaches the observer's image plane at (alpha, beta), started on the sphere r = r0.
        // Observer frame: line of sight n = (sin i, 0, cos i), e_alpha = (0,1,0), e_beta = (-cos i, 0, sin i).
        const double D2 = p.r0 * p.r0 - alpha * alpha - beta * beta;
        if (D2 > 0.0) {
            const double D = sqrt(D2);
            const double X = -beta * p.cos_i + D * p.sin_i, Y = alpha, Z = beta * p.sin_i + D * p.cos_i;
            const double Vx = -p.sin_i, Vy = 0.0, Vz = -p.cos_i;
            const double r = sqrt(X * X + Y * Y + Z * Z);
            const double m = Z / r;
            const double kr = (X * Vx + Y * Vy + Z * Vz) / r;
            x[1] = r; x[2] = m; x[3] = atan2(Y, X);
            k[0] = 1.0;
            k[1] = kr;
            k[2] = -(Vz - m * kr) / (r * sqrt(1. - m * m));
            k[3] = (X * Vy - Y * Vx) / (X * X + Y * Y);
            raytrace_prepare(p.a, x, k, p.precision, p.options, s);
            good = 1;
        }
    } else if (init_inf(p.incl, p.sin_i, p.cos_i, p.a, alpha, beta, gd, err, cache) && p.r0 > gd.rp) {
        const double P0 = P_int(gd, p.r0, 0);
        x[2] = position_pol(gd, P0);
        momentum(gd, P0, p.r0, x[2], k);
        if (!isnan(k[0]) && !isnan(x[2])) {
            raytrace_prepare(p.a, x, k, p.precision, p.options, s);
            good = 1;
        }
    }
#if S5_FAST
    // R and M of the start as the momentum holds them (k^r S, k^theta S) against the size of their terms
    if (good && !(p.options & 1) && !redo) {
        const double S = p.r0 * p.r0 + (gd.a * gd.a) * (x[2] * x[2]);
        const double rootR = k[1] * S, rootM = k[2] * S, big = p.r0 * p.r0 + gd.a * gd.a;
        if ((rootM * rootM < 1e-6 * (fabs(gd.q) + gd.l * gd.l + gd.a * gd.a)) || (rootR * rootR < 1e-6 * (big * big))) good = START_AGAIN;
    }
#endif
    const size_t n = st.cap;
    if (take) {
    cols[COL_X0 * n + i] = x[0]; cols[COL_X1 * n + i] = x[1]; cols[COL_X2 * n + i] = x[2]; cols[COL_X3 * n + i] = x[3];
    cols[COL_K0 * n + i] = k[0]; cols[COL_K1 * n + i] = k[1]; cols[COL_K2 * n + i] = k[2]; cols[COL_K3 * n + i] = k[3];
    cols[COL_DK0 * n + i] = s.dk[0]; cols[COL_DK1 * n + i] = s.dk[1];
    cols[COL_DK2 * n + i] = s.dk[2]; cols[COL_DK3 * n + i] = s.dk[3];
    cols[COL_KT * n + i] = s.kt; cols[COL_Q * n + i] = s.Q;
    ok[i] = good;
    }
    if (redo) return;

    // ORDER OF THE MARCH (scheduling only: a ray's result does not depend on it).  The march kernel hands rays out along
    // `order`; a ray that needs 2 000 raytrace() calls handed out last keeps one lane of one wave busy for ~10 ms after
    // everything else is done (measured: T = 9.9 ms + steps / 2.7e10 per s over jobs of 512^2 .. 2048^2 rays in row-major
    // order).  The long rays are known by their constants of motion: those that wind around the photon orbit -- the radial
    // turning point nearly a double root, (r1 - r2)/r1 < 0.2, or the complex pair of a plunging ray nearly real, |Im r3| <
    // 0.2 |Re r3| -- and those that pass the polar axis (1 - m2p < 0.01: the step size follows sin theta).  On the C4 job
    // that rule marks 7.7 % of the rays, among them every ray above 1 000 calls and 89 % of those above 800 (median 506;
    // tests/tools/torus_long_predict.py).  They are dealt into the head of the order, every second position (torus_order_kernel), the
    // other rays fill the gaps and follow in the order of their ranks (which come from atomics: the order within a class
    // differs from run to run, a ray's result does not); a ray wrongly taken for long costs nothing.
    // (A third class -- the SHORTEST rays last: those that fall into the hole, 250 - 450 calls -- is built and measured,
    // -DS5_SHORT_CLASS: 25.85 against 25.55 ms without it; not used.)
#define S5_FAR_CLASS 8
    int cls = 0;                                                         // 0 ordinary, 1 long, 2 last (far rays; -DS5_SHORT_CLASS: the short ones too)
    if (good && !(p.options & 1)) {
        const double crit = (gd.nrr == 4) ? (gd.r1[0] - gd.r2[0]) / gd.r1[0]
                          : (gd.nrr == 2) ? fabs(gd.r3[1]) / fmax(fabs(gd.r3[0]), 1e-9) : 9.0;
#define S5_LONG_CRIT 0.2
#define S5_LONG_POLE 0.01
        if ((crit < S5_LONG_CRIT) || (1.0 - gd.m2p < S5_LONG_POLE)) cls = 1;
        // THE RAYS HANDED OUT LAST decide when a CU ends (its pool drains alone: the CUs of the C4 job ended between 90 % and
        // 100 % of the run, half of them before 95 %).  The rays that pass far from the hole -- impact parameter above
        // S5_FAR_CLASS gravitational radii -- are the most uniform in length (472 - 520 calls beyond b = 10 on the C4 job,
        // where the others spread from 220 to 1 900), so they close the order: C4 21.4 -> 20.65 ms in one call (thresholds
        // 6 / 8 / 10 / 12: 20.9 / 20.65 / 20.85 / 20.95; a fourth class for b > 12 behind them: 20.55, not kept).
        else if (alpha * alpha + beta * beta > (double)(S5_FAR_CLASS) * (double)(S5_FAR_CLASS)) cls = 2;
    }
    // rank of the ray within its class: counted per wave (ballot), summed per workgroup in LDS, ONE atomic per class and
    // workgroup on the global counters (an atomic per wave and class: 49 k atomics on three addresses, 0.18 ms -- measured);
    // torus_order_kernel turns (class, rank) into the position once the counts are final
    __shared__ unsigned s_cnt[3][4];                 // [class][wave]
    __shared__ unsigned long long s_base[3];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long below = (lane == 0u) ? 0ull : (~0ull >> (64u - lane));
    unsigned long long m[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        m[c] = __builtin_amdgcn_ballot_w64(valid && cls == c);
        if (lane == 0u) s_cnt[c][wave] = (unsigned)__builtin_popcountll(m[c]);
    }
    __syncthreads();
    if (threadIdx.x < 3u) {
        const unsigned c = threadIdx.x;
        const unsigned tot = s_cnt[c][0] + s_cnt[c][1] + s_cnt[c][2] + s_cnt[c][3];
        s_base[c] = tot ? atomicAdd(&counters[c], (unsigned long long)tot) : 0ull;
    }
    __syncthreads();
    if (!valid) return;
    unsigned long long rk = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        unsigned before = 0;
        for (unsigned w = 0; w < wave; ++w) before += s_cnt[c][w];
        if (cls == c) rk = s_base[c] + before + (unsigned long long)__builtin_popcountll(m[c] & below);
    }
    order[i] = (int)(((unsigned)cls << 30) | (unsigned)rk);
}

// The order of the march from the ranks.  The long rays are DEALT into the head of the order, every M-th position (M = 2
// when they are under half of the job), the ordinary ones fill the gaps in their own order and follow, the last class
// (the far rays, torus_start_kernel) closes.  Dealt rather than put in front: a pool needs rays of both kinds -- Verlet attempts and RK4 fallbacks -- to fill its
// batches, and the long rays are the ones that fall back at almost every step.  Measured on MI355X, one call, 1024^2 / 2048^2
// rays: row-major 29.1 / 90.6 ms; all long rays first 26.5 / 95.4 (-8 % asymptotic rate); dealt 1:2 25.9 / 89.8; 1:3 29.2 /
// 92.7; 1:4 27.4 / 89.8.
#define S5_ORDER_DEAL 2
__global__ __launch_bounds__(256)
void torus_order_kernel(size_t n, const int* __restrict__ ranks, const unsigned long long* __restrict__ counters, int* __restrict__ order)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long n_long = counters[1], n_short = counters[2];
    const unsigned long long head = (unsigned long long)n - n_short;             // ordinary + long
    unsigned long long M = (n_long == 0) ? 1ull : head / n_long;
    M = M < 1ull ? 1ull : (M > (unsigned long long)S5_ORDER_DEAL ? (unsigned long long)S5_ORDER_DEAL : M);
    const unsigned v = (unsigned)ranks[i];
    const unsigned cls = v >> 30;
    const unsigned long long j = (unsigned long long)(v & 0x3fffffffu);
    unsigned long long pos;
    if (cls == 1u) pos = M * j;
    else if (cls == 2u) pos = head + j;
    else if (M > 1ull && j < (M - 1ull) * n_long) pos = j + j / (M - 1ull) + 1ull;
    else pos = j + n_long;
    order[pos] = (int)i;
}

template <class PRM>
S5_DEV double torus_density(const PRM& p, double r, double m)
{
    if (p.shape == 1) return (r <= p.torus_w) ? 1.0 : 0.0;
    const double R = r * sqrt(1. - m * m), z = r * m;
    const double d2 = sq(R - p.torus_r) + z * z;
    const double w2 = 2. * p.torus_w * p.torus_w;
    return (d2 < 36. * w2) ? exp(-d2 / w2) : 0.0;       // cut at 6 sqrt(2) w: exp(-36) ~ 2e-16
}

#define S5_MARCH_WAVES (S5_FAST ? 3 : 1)   // fast: THREE waves per SIMD since round 4 (168 VGPRs, nothing spilled: the lean form of the
// kernel below); strict: its bodies need ~261 registers -- one wave per SIMD and no scratch
// Emission and absorption picked up over one accepted step (see the header comment for the model).  `g` is the
// metric at the end point of the step, which both halves of raytrace() have just evaluated there.
// what accumulate_transfer reads of the job (the members of TorusParams it uses, same names): the march kernel's lean form
// fetches them together at the head of its store / transfer / end-test phase
struct TorusModel { int shape; double torus_w, torus_r, cut_d2, inv_2w2, torus_l, emis0, absorb0; };
// a value loaded through the constant address space is in its scalar register(s) from HERE on (the optimiser would otherwise
// sink the load to its use, behind whatever branch that sits)
S5_DEV void keep_here(const double& v) { asm volatile("" :: "s"(v)); }
S5_DEV void keep_here(const int& v) { asm volatile("" :: "s"(v)); }

template <class PRM>
S5_DEV void accumulate_transfer(const PRM& p, const bool no_absorption, const RayState& s, const Metric& g, const double x[4],
                                const double k[4], double dl_taken, double& I, double& tau)
{
#if S5_FAST
    // same quantities with the divisions folded: exp(-d2 / 2w^2) with the wave-uniform 1/(2w^2), u^t from one rsqrt,
    // g^4 dl/g = g^3 dl; the step through the absorber only when there is one (absorb0 is wave-uniform)
    double rho;
    if (p.shape == 1) rho = (x[1] <= p.torus_w) ? 1.0 : 0.0;
    else {
        const double R = x[1] * msqrt(1. - x[2] * x[2]), z = x[1] * x[2];
        const double d2 = sq(R - p.torus_r) + z * z;
        rho = (d2 < p.cut_d2) ? mexp(-d2 * p.inv_2w2) : 0.0;
    }
    if (!(rho > 0.0)) return;
    // Omega = A/B, u^t = |B| / sqrt(Q) with Q = -(g00 B^2 + 2 A B g03 + A^2 g33): the orbit is time-like iff Q > 0, and
    // g = E / (u^t (k_t + Omega k_phi)) = E sign(B) sqrt(Q) / (k_t B + A k_phi) -- one square root and one division
    const double A = -(g.g03 + p.torus_l * g.g00), B = g.g33 + p.torus_l * g.g03;
    const double Q = -(g.g00 * (B * B) + 2. * (A * B) * g.g03 + (A * A) * g.g33);
    if (!(Q > 0.0)) return;
    const double k_t = k[0] * g.g00 + k[3] * g.g03;
    const double k_f = k[3] * g.g33 + k[0] * g.g03;
    const double sQ = sqrt_pos(Q);
    const double gfac = mdiv((B >= 0.0 ? s.E : -s.E) * sQ, k_t * B + A * k_f);
    // (the two sums through local increments: with `I += ..` in one branch and `I += ..; tau += ..` in the other the compiler
    // kept I and tau in a 16-byte stack slot indexed by the branch -- the kernel's only scratch use, inside the step loop)
    double add_I, add_tau = 0.0;
    if (no_absorption) {
        add_I = (gfac * gfac * gfac) * (p.emis0 * rho) * dl_taken;
    } else {
        const double ds = mdiv(dl_taken, gfac);
        const double g2 = gfac * gfac;
        add_I = (g2 * g2) * p.emis0 * rho * mexp(-tau) * ds;
        add_tau = p.absorb0 * rho * ds;
    }
    I += add_I;
    if (!no_absorption) tau += add_tau;
#else
    const double rho = torus_density(p, x[1], x[2]);
    if (!(rho > 0.0)) return;
    const double Om = omega_from_ell(p.torus_l, g);
    const double nrm = -(g.g00 + 2. * Om * g.g03 + Om * Om * g.g33);
    if (!(nrm > 0.0)) return;                           // no time-like circular orbit with this ell here
    const double ut = mdiv(1., msqrt(nrm));
    const double k_t = k[0] * g.g00 + k[3] * g.g03;
    const double k_f = k[3] * g.g33 + k[0] * g.g03;
    const double gfac = mdiv(s.E, ut * (k_t + Om * k_f));   // E_inf / E_local
    const double ds = mdiv(dl_taken, gfac);
    const double g2 = gfac * gfac;
    const double att = no_absorption ? 1.0 : exp(-tau);          // tau stays 0 without absorption (wave-uniform test)
    I += (g2 * g2) * p.emis0 * rho * att * ds;
    tau += p.absorb0 * rho * ds;
#endif
}

template <class AUX>
S5_DEV void write_ray_end(const AUX& aux, sim5gpu_stokes* __restrict__ out, size_t ray,
                          const double x[4], const double k[4], const RayState& s, double I, double tau, float worst)
{
    sim5gpu_stokes rec = { I, 0.0, 0.0, 0.0, tau };
    out[ray] = rec;
    if (aux.steps) aux.steps[ray] = s.pass;
    if (aux.max_step_error) aux.max_step_error[ray] = worst;
    if (aux.carter_error) aux.carter_error[ray] = raytrace_error(x, k, s);
    if (aux.x_end) { double* __restrict__ o = aux.x_end + 4 * ray; o[0] = x[0]; o[1] = x[1]; o[2] = x[2]; o[3] = x[3]; }
#ifdef S5_TORUS_DEBUG
    if (aux.x_end) aux.x_end[4 * ray] = (double)wall_clock64();      // debug builds: WHEN the ray ended, in place of its t
#endif
#ifndef S5_TORUS_DEBUG
    if (aux.k_end) { double* __restrict__ o = aux.k_end + 4 * ray; o[0] = k[0]; o[1] = k[1]; o[2] = k[2]; o[3] = k[3]; }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// torus_pool_kernel: persistent workgroups of four waves, each workgroup with a pool of rays in LDS.
//
// What occupancy counters showed when 64 image neighbours march in lock-step (C4, 1024^2): at precision 1,
// 76 % of the wave-steps contain a lane whose Verlet attempt fails, 27 of 64 lanes on average; at precision 0.01, 29 % of
// the wave-steps, 2.9 lanes on average.  The RK4 body (5 connection evaluations) costs ~3 Verlet attempts, so
// a wave spends most of its time in RK4 with a fraction of its lanes -- whatever the order of the rays.
// Here the two halves of raytrace() are separate batches over the wave's pool:
//   V batch: up to 64 pool rays whose next action is a Verlet attempt.  Accepted -> the step is finished
//            (transfer, end test) and the ray stays V.  Rejected -> nothing was modified (the attempt
//            restores x, k; ref :221-222), the ray is tagged R.
//   R batch: up to 64 rays tagged R: recompute the step size (same expression, same operands as the failed
//            attempt), RK4 step, finish the step, back to V.
// With 128 rays to choose from one of the two kinds always has >= 64 rays while the pool is full, so both bodies
// run with (nearly) all lanes.  A ray's arithmetic is unchanged: same calls, same operands, same order.
// Finished rays are replaced from the global cursor (one atomic per wave and refill pass).  The waves of a workgroup
// share the pool through the slots' tags only (LDS compare-and-swap; no locks, no barriers after the first).
// ---------------------------------------------------------------------------------------------------------
// ONE POOL PER WORKGROUP, THREE QUEUES (round 4).  Until round 3 every wave had a private pool of 128 rays (512 per
// workgroup), scanned its slots' tags for work, and the four pools were shared only in the drain phase.  What a wave needs
// is 64 rays of one kind when it comes back from a batch; while the other three waves each hold at most 64 rays in their
// batches, a pool of 4 x 64 + 64 = 320 rays leaves the returning wave the same 128 to choose from (its own 64 + 64 idle
// ones) as the private pools did -- with 320 rays in flight per workgroup instead of 512.  Rays in flight are what the job
// pays for at its end: when the cursor runs out every pooled ray still has half its steps to make, with batches that thin
// out (T = 7 ms + steps / 2.65e10 at 512 per workgroup).  Sharing by SCANNING the tags of the whole pool costs what the
// smaller pool gains (measured: 28.4 ms at 512 shared slots against 25.9 with private pools, 25.7 at 320), so the slots now
// travel through three ring queues in LDS -- empty slots (E), rays whose next action is a Verlet attempt (V), rays that owe
// the RK4 half of a call (R): a wave takes up to 64 entries of a queue with ONE compare-and-swap on its head and returns
// them with one atomic add on a tail per kind; nothing is scanned, the order is first in first out (no ray waits while
// others are stepped twice), and a wave with nothing to take sleeps until the workgroup's last ray is done instead of
// retiring early.
// Workgroup width.  Fast variant: ALL waves of a CU are one workgroup with one pool of (waves x 64 + 64) rays -- fewer rays in
// flight than with several pools per CU, and the drain phase shares the rays of the whole CU.  Measured, C4, one call each, at
// two waves per SIMD: 4 waves x 256 slots 25.9 ms, x 320 24.1-24.4, x 384 24.2; 8 waves x 576 23.6, x 640 23.6-23.8 (private
// 128-ray pools of round 3: 25.7); at three waves per SIMD (the lean form below): 4-wave workgroups 22.4-22.7, twelve waves x 832
// slots 21.2-21.6, x 896 21.4.  Strict variant: its bodies need one wave per SIMD; four waves per workgroup keep every CU busy.
#define S5_POOL_WG_WAVES (S5_FAST ? 4 * S5_MARCH_WAVES : 4)      // fast: the twelve waves a CU holds are ONE workgroup
#define S5_POOL_WG_SLOTS (64 * S5_POOL_WG_WAVES + 64)     // rays per workgroup pool: every wave's batch + 64
#define S5_POOL_REFILL_MIN 32            // empty slots that make a refill worth its global atomic and loads
constexpr int WG_WAVES = S5_POOL_WG_WAVES;
constexpr int WG_THREADS = 64 * WG_WAVES;
constexpr int WG_SLOTS = S5_POOL_WG_SLOTS;
constexpr int RING = (WG_SLOTS <= 512) ? 512 : 1024;         // entries of a queue: a power of two >= WG_SLOTS (a slot is in one queue at most)
static_assert(WG_SLOTS % 64 == 0 && WG_SLOTS >= 128 && WG_SLOTS <= RING, "pool size");
#define POOL_KEEP_NUM 7                   // ... while at least NUM/DEN of its lanes are still stepping
#define POOL_KEEP_DEN 8
#define POOL_RUN 6                       // Verlet attempts a batch may take before it returns to the pool (measured with the queues
// of round 4, C4, one call: 4 / 6 / 8 -> 23.8 / 23.4 / 23.4 ms at two waves, 21.3-21.4 at three)
enum : int { PC_X0 = 0, PC_X1, PC_X2, PC_X3, PC_K0, PC_K1, PC_K2, PC_K3, PC_DK0, PC_DK1, PC_DK2, PC_DK3,
             PC_KT, PC_E, PC_I, PC_TAU, NPC };
enum : int { TAG_EMPTY = 0, TAG_V = 1, TAG_R = 2 };            // what a ray owes next = the queue its slot goes back to
enum : int { Q_E = 0, Q_V = 1, Q_R = 2, NQ = 3 };
// control words of a workgroup (LDS): head and tail of the three queues, rays alive in the pool, "the cursor is exhausted"
enum : int { CW_HEAD = 0, CW_TAIL = NQ, CW_LIVE = 2 * NQ, CW_DRAINED = 2 * NQ + 1, NCW = 8 };
constexpr unsigned short RING_VOID = 0xffffu;                 // a queue entry that has been reserved but not written yet
// A ray's sixteen doubles lie together, slots PD_STRIDE doubles apart (one double of padding: a stride of 16 doubles would put
// every lane of a batch on the same LDS banks): one address register per ray and small immediate offsets for its columns,
// whatever the size of the pool (column-major, the offsets of the last columns of an 832-slot pool no longer fit the
// instruction's 16-bit offset field and cost address registers the three-wave build does not have).
constexpr int PD_STRIDE = NPC + 1;
#define PD_AT(c, slot) ((slot) * PD_STRIDE + (c))
constexpr int POOL_WG_BYTES = ((PD_STRIDE * WG_SLOTS * 8 + 3 * WG_SLOTS * 4 + NQ * RING * 2 + NCW * 4 + 4 * 8) + 15) / 16 * 16;

S5_DEV void wg_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
S5_DEV void wg_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

S5_DEV void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// All arguments of the march kernel in ONE block.  LEAN (fast variant): the kernel reads it through the constant address space
// where a value is used (param_reload at the head of every phase: nothing of it is held -- or spilled into vector lanes --
// across the step bodies); otherwise (strict) it is an ordinary by-value argument block.
struct PoolArgs {
    TorusParams p;
    RayCols start;
    const int* ok;
    const int* order;
    unsigned long long* cursor;
    sim5gpu_stokes* out;
    TorusAux aux;
};
#define S5_MARCH_LEAN (S5_FAST ? 1 : 0)

#ifdef S5_TORUS_DEBUG
// Instrumented build only (tests/tools/torus_phases.py): cycles between the phase marks of s5_raytrace.hpp, summed per
// workgroup in LDS.  A mark reads the cycle counter FIRST (s_memtime between scheduling barriers, after everything the
// phase issued has completed), then does its book-keeping -- the wave's previous time stamp is a word of LDS (a register
// would be a per-lane copy: the marks sit in divergent code and the lanes that pass them change), the sums are LDS atomics of
// the first active lane -- and stamps the END of the book-keeping as the start of the next phase: the marks themselves are
// outside every interval (PH_N: the interval between two marks with nothing in between, as a check: a few cycles).
struct PhaseClock {
    unsigned long long* acc;      // LDS [PH_N + 1] cycles
    unsigned* cnt;                // LDS [PH_N + 1] marks passed
    unsigned long long* tw;       // LDS: this wave's last time stamp
    bool sampled;                 // this wave keeps the book (a big job: one wave of one workgroup, so that the marks of 3 071 other
                                  // waves do not load the LDS the measured wave shares with eleven of them)
    S5_DEV void mark(int i) const
    {
        if (!sampled) return;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): LDS and scalar loads of the phase (not its global stores)
        const unsigned long long now = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long old = *tw;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(true);
        if (__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)) == 0u) {
            atomicAdd(&acc[i], now - old);
            atomicAdd(&cnt[i], 1u);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_sched_barrier(0);
        *tw = __builtin_readcyclecounter();
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_sched_barrier(0);
    }
};
constexpr int PHASE_LDS_BYTES = (PH_N + 1) * 8 + (PH_N + 1) * 4 + 4 + 16 * 8 + WG_SLOTS * 8;   // sums, counts, pad, one stamp per wave, one per slot
constexpr size_t PHASE_DBG_AT = 12000;        // index into the debug buffer (aux.k_end as 64-bit words): PH_N sums, PH_N counts
#else
constexpr int PHASE_LDS_BYTES = 0;
#endif

__global__ __launch_bounds__(WG_THREADS, S5_MARCH_WAVES)
void torus_pool_kernel(PoolArgs args)
{
    extern __shared__ char pool_raw[];
    const int lane = (int)(threadIdx.x & 63);
    double* pd = (double*)pool_raw;                                  // [WG_SLOTS][PD_STRIDE]
    int* pray = (int*)(pool_raw + PD_STRIDE * WG_SLOTS * 8);         // [WG_SLOTS]
    int* ppass = pray + WG_SLOTS;
    float* pworst = (float*)(ppass + WG_SLOTS);
    unsigned short* ring = (unsigned short*)(pworst + WG_SLOTS);     // [NQ][RING]
    unsigned* cw = (unsigned*)(ring + NQ * RING);                    // [NCW]
    double* cwd = (double*)(cw + NCW);                               // [4]: step_epsilon, r_in, r_out (formed once, below)
#ifdef S5_TORUS_DEBUG
    unsigned long long* ph_acc = (unsigned long long*)(pool_raw + POOL_WG_BYTES);
    unsigned* ph_cnt = (unsigned*)(ph_acc + PH_N + 1);
    unsigned long long* ph_tw = (unsigned long long*)(ph_cnt + PH_N + 2);
    if (threadIdx.x <= PH_N) { ph_acc[threadIdx.x] = 0ull; ph_cnt[threadIdx.x] = 0u; }
    const unsigned long long ph_t0 = __builtin_readcyclecounter();
    ph_tw[threadIdx.x >> 6] = ph_t0;
    unsigned long long* ph_slot = ph_tw + 16;                        // [WG_SLOTS]: when the ray of a slot was last given to a queue
    for (int i = (int)threadIdx.x; i < WG_SLOTS; i += WG_THREADS) ph_slot[i] = 0ull;
    const bool ph_sampled = (args.p.nrays <= 64) || (blockIdx.x == 0 && (threadIdx.x >> 6) == 0);
    const PhaseClock ph = { ph_acc, ph_cnt, ph_tw + (threadIdx.x >> 6), ph_sampled };
#else
    const NoPhases ph;
#endif

#if S5_MARCH_LEAN
    // LEAN: between two attempts a ray lives in its LDS slot, not in registers -- an attempt loads what it needs when it needs
    // it (position and momentum at its head, kt at the precision check, the transfer integrals and the ray's identity after an
    // accepted step) and stores the new state when the step is accepted; a rejected attempt stores nothing.  With the job's
    // parameters read where they are used, the registers a wave carries THROUGH the two step bodies are the slot number and a
    // few flags: the bodies themselves are what is left (round 3: 230 VGPRs, ~66 of them state and uniform parameters).
    const S5_AS4 PoolArgs& A = *(const S5_AS4 PoolArgs*)__builtin_amdgcn_kernarg_segment_ptr();
#else
    const PoolArgs& A = args;
#endif
    const size_t n = A.p.nrays;
    const bool no_absorption = (A.p.absorb0 == 0.0);               // the wave-uniform branch on it stays scalar
    if (threadIdx.x == 0) {
        cwd[0] = S5_DIVC(msqrt(A.p.precision), 10.);
        cwd[1] = A.p.r_stop_in * r_horizon(A.p.a);
        cwd[2] = A.p.r_stop_out * A.p.r0;
    }
    // the pool: every slot in the queue of empty slots, the other two queues void
    for (int i = (int)threadIdx.x; i < NQ * RING; i += WG_THREADS) ring[i] = (i < WG_SLOTS) ? (unsigned short)i : RING_VOID;
    if (threadIdx.x < NCW) cw[threadIdx.x] = (threadIdx.x == CW_TAIL + Q_E) ? (unsigned)WG_SLOTS : 0u;
    __syncthreads();                                                 // the only workgroup barrier: the pool exists
#ifdef S5_TORUS_DEBUG
    // timeline of the wave (100 MHz clock): start, first time the cursor was found exhausted, exit
    const int wave = (int)(threadIdx.x >> 6);
    unsigned long long* tl = (unsigned long long*)A.aux.k_end + 16 + 3 * ((size_t)blockIdx.x * WG_WAVES + wave);
    if (lane == 0) { tl[0] = wall_clock64(); tl[1] = 0; tl[2] = 0; }
#endif

    // Every wait in this kernel is bounded: the spins of the queue primitives by SPIN_CAP (~50 ms; they normally last a few
    // cycles), the main loop by `guard` passes AND by a wall-clock budget far above any job's need (5 s + 1 ns per ray and
    // permitted step): a wave that runs into either leaves the loop, so the grid always drains -- the rays it abandons are
    // never written, which the step counts and hit checks of every caller and test catch.
    constexpr int SPIN_CAP = 1 << 20;
    bool broken = false;
    const unsigned long long t_begin = wall_clock64();
    const unsigned long long t_budget = 100000000ull * 5ull + (unsigned long long)n * (unsigned long long)(A.p.max_steps > 0 ? A.p.max_steps : 1) / 10ull;
    // Queue primitives (wave-uniform calls).  take: up to `want` entries from queue q -- one lane snapshots head and tail and
    // moves the head by compare-and-swap (retried while other waves move it); lane i < count then owns entry head + i: it waits
    // for the entry to be written (a producer reserves its positions before it fills them), reads the slot and voids the entry.
    // give: the lanes with `mine` append their slot to queue q -- one atomic add on the tail, then each lane writes its entry.
    auto q_take = [&](const int q, const int want, int& slot) -> int {
        unsigned h = 0, cnt = 0;
        if (lane == 0) {
            for (;;) {
                h = __hip_atomic_load(&cw[CW_HEAD + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned t = __hip_atomic_load(&cw[CW_TAIL + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned avail = t - h;
                cnt = avail < (unsigned)want ? avail : (unsigned)want;
                if ((int)avail <= 0) { cnt = 0; break; }
                if (atomicCAS(&cw[CW_HEAD + q], h, h + cnt) == h) break;
            }
        }
        h = (unsigned)__builtin_amdgcn_readfirstlane((int)h);
        cnt = (unsigned)__builtin_amdgcn_readfirstlane((int)cnt);
        slot = 0;
        if ((unsigned)lane < cnt) {
            unsigned short* e = &ring[q * RING + ((h + (unsigned)lane) & (unsigned)(RING - 1))];
            unsigned v;
            int spins = 0;
            while ((v = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == RING_VOID && ++spins < SPIN_CAP) __builtin_amdgcn_s_sleep(1);
            if (v == RING_VOID) { broken = true; v = 0; }             // (never seen: the writer is a few instructions behind its reservation)
            __hip_atomic_store(e, RING_VOID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            slot = (int)v;
        }
        wg_acquire();                                               // the state of the taken slots, written by the wave that gave them
        return (int)cnt;
    };
    auto q_give = [&](const int q, const bool mine, const int slot) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
        if (!m) return;
        const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        unsigned base = 0;
        if (mine && rank == 0u) base = atomicAdd(&cw[CW_TAIL + q], (unsigned)__builtin_popcountll(m));
        base = (unsigned)__shfl((int)base, __builtin_ctzll(m), 64);
        if (mine) {
            // the entry is void unless a taker that reserved it a whole ring ago has not read it yet (it would have to sleep
            // through seven batches of the other waves): wait for that read rather than overwrite it
            unsigned short* e = &ring[q * RING + ((base + rank) & (unsigned)(RING - 1))];
            int spins = 0;
            while (__hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != RING_VOID && ++spins < SPIN_CAP) __builtin_amdgcn_s_sleep(1);
            if (spins >= SPIN_CAP) broken = true;
            __hip_atomic_store(e, (unsigned short)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };

    // every pass consumes cursor positions, advances at least one pooled ray by half a raytrace() call, or waits for the other
    // waves of the workgroup (bounded like the rest: a waiting pass is a sleep of ~1 us, the bound allows for it)
    const unsigned long long guard = 64ull * (unsigned long long)(A.p.max_steps + 2) * ((n + 63) / 64 + 2);
    for (unsigned long long it = 0; it < guard; ++it) {
        if (__builtin_amdgcn_ballot_w64(broken)) break;
        if ((it & 255ull) == 255ull && wall_clock64() - t_begin > t_budget) break;
        // ---- 1. new rays into empty slots, as long as the cursor has any: when a batch's worth of slots is empty, or the
        //         queues could not fill a batch otherwise
        const bool drained = __hip_atomic_load(&cw[CW_DRAINED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u;
        unsigned nE = 0, nV = 0, nR = 0;
        {
            unsigned hv = 0;
            if (lane < 2 * NQ) hv = __hip_atomic_load(&cw[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned h = (unsigned)__shfl((int)hv, lane % NQ, 64), t = (unsigned)__shfl((int)hv, NQ + lane % NQ, 64);
            const int d = (int)(t - h);
            const unsigned dd = d > 0 ? (unsigned)d : 0u;
            nE = (unsigned)__shfl((int)dd, Q_E, 64); nV = (unsigned)__shfl((int)dd, Q_V, 64); nR = (unsigned)__shfl((int)dd, Q_R, 64);
        }
        if (!drained && nE > 0u && (nE >= (unsigned)S5_POOL_REFILL_MIN || nV + nR < 64u)) {
            int slot;
            const int got = q_take(Q_E, 64, slot);
            if (got > 0) {
                const bool have = lane < got;
                unsigned long long base = 0;
                const auto& R = param_reload(A);                       // (pointers of the refill: loaded here)
                const double* __restrict__ sc = R.start.d;
                const size_t scap = R.start.cap;
                if (lane == 0) {
                    atomicAdd(&cw[CW_LIVE], (unsigned)got);                 // counted alive BEFORE the cursor is asked (see the exit test)
                    base = atomicAdd(R.cursor, (unsigned long long)got);
                }
                base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                       (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);
                if (base + (unsigned long long)got >= n) {
#ifdef S5_TORUS_DEBUG
                    if (lane == 0 && tl[1] == 0) tl[1] = wall_clock64();
#endif
                    if (lane == 0) __hip_atomic_store(&cw[CW_DRAINED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                const unsigned long long mine = base + (unsigned long long)lane;
                bool started = false;
                if (have && mine < n) {
                    const size_t ray = (size_t)R.order[mine];       // (torus_start_kernel: the long rays first)
                    if (!R.ok[ray]) {
                        // rejected at start-up: an empty record, the slot goes back to the empty ones
                        sim5gpu_stokes z = { 0.0, 0.0, 0.0, 0.0, 0.0 };
                        R.out[ray] = z;
                        if (R.aux.steps) R.aux.steps[ray] = 0;
                        if (R.aux.max_step_error) R.aux.max_step_error[ray] = 0.0f;
                        if (R.aux.carter_error) {
                            // NaN made HERE (an opaque zero added to its high word): as a literal the compiler hoists it out of the
                            // main loop and carries it, two registers, through both step bodies -- at three waves per SIMD the one
                            // value it then spilled
                            int z = 0;
                            asm volatile("" : "+v"(z));
                            R.aux.carter_error[ray] = __hiloint2double(0x7ff80000 + z, z);
                        }
                        if (R.aux.x_end) { for (int c = 0; c < 4; ++c) R.aux.x_end[4 * ray + c] = sc[(COL_X0 + c) * scap + ray]; }
#ifndef S5_TORUS_DEBUG
                        if (R.aux.k_end) { for (int c = 0; c < 4; ++c) R.aux.k_end[4 * ray + c] = sc[(COL_K0 + c) * scap + ray]; }
#endif
                    } else {
#pragma unroll
                        for (int c = 0; c < 13; ++c) pd[PD_AT(c, slot)] = sc[c * scap + ray];   // x, k, dk, kt
                        pd[PD_AT(PC_E, slot)] = sc[COL_KT * scap + ray];
                        pd[PD_AT(PC_I, slot)] = 0.0;
                        pd[PD_AT(PC_TAU, slot)] = 0.0;
                        pray[slot] = (int)ray; ppass[slot] = 0; pworst[slot] = 0.0f;
                        started = true;
                    }
                }
                wg_release();                                          // the state before the queue entry
                const unsigned long long sm = __builtin_amdgcn_ballot_w64(started);
                const int unused = got - __builtin_popcountll(sm);
                if (unused > 0 && lane == 0) atomicSub(&cw[CW_LIVE], (unsigned)unused);
                q_give(Q_V, started, slot);
                q_give(Q_E, have && !started, slot);                   // rejected rays and positions past the end
                continue;                                              // look at the queues again
            }
        }

        // ---- 2. what is there to do
        if (nV + nR == 0u) {
            // nothing to take: the rays of the workgroup are all in other waves' batches (or finished).  Leave when the cursor
            // is exhausted and no ray of the pool is alive; wait otherwise.
            if (drained && __hip_atomic_load(&cw[CW_LIVE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) break;
            __builtin_amdgcn_s_sleep(32);
            continue;
        }
        const bool do_rk4 = (nR >= 64u) || (nV == 0u) || (nV < 64u && nR > nV);
        // (Round 6 built and measured a THIN mode for the end of the job -- once the cursor is exhausted and the pool holds at most
        // eight rays per wave, a batch keeps its rays, runs the RK4 half of a rejected call in the same wave at the next pass and
        // stays away from the queues for 48 calls -- and dropped it: C4 20.8 -> 21.5 ms at 96 rays, 21.0 at 24 (one call, twice:
        // profiles/r06_torus_ab_thin.txt).  A wave that serves both halves runs them one after the other for all its lanes; two
        // waves on two SIMDs run them side by side.  A shorter sleep of the idle waves after the cursor's end: +-0.)

        // ---- 3. take up to 64 rays of the chosen kind
        int slot;
        const int take = q_take(do_rk4 ? Q_R : Q_V, 64, slot);
        if (take == 0) continue;                     // another wave was faster: look again
        const bool active = lane < take;
#ifdef S5_TORUS_DEBUG
        if (active && ph_sampled) {
            // PH_QUEUES = the transit of a RAY through the queues: from the stamp its slot got when it was given back to this take
            // (the first active lane's ray stands for the batch); the wave's own stamp starts here -- its idle time is no phase
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned long long now = __builtin_readcyclecounter();
            const unsigned long long given = ph_slot[slot];
            if (lane == 0 && given != 0ull) { atomicAdd(&ph_acc[PH_QUEUES], now - given); atomicAdd(&ph_cnt[PH_QUEUES], 1u); }
            ph_tw[threadIdx.x >> 6] = __builtin_readcyclecounter();
        }
#endif

        // ---- 4. the batch: [RK4 half of the pending call for an R batch], then up to POOL_RUN Verlet attempts;
        //         a lane whose attempt is rejected stops (tag R), the wave goes back to the pool when an eighth
        //         of the batch has stopped.  Consecutive accepted steps stay in registers.
        int tag = TAG_EMPTY;
        bool finished = false;
        if (active) {
            tag = do_rk4 ? TAG_R : TAG_V;
            bool on = true;
#if !S5_MARCH_LEAN
            // by-value parameters (strict variant): the state of the ray in registers through the batch, as in round 3
            const TorusParams& p = A.p;
            const double* __restrict__ sc = A.start.d;
            const size_t scap = A.start.cap;
            const double r_in = cwd[1], r_out = cwd[2];
            RayState s;
            s.opt_gr = !((p.options & 1) == 1);
            s.opt_pol = 0;
            s.step_epsilon = cwd[0];
            s.bh_spin = p.a;
            s.refines = 0; s.Q = 0.0;
            double x[4], k[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                x[c] = pd[PD_AT(PC_X0 + c, slot)];
                k[c] = pd[PD_AT(PC_K0 + c, slot)];
                s.dk[c] = pd[PD_AT(PC_DK0 + c, slot)];
            }
            s.kt = pd[PD_AT(PC_KT, slot)];
            s.E = pd[PD_AT(PC_E, slot)];
            s.pass = ppass[slot];
            s.error = 0.0f;
            const size_t ray = (size_t)pray[slot];
            double I = pd[PD_AT(PC_I, slot)], tau = pd[PD_AT(PC_TAU, slot)];
            float worst = pworst[slot];
#endif
#pragma unroll 1
            for (int run = 0; run <= POOL_RUN; ++run) {
                if (on) {
                    const bool was_rk4 = (run == 0 && do_rk4);     // the RK4 half of the pending call, then Verlet attempts
                    double dl;
                    bool advanced;
                    Metric g;                                      // metric at the end point of the step
#if S5_MARCH_LEAN
                    // ---- the ray's dynamical state from its slot; the job's parameters from the argument segment, from here
                    const auto& P = param_reload(A);
                    RayState s;
                    s.opt_gr = !((P.p.options & 1) == 1);
                    s.opt_pol = 0;
                    s.step_epsilon = cwd[0];
                    s.bh_spin = P.p.a;
                    s.refines = 0; s.Q = 0.0; s.E = 0.0;
                    double x[4], k[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        x[c] = pd[PD_AT(PC_X0 + c, slot)];
                        k[c] = pd[PD_AT(PC_K0 + c, slot)];
                        s.dk[c] = pd[PD_AT(PC_DK0 + c, slot)];
                    }
                    s.kt = was_rk4 ? 0.0 : pd[PD_AT(PC_KT, slot)];      // (RK4: read after the step, below)
                    s.pass = ppass[slot];
                    s.error = 0.0f;
                    const double dl_max = P.p.dl_max;
#else
                    const double dl_max = p.dl_max;
#endif
                    ph.mark(PH_LOAD);
                    if (was_rk4) {
                        dl = next_step_size(k, dl_max, s);        // the value the rejected attempt used
                        ph.mark(PH_STEPSIZE);
#if S5_MARCH_LEAN
                        {
                            auto late = [&](int c) -> double { return pd[PD_AT(PC_X0 + c, slot)]; };
                            x[0] = 0.0; x[3] = 0.0;               // (t and phi are read from the slot at the end of the step)
                            rk4_step<true>(x, k, dl, s, g, late, ph);
                        }
#else
                        rk4_step(x, k, dl, s, g);
#endif
#if S5_MARCH_LEAN
                        // the step's error measures k_t against its value before the step (ref :306-308): read from the slot
                        // HERE (same expression, same operands as inside rk4_step) rather than carried through the four stages
                        s.error = (float)rel_diff(k[0] * g.g00 + k[3] * g.g03, pd[PD_AT(PC_KT, slot)]);
#endif
                        if (!s.opt_gr) flat_metric(x[1], x[2], g); // RK4 leaves the Kerr metric (ref :305); the fluid lives in the flat one
                        advanced = true;
                    } else {
                        advanced = verlet_attempt(x, k, dl_max, dl, s, g, ph);
                    }
#ifdef S5_TORUS_DEBUG
                    {
                        unsigned long long* dbg = (unsigned long long*)A.aux.k_end;
                        const unsigned long long mA = __builtin_amdgcn_ballot_w64(true);
                        const int w = was_rk4 ? 2 : 0;
                        if (ph_sampled && __builtin_amdgcn_mbcnt_hi((unsigned)(mA >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mA, 0u)) == 0u) {
                            atomicAdd(&dbg[w], 1ull); atomicAdd(&dbg[w + 1], (unsigned long long)__builtin_popcountll(mA));
                        }
                    }
#endif
#if S5_MARCH_LEAN
                    ppass[slot] = s.pass;                          // the attempt counts as a pass either way (ref :168)
#endif
                    if (!advanced) {
                        tag = TAG_R; on = false;                  // x, k, dk untouched
                    } else {
                        tag = TAG_V;
#if S5_MARCH_LEAN
                        // ---- accepted: the new state into the slot, then the transfer step and the end tests with what they need
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            pd[PD_AT(PC_X0 + c, slot)] = x[c];
                            pd[PD_AT(PC_K0 + c, slot)] = k[c];
                            pd[PD_AT(PC_DK0 + c, slot)] = s.dk[c];
                        }
                        if (!was_rk4) pd[PD_AT(PC_KT, slot)] = s.kt;      // (an RK4 step leaves kt as it was)
                        const auto& T = param_reload(A);
                        // Everything this phase reads of the job and of the pool is asked for HERE, in one go (round 6).  Read where
                        // it was used -- the torus model inside accumulate_transfer, the four operands of the end test behind its
                        // short-circuit branches -- every scalar load and LDS read was a round trip of its own with a wait behind it:
                        // the phase took 2 400 cycles, more than metric + connection, for ~100 vector instructions
                        // (profiles/r06_torus_call_phases.json); asked for together the round trips overlap.
                        const TorusModel tm = { T.p.shape, T.p.torus_w, T.p.torus_r, T.p.cut_d2, T.p.inv_2w2, T.p.torus_l, T.p.emis0, T.p.absorb0 };
                        const double max_error = T.p.max_error;
                        const int max_steps = T.p.max_steps;
                        const double r_in = cwd[1], r_out = cwd[2];
                        const float worst_before = pworst[slot];
                        double I = pd[PD_AT(PC_I, slot)], tau = pd[PD_AT(PC_TAU, slot)];
                        s.E = pd[PD_AT(PC_E, slot)];
                        keep_here(tm.shape); keep_here(tm.torus_w); keep_here(tm.torus_r); keep_here(tm.cut_d2); keep_here(tm.inv_2w2);
                        keep_here(tm.torus_l); keep_here(tm.emis0); keep_here(tm.absorb0); keep_here(max_error); keep_here(max_steps);
                        const float worst = fmaxf(worst_before, s.error);
                        pworst[slot] = worst;
                        accumulate_transfer(tm, no_absorption, s, g, x, k, dl, I, tau);
                        pd[PD_AT(PC_I, slot)] = I;
                        pd[PD_AT(PC_TAU, slot)] = tau;
                        // (no short circuit: four comparisons and three ORs instead of three branches)
                        const bool done = (!(x[1] > r_in)) | (!(x[1] < r_out)) | ((double)s.error > max_error) | (s.pass >= max_steps);
                        if (done) {
                            const size_t ray = (size_t)pray[slot];
                            s.Q = T.start.d[COL_Q * T.start.cap + ray];
                            write_ray_end(T.aux, T.out, ray, x, k, s, I, tau, worst);
                            tag = TAG_EMPTY; on = false; finished = true;
                        }
#else
                        worst = fmaxf(worst, s.error);
                        accumulate_transfer(p, no_absorption, s, g, x, k, dl, I, tau);
                        const bool done = !(x[1] > r_in) || !(x[1] < r_out) || ((double)s.error > p.max_error) ||
                                          (s.pass >= p.max_steps);
                        if (done) {
                            s.Q = sc[COL_Q * scap + ray];
                            write_ray_end(A.aux, A.out, ray, x, k, s, I, tau, worst);
                            tag = TAG_EMPTY; on = false; finished = true;
                        }
#endif
                    }
                    ph.mark(PH_STORE_TRANSFER);
                    ph.mark(PH_N);                                 // (two marks with nothing in between: the cost a mark leaves in an interval)
                }
                if (POOL_KEEP_DEN * __builtin_popcountll(__builtin_amdgcn_ballot_w64(on)) < POOL_KEEP_NUM * take) break;
            }
#if !S5_MARCH_LEAN
            if (tag != TAG_EMPTY) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    pd[PD_AT(PC_X0 + c, slot)] = x[c];
                    pd[PD_AT(PC_K0 + c, slot)] = k[c];
                    pd[PD_AT(PC_DK0 + c, slot)] = s.dk[c];
                }
                pd[PD_AT(PC_KT, slot)] = s.kt;
                pd[PD_AT(PC_I, slot)] = I;
                pd[PD_AT(PC_TAU, slot)] = tau;
                ppass[slot] = s.pass;
                pworst[slot] = worst;
            }
#endif
        }
        // ---- 5. the slots back into their queues (the state before the entries)
#ifdef S5_TORUS_DEBUG
        if (active) ph_slot[slot] = (tag != TAG_EMPTY) ? __builtin_readcyclecounter() : 0ull;
#endif
        wg_release();
        q_give(Q_V, active && tag == TAG_V, slot);
        q_give(Q_R, active && tag == TAG_R, slot);
        q_give(Q_E, active && tag == TAG_EMPTY, slot);
        {
            const int nf = __builtin_popcountll(__builtin_amdgcn_ballot_w64(finished));
            if (nf > 0 && lane == 0) atomicSub(&cw[CW_LIVE], (unsigned)nf);
        }
    }
#ifdef S5_TORUS_DEBUG
    if (lane == 0) tl[2] = wall_clock64();
    __syncthreads();                                                 // (debug build: every wave has left the loop)
    if (threadIdx.x <= PH_N) {
        unsigned long long* dbg = (unsigned long long*)A.aux.k_end + PHASE_DBG_AT;
        atomicAdd(&dbg[threadIdx.x], ph_acc[threadIdx.x]);
        atomicAdd(&dbg[PH_N + 1 + threadIdx.x], (unsigned long long)ph_cnt[threadIdx.x]);
    }
    if (threadIdx.x == 0) {                                          // the two clocks over the workgroup's life: cycles per 100 MHz tick
        unsigned long long* dbg = (unsigned long long*)A.aux.k_end + PHASE_DBG_AT + 2 * (PH_N + 1);
        atomicAdd(&dbg[0], __builtin_readcyclecounter() - ph_t0);
        atomicAdd(&dbg[1], wall_clock64() - t_begin);
    }
#endif
}

// Workspace of the torus job (start state of every ray, start-up flags, cursor): one grow-only device
// allocation per process, made before the launches (no allocation call sits between kernels).  A job on
// another stream than the previous one first waits for that stream, so two jobs never share it.
struct TorusWorkspace {
    char* base = nullptr;
    size_t cap = 0;
    hipStream_t last = nullptr;
    bool used = false;
    bool attr_set = false;         // dynamic-LDS limit of the march kernel raised on this device
};
constexpr int MAX_DEVICES = 64;
static TorusWorkspace g_ws_dev[MAX_DEVICES];   // one workspace per device: a process that switches device never hands
                                               // a kernel on GPU B memory that lives on GPU A
static std::mutex g_ws_lock;       // jobs from several host threads take turns at the shared workspace

#if !S5_FAST
// the start kernel of this variant for the rays the fast job marked START_AGAIN (torus_start_kernel)
hipError_t launch_torus_start_again_strict(const TorusParams& p, double* cols, size_t cap, int* ok, hipStream_t stream)
{
    RayCols start;
    start.d = cols; start.cap = cap;
    hipLaunchKernelGGL(torus_start_kernel, dim3((unsigned)((p.nrays + 255) / 256)), dim3(256), 0, stream, p, start, ok, (int*)nullptr,
                       (unsigned long long*)nullptr, 1);
    return hipGetLastError();
}
#endif

#if S5_FAST
int launch_torus_fast(const TorusParams& p, sim5gpu_stokes* out, const TorusAux& aux, hipStream_t stream)
#else
int launch_torus_strict(const TorusParams& p, sim5gpu_stokes* out, const TorusAux& aux, hipStream_t stream)
#endif
{
    std::lock_guard<std::mutex> hold(g_ws_lock);
    int dev = 0;
    hipError_t e;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return (int)e;
    if (dev < 0 || dev >= MAX_DEVICES) return (int)hipErrorInvalidDevice;
    TorusWorkspace& g_ws = g_ws_dev[dev];
    const size_t n = p.nrays;
    if (n > 0x3ffffff0ull) return (int)hipErrorInvalidValue;            // ray numbers are kept as int in the pool, class and rank in 32 bits
    const size_t dcol_bytes = (sizeof(double) * NCOL * n + 255) & ~size_t(255);
    const size_t ok_bytes = (sizeof(int) * n + 255) & ~size_t(255);
    const size_t need = dcol_bytes + 3 * ok_bytes + 256;        // start columns, start-up flags, class ranks, order of the march, cursor + 2 counters
    if (g_ws.used && g_ws.last != stream) {
        if ((e = hipStreamSynchronize(g_ws.last)) != hipSuccess) return (int)e;
    }
    if (need > g_ws.cap) {
        if (g_ws.base) {
            if ((e = hipDeviceSynchronize()) != hipSuccess) return (int)e;
            (void)hipFree(g_ws.base);
            g_ws.base = nullptr; g_ws.cap = 0;
        }
        if ((e = hipMalloc((void**)&g_ws.base, need)) != hipSuccess) return (int)e;
        g_ws.cap = need;
    }
    g_ws.last = stream; g_ws.used = true;
    RayCols start;
    start.d = (double*)g_ws.base;
    start.cap = n;
    int* ok = (int*)(g_ws.base + dcol_bytes);
    int* ranks = (int*)(g_ws.base + dcol_bytes + ok_bytes);
    int* order = (int*)(g_ws.base + dcol_bytes + 2 * ok_bytes);
    unsigned long long* cursor = (unsigned long long*)(g_ws.base + dcol_bytes + 3 * ok_bytes);
    if ((e = hipMemsetAsync(cursor, 0, 4 * sizeof(unsigned long long), stream)) != hipSuccess) return (int)e;

    const unsigned blocks_a = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(torus_start_kernel, dim3(blocks_a), dim3(256), 0, stream, p, start, ok, ranks, cursor + 1, 0);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
#if S5_FAST
    if ((e = s5::launch_torus_start_again_strict(p, start.d, start.cap, ok, stream)) != hipSuccess) return (int)e;
#endif
    hipLaunchKernelGGL(torus_order_kernel, dim3(blocks_a), dim3(256), 0, stream, n, ranks, cursor + 1, order);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;

    // persistent grid: the waves a CU holds (4 SIMDs x S5_MARCH_WAVES) in workgroups of WG_WAVES -- fast: one 12-wave workgroup
    // per CU, 129 KB of LDS; never more workgroups than pools to fill
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    size_t blocks_b = (size_t)cus * S5_MARCH_WAVES * 4 / WG_WAVES;
    const size_t needed = (n + WG_SLOTS - 1) / WG_SLOTS;
    if (blocks_b > needed) blocks_b = needed;
    const size_t lds = (size_t)POOL_WG_BYTES + (size_t)PHASE_LDS_BYTES;
    if (!g_ws.attr_set) {
        if ((e = hipFuncSetAttribute((const void*)torus_pool_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return (int)e;
        g_ws.attr_set = true;
    }
    PoolArgs pa;
    pa.p = p; pa.start = start; pa.ok = ok; pa.order = order; pa.cursor = cursor; pa.out = out; pa.aux = aux;
    hipLaunchKernelGGL(torus_pool_kernel, dim3((unsigned)blocks_b), dim3(WG_THREADS), lds, stream, pa);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    return 0;
}

// give the grow-only workspaces of every device back (sim5gpu_release_workspaces): waits for the device that owns a block
#if S5_FAST
size_t release_torus_workspace_fast()
#else
size_t release_torus_workspace_strict()
#endif
{
    std::lock_guard<std::mutex> hold(g_ws_lock);
    int cur = 0;
    (void)hipGetDevice(&cur);
    size_t freed = 0;
    for (int d = 0; d < MAX_DEVICES; ++d) {
        TorusWorkspace& w = g_ws_dev[d];
        if (!w.base) continue;
        (void)hipSetDevice(d);
        (void)hipDeviceSynchronize();
        (void)hipFree(w.base);
        freed += w.cap;
        w.base = nullptr; w.cap = 0; w.used = false; w.last = nullptr;
    }
    (void)hipSetDevice(cur);
    return freed;
}

} // namespace S5NS
