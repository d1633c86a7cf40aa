// k_torus.hip -- step-wise (Verlet) ray tracer with radiative transfer through an optically thin
// torus, gfx950.
//
// Two kernels:
//  (A) torus_start_kernel: one lane per pixel sets up the ray where it enters the integration
//      domain: geodesic_init_inf -> geodesic_P_int(r0, before pericentre) -> geodesic_position_pol
//      -> geodesic_momentum -> raytrace_prepare (ref src/sim5kerr-geod.c:42,179,363,787;
//      src/sim5raytrace.c:44).  The state goes to HBM as a structure of arrays (one 8-B column
//      per quantity, so every load/store below is a coalesced 512-B wave access).
//  (B) torus_round_kernel: the rays march in global lock-step, K raytrace() calls (ref
//      src/sim5raytrace.c:109-245) per launch ("round"), accumulating the transfer integral after each
//      accepted step.  A round reads the state of the rays that are still alive from one SoA buffer and
//      writes the survivors, compacted, to the other (wave ballot, ONE atomic per wave reserves the
//      slots, lanes take their prefix rank; runs of image neighbours stay together).  Why lock-step:
//      raytrace() is a Verlet attempt plus, when its precision check fails, an RK4 step of the same
//      size, and a ray fails for long stretches (near the hole P(RK4 | previous RK4) = 0.9-1.0, elsewhere
//      P(RK4 | previous Verlet) = 0.02; a third of all calls at precision 1).  Image neighbours at the
//      SAME step index are in the same regime, so a wave either skips the RK4 body or runs it with most
//      lanes.  The earlier persistent kernel refilled idle lanes one by one from a cursor; that keeps
//      lanes busy but mixes rays at unrelated step indices in a wave, which then pays Verlet + RK4 on
//      nearly every call with a third of the lanes active in RK4 (measured VALU lane utilisation 54 %).
//      Every launch is bounded (K steps), there is no persistent loop and no inter-wave waiting.
//
// Transfer model (the reference has no transfer integrator nor torus; this is
// this project's definition, stated in DESIGN.md): fluid on circular orbits with constant specific
// angular momentum ell; density rho = exp(-((R-R_t)^2 + z^2)/(2 w^2)), R = r sin(theta),
// z = r cos(theta) (shape 0), or rho = 1 inside a sphere of radius w (shape 1, analytic test case);
// with k normalised to k_t = -1 at infinity the local photon energy is -k.U = 1/g, the proper length
// of a step of affine size dl is dl/g, and per accepted step
//      dtau = absorb0 * rho * dl / g,     dI = g^4 * emis0 * rho * exp(-tau) * dl / g .
#include "s5_disk.hpp"
#include "s5_raytrace.hpp"
#include "k_torus.hpp"

namespace S5NS {

using namespace s5abi;

enum : int { COL_X0 = 0, COL_X1, COL_X2, COL_X3, COL_K0, COL_K1, COL_K2, COL_K3,
             COL_DK0, COL_DK1, COL_DK2, COL_DK3, COL_KT, COL_Q, COL_E, COL_I, COL_TAU, NCOL };
// integer / float columns of the state (4 B each), after the NCOL double columns of a buffer
enum : int { ICOL_RAY = 0, ICOL_PASS, ICOL_WORST, NICOL };

struct RayCols {                 // one SoA state buffer: column c of ray slot i is d[c * cap + i]
    double* d;
    int* i32;
    size_t cap;
};

__global__ __launch_bounds__(256, 2)
void torus_start_kernel(TorusParams p, RayCols st, int* __restrict__ ok)
{
    double* __restrict__ cols = st.d;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.nrays) return;
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
    const size_t n = st.cap;
    cols[COL_X0 * n + i] = x[0]; cols[COL_X1 * n + i] = x[1]; cols[COL_X2 * n + i] = x[2]; cols[COL_X3 * n + i] = x[3];
    cols[COL_K0 * n + i] = k[0]; cols[COL_K1 * n + i] = k[1]; cols[COL_K2 * n + i] = k[2]; cols[COL_K3 * n + i] = k[3];
    cols[COL_DK0 * n + i] = s.dk[0]; cols[COL_DK1 * n + i] = s.dk[1];
    cols[COL_DK2 * n + i] = s.dk[2]; cols[COL_DK3 * n + i] = s.dk[3];
    cols[COL_KT * n + i] = s.kt; cols[COL_Q * n + i] = s.Q;
    cols[COL_E * n + i] = s.kt; cols[COL_I * n + i] = 0.0; cols[COL_TAU * n + i] = 0.0;
    st.i32[ICOL_RAY * n + i] = (int)i; st.i32[ICOL_PASS * n + i] = 0;
    ((float*)st.i32)[ICOL_WORST * n + i] = 0.0f;
    ok[i] = good;
}

S5_DEV double torus_density(const TorusParams& p, double r, double m)
{
    if (p.shape == 1) return (r <= p.torus_w) ? 1.0 : 0.0;
    const double R = r * sqrt(1. - m * m), z = r * m;
    const double d2 = sq(R - p.torus_r) + z * z;
    const double w2 = 2. * p.torus_w * p.torus_w;
    return (d2 < 36. * w2) ? exp(-d2 / w2) : 0.0;       // cut at 6 sqrt(2) w: exp(-36) ~ 2e-16
}

#ifndef S5_MARCH_WAVES
#define S5_MARCH_WAVES 2
#endif
// Emission and absorption picked up over one accepted step (see the header comment for the model).
S5_DEV void accumulate_transfer(const TorusParams& p, const RayState& s, const double x[4], const double k[4],
                         double dl_taken, double& I, double& tau)
{
    const double rho = torus_density(p, x[1], x[2]);
    if (!(rho > 0.0)) return;
    Metric g;
    rt_metric(s, x[1], x[2], g);
    const double Om = omega_from_ell(p.torus_l, g);
    const double nrm = -(g.g00 + 2. * Om * g.g03 + Om * Om * g.g33);
    if (!(nrm > 0.0)) return;                           // no time-like circular orbit with this ell here
    const double ut = mdiv(1., msqrt(nrm));
    const double k_t = k[0] * g.g00 + k[3] * g.g03;
    const double k_f = k[3] * g.g33 + k[0] * g.g03;
    const double gfac = mdiv(s.E, ut * (k_t + Om * k_f));   // E_inf / E_local
    const double ds = mdiv(dl_taken, gfac);
    const double g2 = gfac * gfac;
    I += (g2 * g2) * p.emis0 * rho * exp(-tau) * ds;
    tau += p.absorb0 * rho * ds;
}

S5_DEV void write_ray_end(const TorusParams& p, const TorusAux& aux, sim5gpu_stokes* __restrict__ out, size_t ray,
                          const double x[4], const double k[4], const RayState& s, double I, double tau, float worst)
{
    sim5gpu_stokes rec = { I, 0.0, 0.0, 0.0, tau };
    out[ray] = rec;
    if (aux.steps) aux.steps[ray] = s.pass;
    if (aux.max_step_error) aux.max_step_error[ray] = worst;
    if (aux.carter_error) aux.carter_error[ray] = raytrace_error(x, k, s);
    if (aux.x_end) { for (int c = 0; c < 4; ++c) aux.x_end[4 * ray + c] = x[c]; }
#ifndef S5_TORUS_DEBUG
    if (aux.k_end) { for (int c = 0; c < 4; ++c) aux.k_end[4 * ray + c] = k[c]; }
#endif
}

// One round: lanes [0, *n_in) of `src` advance by at most `k_steps` raytrace() calls; survivors go, compacted,
// to `dst` and are counted in *n_out (zeroed by the host before the launch).  `first` marks round 0, which
// also writes the empty records of the rays the start kernel rejected.
__global__ __launch_bounds__(256, S5_MARCH_WAVES)
void torus_round_kernel(TorusParams p, RayCols src, RayCols dst, const unsigned* __restrict__ n_in,
                        unsigned* __restrict__ n_out, const int* __restrict__ ok, int first, int k_steps,
                        sim5gpu_stokes* __restrict__ out, TorusAux aux)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t count = (size_t)*n_in;
    const double r_in = p.r_stop_in * r_horizon(p.a);
    const double r_out = p.r_stop_out * p.r0;
    bool alive = i < count;

    size_t ray = 0;
    double x[4] = { 0.0, 0.0, 0.0, 0.0 }, k[4] = { 0.0, 0.0, 0.0, 0.0 };
    RayState s;
    double I = 0.0, tau = 0.0;
    float worst = 0.0f;
    s.opt_gr = !((p.options & 1) == 1);
    s.opt_pol = 0;
    s.step_epsilon = S5_DIVC(msqrt(p.precision), 10.);
    s.bh_spin = p.a;
    s.refines = 0; s.error = 0.0f; s.pass = 0;
    s.dk[0] = s.dk[1] = s.dk[2] = s.dk[3] = 0.0; s.kt = 0.0; s.E = 0.0; s.Q = 0.0;
    if (alive) {
        const size_t n = src.cap;
        const double* __restrict__ c = src.d;
        ray = (size_t)src.i32[ICOL_RAY * n + i];
        s.pass = src.i32[ICOL_PASS * n + i];
        worst = ((const float*)src.i32)[ICOL_WORST * n + i];
        x[0] = c[COL_X0 * n + i]; x[1] = c[COL_X1 * n + i]; x[2] = c[COL_X2 * n + i]; x[3] = c[COL_X3 * n + i];
        k[0] = c[COL_K0 * n + i]; k[1] = c[COL_K1 * n + i]; k[2] = c[COL_K2 * n + i]; k[3] = c[COL_K3 * n + i];
        s.dk[0] = c[COL_DK0 * n + i]; s.dk[1] = c[COL_DK1 * n + i];
        s.dk[2] = c[COL_DK2 * n + i]; s.dk[3] = c[COL_DK3 * n + i];
        s.kt = c[COL_KT * n + i]; s.Q = c[COL_Q * n + i]; s.E = c[COL_E * n + i];
        I = c[COL_I * n + i]; tau = c[COL_TAU * n + i];
        if (first && !ok[ray]) {
            // ray rejected at start-up: an empty record, no steps
            sim5gpu_stokes z = { 0.0, 0.0, 0.0, 0.0, 0.0 };
            out[ray] = z;
            if (aux.steps) aux.steps[ray] = 0;
            if (aux.max_step_error) aux.max_step_error[ray] = 0.0f;
            if (aux.carter_error) aux.carter_error[ray] = NAN;
            if (aux.x_end) { for (int cc = 0; cc < 4; ++cc) aux.x_end[4 * ray + cc] = x[cc]; }
#ifndef S5_TORUS_DEBUG
            if (aux.k_end) { for (int cc = 0; cc < 4; ++cc) aux.k_end[4 * ray + cc] = k[cc]; }
#endif
            alive = false;
        }
    }

#pragma unroll 1
    for (int step = 0; step < k_steps; ++step) {
        if (!wave_any(alive)) break;
        if (alive) {
            // raytrace() = Verlet attempt and, if its precision check fails, an RK4 step of the same size
            // (ref src/sim5raytrace.c:220-227)
            double dl;
#ifdef S5_TORUS_DEBUG            // lane-occupancy counters (scratch builds only): aux.k_end is the counter block
            const bool v_ok = verlet_attempt(x, k, p.dl_max, dl, s);
            {
                unsigned long long* dbg = (unsigned long long*)aux.k_end;
                const unsigned long long mA = __builtin_amdgcn_ballot_w64(true), mR = __builtin_amdgcn_ballot_w64(!v_ok);
                const double rho_dbg = torus_density(p, x[1], x[2]);
                const unsigned long long mT = __builtin_amdgcn_ballot_w64(rho_dbg > 0.0);
                if (__builtin_amdgcn_mbcnt_hi((unsigned)(mA >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mA, 0u)) == 0u) {
                    atomicAdd(&dbg[0], 1ull); atomicAdd(&dbg[1], (unsigned long long)__builtin_popcountll(mA));
                    if (mR) { atomicAdd(&dbg[2], 1ull); atomicAdd(&dbg[3], (unsigned long long)__builtin_popcountll(mR)); }
                    if (mT) { atomicAdd(&dbg[4], 1ull); atomicAdd(&dbg[5], (unsigned long long)__builtin_popcountll(mT)); }
                }
            }
            if (!v_ok) rk4_step(x, k, dl, s);
#else
            if (!verlet_attempt(x, k, p.dl_max, dl, s)) rk4_step(x, k, dl, s);
#endif
            worst = fmaxf(worst, s.error);
            // transfer over the step just taken, evaluated at its end point
            accumulate_transfer(p, s, x, k, dl, I, tau);
            const bool done = !(x[1] > r_in) || !(x[1] < r_out) || ((double)s.error > p.max_error) ||
                              (s.pass >= p.max_steps);
            if (done) {
                write_ray_end(p, aux, out, ray, x, k, s, I, tau, worst);
                alive = false;
            }
        }
    }

    // ---- survivors to the other buffer, compacted (wave-aggregated reservation) ----
    const unsigned long long live = __builtin_amdgcn_ballot_w64(alive);
    if (live) {
        const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(live >> 32),
                                  __builtin_amdgcn_mbcnt_lo((unsigned)live, 0u));
        const int leader = __builtin_ctzll(live);
        unsigned base = 0;
        if (alive && rank == 0u) base = atomicAdd(n_out, (unsigned)__builtin_popcountll(live));
        base = (unsigned)__shfl((int)base, leader, 64);
        if (alive) {
            const size_t n = dst.cap, j = (size_t)base + rank;
            double* __restrict__ c = dst.d;
            dst.i32[ICOL_RAY * n + j] = (int)ray;
            dst.i32[ICOL_PASS * n + j] = s.pass;
            ((float*)dst.i32)[ICOL_WORST * n + j] = worst;
            c[COL_X0 * n + j] = x[0]; c[COL_X1 * n + j] = x[1]; c[COL_X2 * n + j] = x[2]; c[COL_X3 * n + j] = x[3];
            c[COL_K0 * n + j] = k[0]; c[COL_K1 * n + j] = k[1]; c[COL_K2 * n + j] = k[2]; c[COL_K3 * n + j] = k[3];
            c[COL_DK0 * n + j] = s.dk[0]; c[COL_DK1 * n + j] = s.dk[1];
            c[COL_DK2 * n + j] = s.dk[2]; c[COL_DK3 * n + j] = s.dk[3];
            c[COL_KT * n + j] = s.kt; c[COL_Q * n + j] = s.Q; c[COL_E * n + j] = s.E;
            c[COL_I * n + j] = I; c[COL_TAU * n + j] = tau;
        }
    }
}

// Workspace of the torus job (two ray-state buffers, start-up flags, two counters): one grow-only device
// allocation per process plus one pinned host word for the per-round count.  A job on another stream than
// the previous one first waits for that stream, so two jobs never share it.
struct TorusWorkspace {
    char* base = nullptr;
    size_t cap = 0;
    unsigned* host_count = nullptr;
    hipStream_t last = nullptr;
    bool used = false;
};
static TorusWorkspace g_ws;

#ifndef S5_ROUND_STEPS
#define S5_ROUND_STEPS 32
#endif

// The job synchronises `stream` once per round (it needs the survivor count to size the next launch): on
// return all results are complete.
#if S5_FAST
int launch_torus_fast(const TorusParams& p, sim5gpu_stokes* out, const TorusAux& aux, hipStream_t stream)
#else
int launch_torus_strict(const TorusParams& p, sim5gpu_stokes* out, const TorusAux& aux, hipStream_t stream)
#endif
{
    const size_t n = p.nrays;
    if (n > 0xfffffff0ull) return (int)hipErrorInvalidValue;            // ray slots are 32-bit
    const size_t dcol_bytes = (sizeof(double) * NCOL * n + 255) & ~size_t(255);
    const size_t icol_bytes = (sizeof(int) * NICOL * n + 255) & ~size_t(255);
    const size_t ok_bytes = (sizeof(int) * n + 255) & ~size_t(255);
    const size_t need = 2 * (dcol_bytes + icol_bytes) + ok_bytes + 256;
    hipError_t e;
    if (g_ws.used && g_ws.last != stream) {
        if ((e = hipStreamSynchronize(g_ws.last)) != hipSuccess) return (int)e;
    }
    if (!g_ws.host_count) {
        if ((e = hipHostMalloc((void**)&g_ws.host_count, 64, hipHostMallocDefault)) != hipSuccess) return (int)e;
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
    RayCols buf[2];
    char* q = g_ws.base;
    for (int b = 0; b < 2; ++b) {
        buf[b].d = (double*)q; q += dcol_bytes;
        buf[b].i32 = (int*)q; q += icol_bytes;
        buf[b].cap = n;
    }
    int* ok = (int*)q; q += ok_bytes;
    unsigned* counts = (unsigned*)q;                                     // counts[0], counts[1]

    const unsigned blocks_a = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(torus_start_kernel, dim3(blocks_a), dim3(256), 0, stream, p, buf[0], ok);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;

    unsigned alive = (unsigned)n;
    if ((e = hipMemcpyAsync(&counts[0], &alive, sizeof(unsigned), hipMemcpyHostToDevice, stream)) != hipSuccess) return (int)e;
    int cur = 0;
    // every ray takes at least one step per round it is alive in and at most max_steps in all
    const long long max_rounds = (long long)p.max_steps + 2;
    for (long long round = 0; round < max_rounds && alive > 0; ++round) {
        // short rounds while most rays are alive (less idling behind rays that end inside a round), longer
        // ones for the thin tail of long rays (fewer launches)
        const int k_steps = (alive > n / 16) ? S5_ROUND_STEPS : 8 * S5_ROUND_STEPS;
        if ((e = hipMemsetAsync(&counts[cur ^ 1], 0, sizeof(unsigned), stream)) != hipSuccess) return (int)e;
        const unsigned blocks = (unsigned)(((size_t)alive + 255) / 256);
        hipLaunchKernelGGL(torus_round_kernel, dim3(blocks), dim3(256), 0, stream, p, buf[cur], buf[cur ^ 1],
                           &counts[cur], &counts[cur ^ 1], ok, (int)(round == 0), k_steps, out, aux);
        if ((e = hipGetLastError()) != hipSuccess) return (int)e;
        if ((e = hipMemcpyAsync(g_ws.host_count, &counts[cur ^ 1], sizeof(unsigned), hipMemcpyDeviceToHost, stream)) != hipSuccess) return (int)e;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return (int)e;
        alive = *g_ws.host_count;
        cur ^= 1;
    }
    return 0;
}

} // namespace S5NS
