// k_torus.hip -- step-wise (Verlet) ray tracer with radiative transfer through an optically thin
// torus, gfx950.
//
// Two kernels:
//  (A) torus_start_kernel: one lane per pixel sets up the ray where it enters the integration
//      domain: geodesic_init_inf -> geodesic_P_int(r0, before pericentre) -> geodesic_position_pol
//      -> geodesic_momentum -> raytrace_prepare (ref src/sim5kerr-geod.c:42,179,363,787;
//      src/sim5raytrace.c:44).  The state goes to HBM as a structure of arrays (one 8-B column
//      per quantity, so every load/store below is a coalesced 512-B wave access).
//  (B) torus_march_kernel: persistent waves advance rays with raytrace() (ref
//      src/sim5raytrace.c:109-245), accumulating the transfer integral after each accepted step.
//      Rays end after very different step counts, so a lane whose ray has ended takes the next
//      unprocessed ray from a global cursor: idle lanes are counted by a wave ballot, one atomic
//      per wave reserves that many rays, and each idle lane takes its rank (prefix count of the
//      ballot) within the reservation.  A wave leaves when the cursor is exhausted and none of its
//      lanes holds a ray, which every wave reaches because the cursor only grows.
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
             COL_DK0, COL_DK1, COL_DK2, COL_DK3, COL_KT, COL_Q, NCOL };

__global__ __launch_bounds__(256, 2)
void torus_start_kernel(TorusParams p, double* __restrict__ cols, int* __restrict__ ok)
{
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
    const size_t n = p.nrays;
    cols[COL_X0 * n + i] = x[0]; cols[COL_X1 * n + i] = x[1]; cols[COL_X2 * n + i] = x[2]; cols[COL_X3 * n + i] = x[3];
    cols[COL_K0 * n + i] = k[0]; cols[COL_K1 * n + i] = k[1]; cols[COL_K2 * n + i] = k[2]; cols[COL_K3 * n + i] = k[3];
    cols[COL_DK0 * n + i] = s.dk[0]; cols[COL_DK1 * n + i] = s.dk[1];
    cols[COL_DK2 * n + i] = s.dk[2]; cols[COL_DK3 * n + i] = s.dk[3];
    cols[COL_KT * n + i] = s.kt; cols[COL_Q * n + i] = s.Q;
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

__global__ __launch_bounds__(256, S5_MARCH_WAVES)
void torus_march_kernel(TorusParams p, const double* __restrict__ cols, const int* __restrict__ ok,
                        unsigned long long* __restrict__ cursor, sim5gpu_stokes* __restrict__ out,
                        TorusAux aux)
{
    const size_t n = p.nrays;
    const double r_in = p.r_stop_in * r_horizon(p.a);
    const double r_out = p.r_stop_out * p.r0;

    bool holding = false;          // this lane owns a ray
    bool drained = false;          // the cursor has run past the last ray
    size_t ray = 0;
    double x[4], k[4];
    RayState s;
    double I = 0.0, tau = 0.0;
    float worst = 0.0f;

    s.opt_gr = !((p.options & 1) == 1);
    s.opt_pol = 0;
    s.step_epsilon = S5_DIVC(msqrt(p.precision), 10.);
    s.bh_spin = p.a;

    // the loop is bounded: every pass either advances a held ray by one step (at most max_steps
    // per ray) or consumes cursor positions; `guard` is a belt-and-braces cap
    const unsigned long long guard = (unsigned long long)(p.max_steps + 2) * ((n + 63) / 64 + 1);
    for (unsigned long long pass = 0; pass < guard; ++pass) {
        // ---- refill idle lanes from the cursor (wave-aggregated reservation) ----
        const bool want = !holding && !drained;
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(want);
        if (idle) {
            // rank = number of idle lanes below this one (prefix count of the ballot)
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32),
                                      __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
            const int leader = __builtin_ctzll(idle);
            unsigned long long base = 0;
            if (want && rank == 0u)                                       // lowest idle lane reserves
                base = atomicAdd(cursor, (unsigned long long)__builtin_popcountll(idle));
            base = ((unsigned long long)(unsigned)__shfl((int)(base >> 32), leader, 64) << 32) |
                   (unsigned long long)(unsigned)__shfl((int)(unsigned)base, leader, 64);
            if (want) {
                const unsigned long long mine = base + rank;
                if (mine >= n) {
                    drained = true;
                } else {
                    ray = (size_t)mine;
                    x[0] = cols[COL_X0 * n + ray]; x[1] = cols[COL_X1 * n + ray];
                    x[2] = cols[COL_X2 * n + ray]; x[3] = cols[COL_X3 * n + ray];
                    k[0] = cols[COL_K0 * n + ray]; k[1] = cols[COL_K1 * n + ray];
                    k[2] = cols[COL_K2 * n + ray]; k[3] = cols[COL_K3 * n + ray];
                    s.dk[0] = cols[COL_DK0 * n + ray]; s.dk[1] = cols[COL_DK1 * n + ray];
                    s.dk[2] = cols[COL_DK2 * n + ray]; s.dk[3] = cols[COL_DK3 * n + ray];
                    s.kt = cols[COL_KT * n + ray]; s.E = s.kt; s.Q = cols[COL_Q * n + ray];
                    s.pass = 0; s.refines = 0; s.error = 0.0f;
                    I = 0.0; tau = 0.0; worst = 0.0f;
                    holding = true;
                    if (!ok[ray]) {
                        // ray rejected at start-up: write an empty record now
                        sim5gpu_stokes z = { 0.0, 0.0, 0.0, 0.0, 0.0 };
                        out[ray] = z;
                        if (aux.steps) aux.steps[ray] = 0;
                        if (aux.max_step_error) aux.max_step_error[ray] = 0.0f;
                        if (aux.carter_error) aux.carter_error[ray] = NAN;
                        if (aux.x_end) { for (int c = 0; c < 4; ++c) aux.x_end[4 * ray + c] = x[c]; }
                        if (aux.k_end) { for (int c = 0; c < 4; ++c) aux.k_end[4 * ray + c] = k[c]; }
                        holding = false;
                    }
                }
            }
        }
        if (!wave_any(holding)) {
            if (!wave_any(!drained)) break;       // nothing held, nothing left: the wave retires
            continue;                             // some lane just dropped a rejected ray: refill again
        }

        // ---- one raytrace() call for every lane that holds a ray --------------------------------------
        // raytrace() = Verlet attempt and, if its precision check fails, an RK4 step of the same size
        // (ref src/sim5raytrace.c:220-227; at precision 1 a third of all calls fall back).  Parking the
        // failed lanes until many of them can run the RK4 body together was measured and is not faster
        // (73.7 vs 71.5 ms on the C4 job): the kernel is limited by register pressure, not divergence.
        bool stepped = false;
        double dl_taken = 0.0;
        if (holding) {
            double dl;
            if (!verlet_attempt(x, k, p.dl_max, dl, s)) rk4_step(x, k, dl, s);
            stepped = true; dl_taken = dl;
        }

        if (stepped) {
            worst = fmaxf(worst, s.error);
            // transfer over the step just taken, evaluated at its end point
            accumulate_transfer(p, s, x, k, dl_taken, I, tau);

            const bool done = !(x[1] > r_in) || !(x[1] < r_out) || ((double)s.error > p.max_error) ||
                              (s.pass >= p.max_steps);
            if (done) {
                sim5gpu_stokes rec = { I, 0.0, 0.0, 0.0, tau };
                out[ray] = rec;
                if (aux.steps) aux.steps[ray] = s.pass;
                if (aux.max_step_error) aux.max_step_error[ray] = worst;
                if (aux.carter_error) aux.carter_error[ray] = raytrace_error(x, k, s);
                if (aux.x_end) { for (int c = 0; c < 4; ++c) aux.x_end[4 * ray + c] = x[c]; }
                if (aux.k_end) { for (int c = 0; c < 4; ++c) aux.k_end[4 * ray + c] = k[c]; }
                holding = false;
            }
        }
    }
}

// Workspace of the torus job (ray-state columns, start-up flags, cursor): one grow-only device
// allocation per process, made before the launches (no allocation call sits between kernels).  A job on
// another stream than the previous one first waits for that stream, so two jobs never share it.
struct TorusWorkspace {
    char* base = nullptr;
    size_t cap = 0;
    hipStream_t last = nullptr;
    bool used = false;
};
static TorusWorkspace g_ws;

#if S5_FAST
int launch_torus_fast(const TorusParams& p, sim5gpu_stokes* out, const TorusAux& aux, hipStream_t stream)
#else
int launch_torus_strict(const TorusParams& p, sim5gpu_stokes* out, const TorusAux& aux, hipStream_t stream)
#endif
{
    const size_t n = p.nrays;
    const size_t cols_bytes = (sizeof(double) * NCOL * n + 255) & ~size_t(255);
    const size_t ok_bytes = (sizeof(int) * n + 255) & ~size_t(255);
    const size_t need = cols_bytes + ok_bytes + 256;
    hipError_t e;
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
    double* cols = (double*)g_ws.base;
    int* ok = (int*)(g_ws.base + cols_bytes);
    unsigned long long* cursor = (unsigned long long*)(g_ws.base + cols_bytes + ok_bytes);
    if ((e = hipMemsetAsync(cursor, 0, sizeof(unsigned long long), stream)) != hipSuccess) return (int)e;

    const unsigned blocks_a = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(torus_start_kernel, dim3(blocks_a), dim3(256), 0, stream, p, cols, ok);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;

    // persistent grid: 2 workgroups of 4 waves per CU (occupancy 2 waves/SIMD), never more waves than rays
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    size_t blocks_b = (size_t)cus * S5_MARCH_WAVES;
    const size_t needed = (n + 255) / 256;
    if (blocks_b > needed) blocks_b = needed;
    hipLaunchKernelGGL(torus_march_kernel, dim3((unsigned)blocks_b), dim3(256), 0, stream, p, cols, ok, cursor, out, aux);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    return 0;
}

} // namespace S5NS
