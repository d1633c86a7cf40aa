// k_surface.hip -- where does a ray from infinity meet the photosphere of a geometrically thick disk?
//
// Batched counterpart of the reference's Python surface search (python/sim5diskraytrace.py:
// DiskRaytrace.geodesic :214-253 with flat=False and __find_surface :257-335): start far out on the
// incoming branch (geodesic_P_int at r0), walk down the geodesic with geodesic_follow in steps tied to the
// height above the surface, back up and refine when the surface is crossed, fall back to the equatorial
// crossing when the ray reaches the midplane, retry from further out when it escapes.  One lane follows one
// ray through the same sequence of SIM5 calls, decisions and constants as the Python code; the three
// nested levels (retry, walk, sub-steps of geodesic_follow) all carry hard caps.
//
// The disk surface H(R) is what the reference gets from the Python disk model's h(R).  Here it is a table
// (R_i ascending, H_i) staged once per workgroup into LDS and interpolated linearly, with a constant opening
// angle beyond the last point and H[0] below the first; every lane evaluates it many hundred times.
//
// The walk evaluates r(P) and mu(P) of ONE geodesic about a thousand times.  Both are Jacobi functions whose moduli
// are constants of the ray, so the AGM rungs of the two Landen ladders are climbed once per ray and kept in LDS
// (GeodTrack, s5_geod.hpp): a sub-step costs two ladder descents instead of two full sncndn evaluations
// (ref src/sim5kerr-geod.c:891-960, geodesic_follow only changes P).  Dynamic LDS = surface table + ladders.
#include "s5_disk.hpp"
#include "kernels.hpp"

namespace S5NS {

using namespace s5abi;

constexpr int SURF_MAX_TABLE = 4096;        // 2 x 32 KiB of LDS

// slope of the same piecewise-linear surface (what a disk model built on the table returns as dhdr) and
// the node values of another profile interpolated the same way
S5_DEV void surface_segment(const double* sR, int n, double R, int& lo, int& hi)
{
    lo = 0; hi = n - 1;                      // invariant sR[lo] < R <= sR[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (sR[mid] < R) lo = mid; else hi = mid;
    }
}

S5_DEV double surface_height(const double* sR, const double* sH, int n, double R)
{
    if (!(R > sR[0])) return sH[0];
    if (R >= sR[n - 1]) return sH[n - 1] * (R / sR[n - 1]);
    int lo = 0, hi = n - 1;                  // invariant sR[lo] < R <= sR[hi] ... bisection
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (sR[mid] < R) lo = mid; else hi = mid;
    }
    const double w = (R - sR[lo]) / (sR[hi] - sR[lo]);
    return sH[lo] + w * (sH[hi] - sH[lo]);
}

// the two ladders of a lane take 2 x 2 x LADDER_RUNGS_VALID x 8 B = 256 B of LDS: 64 KB per workgroup, two
// workgroups (8 waves) per CU next to a surface table of up to ~1 000 nodes
constexpr int SURF_BLOCK = 256;
constexpr size_t SURF_LADDER_BYTES = (size_t)2 * 2 * LADDER_RUNGS_VALID * SURF_BLOCK * sizeof(double);
#ifndef S5_SURF_WAVES
#define S5_SURF_WAVES 2
#endif

__global__ __launch_bounds__(SURF_BLOCK, S5_SURF_WAVES)
void disk_surface_kernel(SurfaceParams p, const double* __restrict__ tabR, const double* __restrict__ tabH,
                         const double* __restrict__ alpha, const double* __restrict__ beta,
                         double* __restrict__ outP, double* __restrict__ outR, double* __restrict__ outM,
                         double* __restrict__ outK, int* __restrict__ outStatus)
{
    extern __shared__ double lds[];
    double* sLad = lds;                                              // [2 ladders][2 * LADDER_RUNGS_VALID][SURF_BLOCK]
    double* sR = lds + SURF_LADDER_BYTES / sizeof(double);
    double* sH = sR + p.n_table;
    for (int i = threadIdx.x; i < p.n_table; i += SURF_BLOCK) { sR[i] = tabR[i]; sH[i] = tabH[i]; }
    __syncthreads();

    const size_t i = (size_t)blockIdx.x * SURF_BLOCK + threadIdx.x;
    if (i >= p.n) return;

    int status = 0;                          // 1 = surface point found, 0 = no intersection / error
    double P = NAN, r = 0.0, m = 0.0;
    double kout[4] = { NAN, NAN, NAN, NAN };

    Geod gd;
    GeodCache cache;
    int err = 0;
    const bool ok = init_inf(p.incl, p.sin_i, p.cos_i, p.a, alpha[i], beta[i], gd, err, cache);
    if (ok) {
        GeodTrack<SURF_BLOCK> trk;
        trk.build(gd, sLad + threadIdx.x);
        const double accuracy = 1e-2;                                            // ref py :268
        const double rbh = r_horizon(p.a);
        const double rbh_follow = 1.01 * r_horizon(gd.a);                        // geodesic_follow's own limit, ref c :913
        const double disk_theta = atan(surface_height(sR, sH, p.n_table, 1e6) / 1e6);   // :263
        const double alpha_beta = sqrt(gd.alpha * gd.alpha + gd.beta * gd.beta);
        const double cos_view = cos(gd.incl + disk_theta);

        // The three nested loops of the reference (retry from further out :258, walk :296, sub-steps of
        // geodesic_follow c :903) flattened into ONE loop whose body evaluates r(P), mu(P) exactly once: whatever
        // a lane is doing -- looking for its starting radius, stepping down the geodesic, backing up after a hit,
        // taking the equatorial crossing -- the wave shares the one inlined copy of the two ladder descents.  Per
        // ray the sequence of calls, operands and decisions is the reference's.
        enum : int { ST_GROW, ST_FOLLOW, ST_MID, ST_DONE };
        enum : int { FWD, BACK_FULL, BACK_HALF };
        int state = ST_GROW, purpose = FWD, iteration = 0, grow = 0, sub_it = 0;
        long walk_it = 0;
        bool found = false;
        double r0 = fmax(fmax(200.0, 1.1 * gd.rp), (0.5 + iteration) * alpha_beta / cos_view);        // :265
        double H1 = NAN, Hd = NAN, step = 0.0, fstep = 0.0, truestep = 0.0, step_factor = 1.0;

        // start of a walk step (:297-299): the step towards the surface, then geodesic_follow(step)
        auto begin_forward = [&]() {
            if (walk_it >= 2000000) { state = ST_DONE; return; }                 // neither found nor failed: failed (:331)
            ++walk_it;
            step = fmax(accuracy / 2., fmin((H1 - Hd) / 2., 0.5 * (sqrt(r) - 0.99) * step_factor));
            fstep = step; purpose = FWD; sub_it = 0; state = ST_FOLLOW;
        };

        // Two phases per round.  The slow one serves lanes that look for their starting radius (P_int: a Carlson
        // integral) or take the equatorial crossing (inverse cn with its special cases) -- once or twice per ray; the
        // hot one is the sub-step of geodesic_follow, ~300 times per ray, and carries none of that code or its
        // registers.  A lane that leaves the hot phase (retry from further out, equatorial crossing, done) idles
        // until the wave's hot phase ends.
        for (int round = 0; round < 64; ++round) {
            if (!wave_any(state != ST_DONE)) break;
            for (int guard = 0; guard < 4096; ++guard) {
                const bool slow = (state == ST_GROW) || (state == ST_MID);
                if (!wave_any(slow)) break;
                if (slow) {
                    if (state == ST_GROW) {                                      // :272-280
                        const double Pe = P_int(gd, r0, 0);
                        const double re = trk.rad(Pe), me = trk.pol(Pe);
                        const double R1 = re * sqrt(1. - me * me);
                        H1 = re * me;
                        Hd = surface_height(sR, sH, p.n_table, R1);
                        if ((Hd < H1) || (r0 > 5e6) || (grow + 1 >= 64)) {
                            if (!(Hd < H1)) state = ST_DONE;                      // :283 (Hd >= H1, or NaN): failed
                            else { P = Pe; r = re; m = me; step_factor = 1.0; walk_it = 0; begin_forward(); }
                        } else { r0 = 2.0 * r0; ++grow; }
                    } else {                                                     // :317-320
                        P = midplane_crossing(gd, 0, cache);
                        r = trk.rad(P);
                        m = trk.pol(P);
                        found = true; state = ST_DONE;
                    }
                }
            }
            for (long guard = 0; guard < 400000000L; ++guard) {
                if (!wave_any(state == ST_FOLLOW)) break;
                if (state == ST_FOLLOW) {
                    // one sub-step of geodesic_follow, c :904-924
                    truestep = mdiv(fstep, fabs(fstep)) * fmin(fabs(fstep), 5e-2 * msqrt(r));
                    P = P + mdiv(truestep, sq(r) + sq(gd.a * m));
                    r = trk.rad(P);
                    m = trk.pol(P);
                    int ended = 0, st = 1;
                    if (r < rbh_follow) { ended = 1; st = 0; }
                    else if ((P < 0.0) || (P > 2. * trk.Rpc)) { ended = 1; st = 0; }
                    else {
                        fstep -= truestep;
                        ++sub_it;
                        if (!(fabs(fstep) > 1e-5) || sub_it >= 100000) ended = 1;
                    }
                    if (ended) {
                        if (purpose == BACK_HALF) { found = true; state = ST_DONE; }       // :309-311
                        else if (purpose == BACK_FULL) { step_factor = step_factor / 5.; begin_forward(); }   // :312-314
                        else if (!st) state = ST_DONE;                                    // :301 failed
                        else {
                            const double R1 = r * sqrt(1. - m * m);
                            H1 = r * m;
                            Hd = surface_height(sR, sH, p.n_table, R1);
                            if (H1 <= Hd) {                                               // surface hit? :307
                                if (step < accuracy) { fstep = -step / 2.; purpose = BACK_HALF; }
                                else { fstep = -step; purpose = BACK_FULL; }
                                sub_it = 0;
                            }
                            else if (H1 < 1e-4) state = ST_MID;                           // equatorial plane hit? :316
                            else if (r < 1.05 * rbh) state = ST_DONE;                     // :324
                            else if (r > 1.1 * r0) {                                      // :325 retry from further out
                                ++iteration;
                                if (iteration > 3) state = ST_DONE;
                                else {
                                    r0 = fmax(fmax(200.0, 1.1 * trk.rp), (0.5 + iteration) * alpha_beta / cos_view);
                                    grow = 0; state = ST_GROW;
                                }
                            }
                            else if (m < 0.0) state = ST_DONE;                            // :326
                            else if (step < accuracy / 2.) state = ST_DONE;               // :327
                            else begin_forward();
                        }
                    }
                }
            }
        }
        if (found && !isnan(r) && !isnan(P)) {                                   // ref py :244-248
            status = 1;
            photon_momentum(p.a, r, m, gd.l, gd.q, gd.Rpc - P, 1.0, kout);       // :250
        } else {
            P = NAN; r = 0.0; m = 0.0;
        }
    }
    if (p.out_g) {
        // local frame of the disk surface at the point found: the reference's __tetrad / __gfactor /
        // __emission_angle (python/sim5diskraytrace.py:340-390) with the slope of the tabulated surface, the
        // Novikov-Thorne angular momentum and flux, and the tabulated radial velocity
        double gfac = NAN, mue = NAN, F = NAN;
        if (status == 1) {
            const double R = r * sqrt(1. - m * m);
            F = disk_flux(p.disk, R);
            double dhdr = 0.0, V = 0.0;
            const int nt = p.n_table;
            if (!(R > sR[0])) { dhdr = 0.0; V = p.tab_vr ? p.tab_vr[0] : 0.0; }
            else if (R >= sR[nt - 1]) { dhdr = sH[nt - 1] / sR[nt - 1]; V = p.tab_vr ? p.tab_vr[nt - 1] : 0.0; }
            else {
                int lo, hi;
                surface_segment(sR, nt, R, lo, hi);
                dhdr = (sH[hi] - sH[lo]) / (sR[hi] - sR[lo]);
                if (p.tab_vr) { const double w = (R - sR[lo]) / (sR[hi] - sR[lo]); V = p.tab_vr[lo] + w * (p.tab_vr[hi] - p.tab_vr[lo]); }
            }
            if (!(m > 0.0)) dhdr = 0.0;                                           // ref py :346
            Metric mt;
            kerr_metric(p.a, r, m, mt);
            Tetrad tt;
            tetrad_surface(mt, omega_from_ell(disk_ell(p.disk, R), mt), V, dhdr, tt);
            const double e0[4] = { 1.0, 0.0, 0.0, 0.0 }, e2[4] = { 0.0, 0.0, 1.0, 0.0 };
            double U[4], N[4];
            on2bl(e0, U, tt);
            on2bl(e2, N, tt);
            const double kU = dot(kout, U, mt);
            gfac = (kout[0] * mt.g00 + kout[3] * mt.g03) / kU;
            if (!(gfac > 0.0)) gfac = 0.0;                                        // ref py :360
            mue = dot(kout, N, mt) / kU;
            if ((mue < 0.0) && (mue > -1e-2)) mue = 1e-3;                         // ref py :387
        }
        p.out_g[i] = gfac; p.out_mue[i] = mue; p.out_flux[i] = F;
    }
    outP[i] = P; outR[i] = r; outM[i] = m; outStatus[i] = status;
    if (outK) { outK[4 * i] = kout[0]; outK[4 * i + 1] = kout[1]; outK[4 * i + 2] = kout[2]; outK[4 * i + 3] = kout[3]; }
}

} // namespace S5NS

#if S5_FAST
int s5_launch_disk_surface_fast(const s5abi::SurfaceParams& p, const double* tabR, const double* tabH,
#else
int s5_launch_disk_surface_strict(const s5abi::SurfaceParams& p, const double* tabR, const double* tabH,
#endif
                                  const double* alpha, const double* beta, double* P, double* r, double* m,
                                  double* k, int* status, hipStream_t stream)
{
    using namespace S5NS;
    const unsigned blocks = (unsigned)((p.n + SURF_BLOCK - 1) / SURF_BLOCK);
    const size_t lds = SURF_LADDER_BYTES + 2 * sizeof(double) * (size_t)p.n_table;
    // more than 64 KB of dynamic LDS has to be allowed once per device
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)disk_surface_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(SURF_LADDER_BYTES + 2 * sizeof(double) * (size_t)SURF_MAX_TABLE));
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(disk_surface_kernel, dim3(blocks), dim3(SURF_BLOCK), lds, stream, p, tabR, tabH, alpha, beta,
                       P, r, m, k, status);
    return (int)hipGetLastError();
}
