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
// (ref src/sim5kerr-geod.c:891-960, geodesic_follow only changes P).  In the fast variant most sub-steps do not even
// descend: consecutive points are close, so the two triples (sn, cn, dn) are advanced by the addition theorems
// (GeodTrack::Along) and the full evaluation re-anchors them every 48 sub-steps.
//
// FOUR kernels per job, the per-ray state between them in a workspace in HBM (the geodesic record and ~100 B of walk
// state per ray): a register allocation is the maximum over everything a kernel inlines, and the set-up (closed-form
// quartic, Carlson integrals, inverse-cn special cases) needs 256 VGPRs and spills, while the loop that does the work
// -- the sub-step of geodesic_follow, ~550 per ray -- needs half of that.
//   surface_setup_kernel   geodesic_init_inf, search for the starting radius                      (once)
//   surface_walk_kernel    the hot loop only: sub-steps until the ray leaves the walk            (4 rounds; a second
//                          instance with full-depth ladders for the few near-critical rays runs beside it)
//   surface_slow_kernel    equatorial crossing, retry from further out                            (4 rounds; rare)
//   surface_finish_kernel  photon momentum, local frame of the surface                            (once)
// A ray retries at most three times (ref py :258), so four walk/slow rounds finish every ray; later rounds find almost
// nothing to do and return at once.
#include <mutex>
// (this file keeps the explicit wave votes around skippable work, s5_math.hpp S5_ANY: as plain divergent branches the walk
// kernel of the fast variant needs 170 registers against its cap of 168 for three waves per SIMD, and spills)
#define S5_WAVE_VOTES 1
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

// The same value with the segment GUESSED first (round 6).  The segment of R -- the one pair of neighbours with sR[lo] < R <=
// sR[lo + 1]; unique, the table is strictly ascending -- is looked for where a table in equal steps would have it (index from
// (R - R_0) (n - 1) / (R_(n-1) - R_0)): two table reads in one round trip, and one more read for the neighbour on the side the
// comparison points to.  Where neither holds R (a table in unequal steps) the bisection above decides.  Whatever finds the
// segment, the interpolation is the same expression on the same nodes: the same bits.  Why: the bisection is eight DEPENDENT
// LDS round trips, and a walk step of some lane of a wave ends at almost every sub-step of the wave -- the walk kernel's
// waves waited on these reads (VALU busy 69 % at three waves per SIMD: profiles/r05_jobs_summary.json; the job 4.00 -> 3.80 ms
// in one call, profiles/r06_surface_guess_ab.txt).
S5_DEV double surface_height_guess(const double* sR, const double* sH, int n, double R, double R_first, double steps_per_R)
{
    if (!(R > R_first)) return sH[0];
    const double R_last = sR[n - 1];
    if (R >= R_last) return sH[n - 1] * (R / R_last);
    int g = (int)((R - R_first) * steps_per_R);
    g = g > n - 3 ? n - 3 : g; g = g < 1 ? 1 : g;                     // candidates g - 1, g, g + 1: nodes g - 1 .. g + 2
    int lo = g;
    double xl = sR[g], xh = sR[g + 1];
    if (!(xl < R)) { lo = g - 1; xh = xl; xl = sR[g - 1]; }
    else if (!(R <= xh)) { lo = g + 1; xl = xh; xh = sR[g + 2]; }
    if (n < 4 || !(xl < R && R <= xh)) {                              // a table in unequal steps (or too short to guess in): the bisection
        int hi;
        surface_segment(sR, n, R, lo, hi);
        xl = sR[lo]; xh = sR[hi];
    }
    const double w = (R - xl) / (xh - xl);
    const double hl = sH[lo], hh = sH[lo + 1];
    return hl + w * (hh - hl);
}

S5_DEV double surface_height_guess(const double* sR, const double* sH, int n, double R)
{
    const double R_first = sR[0];
    return surface_height_guess(sR, sH, n, R, R_first, (double)(n - 1) / (sR[n - 1] - R_first));
}

// The two ladders of a lane take 2 x 2 x rungs x 8 B of LDS.  The set-up kernel keeps all 8 rungs a double-precision
// modulus can need (64 KB per workgroup; it runs once).  The walk kernel keeps 6 -- enough unless a modulus is within
// 3e-6 of 1, i.e. the ray within ~1e-6 of the critical curve -- so that three workgroups (48 KB + table each) share a
// CU: 3 waves per SIMD at its 143 VGPRs.  The set-up marks the rays that need the deeper ladder (`deep`); those are
// walked by the slow kernel, which keeps all 8 (same values: a converged ladder is the same ladder).
constexpr int SURF_BLOCK = 256;
constexpr int WALK_RUNGS = 6;
constexpr size_t SURF_LADDER_BYTES = (size_t)2 * 2 * LADDER_RUNGS_VALID * SURF_BLOCK * sizeof(double);
constexpr size_t WALK_LADDER_BYTES = (size_t)2 * 2 * WALK_RUNGS * SURF_BLOCK * sizeof(double);

enum : int { ST_GROW, ST_FOLLOW, ST_MID, ST_DONE };
enum : int { FWD, BACK_FULL, BACK_HALF };

// walk state of one ray between the kernels
struct WalkState {
    double P, r, m, H1, Hd, step, fstep, step_factor, r0;
    long walk_it;
    int state, purpose, iteration, grow, sub_it, found, deep, pad;
};

// counters[]: how many rays the walk kernels of a round will find to do -- [0..3] rays with short ladders that walk in round r,
// [4..7] rays with deep ladders.  Counted (one atomic per wave) by the kernel that puts a ray into that state (set-up: round
// 0; slow steps of round r: round r + 1); a walk kernel reads ONE word and leaves at once when it is zero, instead of every
// workgroup reading the state of its 256 rays to find that out (rounds 2-4 are empty for almost every job: six full-grid
// launches with 64 KB of LDS each that swept the whole workspace).
constexpr int SURF_ROUNDS = 4;
enum : int { CNT_WALK = 0, CNT_DEEP = SURF_ROUNDS, CNT_SLOW = 2 * SURF_ROUNDS, N_SURF_COUNTERS = 3 * SURF_ROUNDS + 4 };
struct SurfaceWork {                 // the job's workspace (device): one record of each per ray
    Geod* gd;
    GeodCache* cache;
    WalkState* ws;
    unsigned* counters;
};

// count the lanes with `mine` into *counter: one atomic per wave
S5_DEV void count_lanes(unsigned* __restrict__ counter, bool mine)
{
    const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
    if (m && (__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)) == 0u) && mine)
        atomicAdd(counter, (unsigned)__builtin_popcountll(m));
}

S5_DEV double first_r0(const Geod& gd, int iteration, double alpha_beta, double cos_view)          // ref py :265
{
    return fmax(fmax(200.0, 1.1 * gd.rp), (0.5 + iteration) * alpha_beta / cos_view);
}

// start of a walk step (ref py :297-299): the step towards the surface, then geodesic_follow(step)
S5_DEV void begin_forward(WalkState& w)
{
    const double accuracy = 1e-2;
    if (w.walk_it >= 2000000) { w.state = ST_DONE; return; }                 // neither found nor failed: failed (:331)
    ++w.walk_it;
    w.step = fmax(accuracy / 2., fmin((w.H1 - w.Hd) / 2., 0.5 * (sqrt(w.r) - 0.99) * w.step_factor));
    w.fstep = w.step; w.purpose = FWD; w.sub_it = 0; w.state = ST_FOLLOW;
}

// one pass of the search for the starting radius (ref py :272-280) with r(P), mu(P) supplied by the caller
S5_DEV void grow_step(WalkState& w, double Pe, double re, double me, const double* sR, const double* sH, int n_table)
{
    const double R1 = re * sqrt(1. - me * me);
    w.H1 = re * me;
    w.Hd = surface_height_guess(sR, sH, n_table, R1);
    if ((w.Hd < w.H1) || (w.r0 > 5e6) || (w.grow + 1 >= 64)) {
        if (!(w.Hd < w.H1)) w.state = ST_DONE;                                // :283 (Hd >= H1, or NaN): failed
        else { w.P = Pe; w.r = re; w.m = me; w.step_factor = 1.0; w.walk_it = 0; begin_forward(w); }
    } else { w.r0 = 2.0 * w.r0; ++w.grow; }
}

// a wave-uniform double in scalar registers (its two halves through v_readfirstlane)
S5_DEV double scalar_copy(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// The radius a retry starts from (ref py :325 -> :265): max(200, 1.1 rp, (0.5 + iteration) sqrt(alpha^2 + beta^2) / cos(view)).
// A ray retries at most three times in a walk of thousands of sub-steps, and the walk sits at its register bound: the
// ray's own terms (rp, alpha, beta) are read from its record when the retry comes instead of being kept in registers
// through the walk; cos(view) is the same for every ray of a job and lives in scalar registers.
struct RetryRadius {
    const Geod* records;       // the job's geodesic records (ray i = blockIdx.x * SURF_BLOCK + threadIdx.x)
    double cos_view;           // wave-uniform
    S5_DEV double operator()(int iteration) const
    {
        const Geod* g = records + ((size_t)blockIdx.x * SURF_BLOCK + threadIdx.x);
        const double alpha = g->alpha, beta = g->beta, rp = g->rp;
        return fmax(fmax(200.0, 1.1 * rp), (0.5 + iteration) * sqrt(alpha * alpha + beta * beta) / cos_view);
    }
};

// The walk: sub-steps of geodesic_follow (ref c :903-924) and the decisions after each call (ref py :296-331) until
// the ray leaves it -- found, failed, equatorial crossing or retry from further out.  `ev` supplies r(P) and mu(P).
template <class Eval>
S5_DEV void follow_loop(WalkState& w, const Eval& ev, double a_in, double a_clamped, double twoRpc, const RetryRadius& retry_radius,
                        const double* sR, const double* sH, int n_table)
{
    // where a table in equal steps has the segment of R: first node and segments per unit of R (wave-uniform: scalar registers)
    const double R_first = scalar_copy(sR[0]);
    const double steps_per_R = scalar_copy((double)(n_table - 1) / (sR[n_table - 1] - sR[0]));
    const double accuracy = 1e-2;                                            // ref py :268
    const double rbh = r_horizon(a_in);
    const double rbh_follow = 1.01 * r_horizon(a_clamped);                   // geodesic_follow's own limit, ref c :913
#if S5_FAST
    // r, mu along the walk by the addition theorems (GeodTrack::Along), re-anchored by the full evaluation every
    // ANCHOR_EVERY sub-steps of the WAVE (a wave-uniform count, so that the lanes take the expensive branch together)
    // and whenever a lane's sub-step is too long for the series
#define S5_ANCHOR_EVERY 48
    constexpr int ANCHOR_EVERY = S5_ANCHOR_EVERY;
    typename Eval::Along along;
    int since_anchor = ANCHOR_EVERY;
#endif
    for (long guard = 0; guard < 400000000L; ++guard) {
        const bool walking = (w.state == ST_FOLLOW);
        if (!wave_any(walking)) break;
#if S5_FAST
        double truestep = 0.0, dP = 0.0;
        if (walking) {
            truestep = copysign(fmin(fabs(w.fstep), 5e-2 * msqrt(w.r)), w.fstep);      // step/|step| = +-1 exactly
            const double am = a_clamped * w.m;
            dP = mdiv(truestep, fma(w.r, w.r, am * am));
            w.P = w.P + dP;
        }
        const bool anchor_now = (since_anchor >= ANCHOR_EVERY) || wave_any(walking && !ev.step_is_small(dP));
        since_anchor = anchor_now ? 1 : since_anchor + 1;
#endif
        if (walking) {
            // one sub-step of geodesic_follow, c :904-924
#if S5_FAST
            if (anchor_now) ev.anchor(w.P, along, w.r, w.m);
            else ev.advance(dP, w.P, along, w.r, w.m);
#else
            const double truestep = mdiv(w.fstep, fabs(w.fstep)) * fmin(fabs(w.fstep), 5e-2 * msqrt(w.r));
            w.P = w.P + mdiv(truestep, sq(w.r) + sq(a_clamped * w.m));
            w.r = ev.rad(w.P);
            w.m = ev.pol(w.P);
#endif
            int ended = 0, st = 1;
            if (w.r < rbh_follow) { ended = 1; st = 0; }
            else if ((w.P < 0.0) || (w.P > twoRpc)) { ended = 1; st = 0; }
            else {
                w.fstep -= truestep;
                ++w.sub_it;
                if (!(fabs(w.fstep) > 1e-5) || w.sub_it >= 100000) ended = 1;
            }
            if (ended) {
                if (w.purpose == BACK_HALF) { w.found = 1; w.state = ST_DONE; }       // :309-311
                else if (w.purpose == BACK_FULL) { w.step_factor = w.step_factor / 5.; begin_forward(w); }   // :312-314
                else if (!st) w.state = ST_DONE;                                      // :301 failed
                else {
                    const double R1 = w.r * sqrt(1. - w.m * w.m);
                    w.H1 = w.r * w.m;
                    w.Hd = surface_height_guess(sR, sH, n_table, R1, R_first, steps_per_R);
                    if (w.H1 <= w.Hd) {                                               // surface hit? :307
                        if (w.step < accuracy) { w.fstep = -w.step / 2.; w.purpose = BACK_HALF; }
                        else { w.fstep = -w.step; w.purpose = BACK_FULL; }
                        w.sub_it = 0;
                    }
                    else if (w.H1 < 1e-4) w.state = ST_MID;                           // equatorial plane hit? :316
                    else if (w.r < 1.05 * rbh) w.state = ST_DONE;                     // :324
                    else if (w.r > 1.1 * w.r0) {                                      // :325 retry from further out
                        ++w.iteration;
                        if (w.iteration > 3) w.state = ST_DONE;
                        else { w.r0 = retry_radius(w.iteration); w.grow = 0; w.state = ST_GROW; }
                    }
                    else if (w.m < 0.0) w.state = ST_DONE;                            // :326
                    else if (w.step < accuracy / 2.) w.state = ST_DONE;               // :327
                    else begin_forward(w);
                }
            }
        }
    }
}

#define S5_SETUP_WAVES 1
__global__ __launch_bounds__(SURF_BLOCK, S5_SETUP_WAVES)
void surface_setup_kernel(SurfaceParams p, SurfaceWork wk, const double* __restrict__ tabR, const double* __restrict__ tabH,
                          const double* __restrict__ alpha, const double* __restrict__ beta)
{
    extern __shared__ double lds[];
    double* sLad = lds;                                              // [2 ladders][2 * LADDER_RUNGS_VALID][SURF_BLOCK]
    double* sR = lds + SURF_LADDER_BYTES / sizeof(double);
    double* sH = sR + p.n_table;
    for (int i = threadIdx.x; i < p.n_table; i += SURF_BLOCK) { sR[i] = tabR[i]; sH[i] = tabH[i]; }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * SURF_BLOCK + threadIdx.x;
    if (i >= p.n) return;

    Geod gd;
    GeodCache cache;
    int err = 0;
    WalkState w;
    w.P = NAN; w.r = 0.0; w.m = 0.0; w.H1 = NAN; w.Hd = NAN; w.step = 0.0; w.fstep = 0.0; w.step_factor = 1.0; w.r0 = 0.0;
    w.walk_it = 0; w.state = ST_DONE; w.purpose = FWD; w.iteration = 0; w.grow = 0; w.sub_it = 0; w.found = 0; w.deep = 0; w.pad = 0;
    const bool ok = init_inf(p.incl, p.sin_i, p.cos_i, p.a, alpha[i], beta[i], gd, err, cache);
    if (ok) {
        GeodTrack<SURF_BLOCK> trk;
        trk.build(gd, sLad + threadIdx.x);
        w.deep = (trk.st_r.top >= WALK_RUNGS || trk.st_m.top >= WALK_RUNGS) ? 1 : 0;
        const double disk_theta = atan(surface_height(sR, sH, p.n_table, 1e6) / 1e6);   // :263
        const double alpha_beta = sqrt(gd.alpha * gd.alpha + gd.beta * gd.beta);
        const double cos_view = cos(gd.incl + disk_theta);
        w.state = ST_GROW;
        w.r0 = first_r0(gd, 0, alpha_beta, cos_view);
        for (int guard = 0; guard < 64; ++guard) {
            if (!wave_any(w.state == ST_GROW)) break;
            if (w.state == ST_GROW) {
                const double Pe = P_int(gd, w.r0, 0);                        // :272
                grow_step(w, Pe, trk.rad(Pe), trk.pol(Pe), sR, sH, p.n_table);
            }
        }
        if (w.state == ST_GROW) w.state = ST_DONE;
    } else {
        gd.type = -1;
    }
    wk.gd[i] = gd; wk.cache[i] = cache; wk.ws[i] = w;
    count_lanes(&wk.counters[CNT_WALK + 0], w.state == ST_FOLLOW && !w.deep);
    count_lanes(&wk.counters[CNT_DEEP + 0], w.state == ST_FOLLOW && w.deep);
}

#define S5_SURF_WAVES 3
#define S5_SURF_WAVES_FAST_PLAIN 3
// NST = WALK_RUNGS, DEEP = false: the rays whose ladders fit (nearly all);  NST = LADDER_RUNGS_VALID, DEEP = true: the
// few that need the full ladders (64 KB of LDS per workgroup; launched beside the other on a second stream)
template <int NST, bool DEEP>
__global__ __launch_bounds__(SURF_BLOCK, DEEP ? 2 : (S5_FAST ? S5_SURF_WAVES_FAST_PLAIN : S5_SURF_WAVES))
void surface_walk_kernel(SurfaceParams p, SurfaceWork wk, const double* __restrict__ tabR, const double* __restrict__ tabH, const int round)
{
    if (wk.counters[(DEEP ? CNT_DEEP : CNT_WALK) + round] == 0u) return;      // no ray of this kind walks in this round
    extern __shared__ double lds[];
    double* sLad = lds;                                              // [2 ladders][2 * NST][SURF_BLOCK]
    double* sR = lds + (size_t)2 * 2 * NST * SURF_BLOCK;
    double* sH = sR + p.n_table;
    const size_t i = (size_t)blockIdx.x * SURF_BLOCK + threadIdx.x;
    const size_t ic = (i < p.n) ? i : 0;
    const bool mine = (i < p.n) && (wk.ws[ic].state == ST_FOLLOW) && ((wk.ws[ic].deep != 0) == DEEP);
    if (!__syncthreads_or(mine)) return;                             // nothing to walk in this workgroup
    for (int j = threadIdx.x; j < p.n_table; j += SURF_BLOCK) { sR[j] = tabR[j]; sH[j] = tabH[j]; }
    __syncthreads();
    if (!mine) return;

    WalkState w = wk.ws[i];
    GeodTrack<SURF_BLOCK, NST> trk;
    double a_clamped;
    RetryRadius retry_radius;
    {
        const Geod gd = wk.gd[i];
        trk.build(gd, sLad + threadIdx.x);
        // the clamped spin and the inclination are the job's, the same in every record: scalar registers
        a_clamped = scalar_copy(gd.a);
        const double disk_theta = atan(surface_height(sR, sH, p.n_table, 1e6) / 1e6);
        retry_radius.records = wk.gd;
        retry_radius.cos_view = scalar_copy(cos(gd.incl + disk_theta));
    }
    follow_loop(w, trk, p.a, a_clamped, 2. * trk.Rpc, retry_radius, sR, sH, p.n_table);
    wk.ws[i] = w;
    // (the rays this walk hands to the slow steps are NOT counted here: see surface_slow_kernel)
}

// the rare steps: equatorial crossing (ref py :317-320) and a new search for the starting radius after the ray
// escaped (:325 -> :265-280), through the generic per-ray routines (same values as the tracked ones, s5_geod.hpp)
__global__ __launch_bounds__(SURF_BLOCK)
void surface_slow_kernel(SurfaceParams p, SurfaceWork wk, const double* __restrict__ tabR, const double* __restrict__ tabH, const int round)
{
    // Only a walk leaves rays for the slow steps (the set-up and the slow steps themselves leave FOLLOW or DONE): when no ray
    // walked in this round -- one word each, counted by whoever put the rays into FOLLOW -- there is nothing to look for.
    // (Rounds 1-3 of a job without retries: 19 us each for reading the state of every ray.  The walk itself does not count
    // what it hands over: any use of the state after its loop changes the loop's register allocation -- a ballot or a second
    // store cost 40 moves per sub-step, 14 % of the kernel, measured in round 6.)
    if (wk.counters[CNT_WALK + round] == 0u && wk.counters[CNT_DEEP + round] == 0u) return;
    extern __shared__ double lds[];
    double* sR = lds;
    double* sH = sR + p.n_table;
    const size_t i = (size_t)blockIdx.x * SURF_BLOCK + threadIdx.x;
    const int st0 = (i < p.n) ? wk.ws[i].state : ST_DONE;
    const bool mine = (st0 == ST_GROW) || (st0 == ST_MID);
    if (!__syncthreads_or(mine)) return;
    for (int j = threadIdx.x; j < p.n_table; j += SURF_BLOCK) { sR[j] = tabR[j]; sH[j] = tabH[j]; }
    __syncthreads();
    if (!mine) return;
    WalkState w = wk.ws[i];
    const Geod gd = wk.gd[i];
    if (w.state == ST_MID) {
        const GeodCache cache = wk.cache[i];
        w.P = midplane_crossing(gd, 0, cache);
        w.r = position_rad(gd, w.P);
        w.m = position_pol(gd, w.P);
        w.found = 1; w.state = ST_DONE;
    } else {
        for (int guard = 0; guard < 64 && w.state == ST_GROW; ++guard) {
            const double Pe = P_int(gd, w.r0, 0);
            grow_step(w, Pe, position_rad(gd, Pe), position_pol(gd, Pe), sR, sH, p.n_table);
        }
        if (w.state == ST_GROW) w.state = ST_DONE;
    }
    wk.ws[i] = w;
    if (round + 1 < SURF_ROUNDS) {
        count_lanes(&wk.counters[CNT_WALK + round + 1], w.state == ST_FOLLOW && !w.deep);
        count_lanes(&wk.counters[CNT_DEEP + round + 1], w.state == ST_FOLLOW && w.deep);
    }
}

__global__ __launch_bounds__(SURF_BLOCK)
void surface_finish_kernel(SurfaceParams p, SurfaceWork wk, const double* __restrict__ tabR, const double* __restrict__ tabH,
                           double* __restrict__ outP, double* __restrict__ outR, double* __restrict__ outM,
                           double* __restrict__ outK, int* __restrict__ outStatus)
{
    extern __shared__ double lds[];
    double* sR = lds;
    double* sH = sR + p.n_table;
    if (p.out_g) {
        for (int j = threadIdx.x; j < p.n_table; j += SURF_BLOCK) { sR[j] = tabR[j]; sH[j] = tabH[j]; }
        __syncthreads();
    }
    const size_t i = (size_t)blockIdx.x * SURF_BLOCK + threadIdx.x;
    if (i >= p.n) return;
    const WalkState w = wk.ws[i];
    int status = 0;
    double P = w.P, r = w.r, m = w.m;
    double kout[4] = { NAN, NAN, NAN, NAN };
    if (w.found && !isnan(r) && !isnan(P)) {                                     // ref py :244-248
        const Geod gd = wk.gd[i];
        status = 1;
        photon_momentum(p.a, r, m, gd.l, gd.q, gd.Rpc - P, 1.0, kout);           // :250
    } else if (wk.gd[i].type == -1) {
        P = NAN; r = 0.0; m = 0.0;                                               // rejected by geodesic_init_inf
    } else {
        P = NAN; r = 0.0; m = 0.0;
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

namespace {
// workspace of the surface job: one grow-only device allocation per device (a job on another stream than the previous one
// first waits for that stream, so two jobs never share it)
struct SurfaceWorkspace {
    char* base = nullptr; size_t cap = 0; hipStream_t last = nullptr; bool used = false; bool attr_set = false;
    hipStream_t side = nullptr;              // the deep-ladder walk runs here, beside the main walk
    hipEvent_t fork = nullptr, join = nullptr;
};
SurfaceWorkspace g_surface_ws[64];
std::mutex g_surface_lock;
}

// give the grow-only workspaces of every device back (sim5gpu_release_workspaces): waits for the device that owns a block
#if S5_FAST
size_t s5_release_surface_workspace_fast()
#else
size_t s5_release_surface_workspace_strict()
#endif
{
    std::lock_guard<std::mutex> hold(g_surface_lock);
    int cur = 0;
    (void)hipGetDevice(&cur);
    size_t freed = 0;
    for (int d = 0; d < 64; ++d) {
        SurfaceWorkspace& W = g_surface_ws[d];
        if (!W.base) continue;
        (void)hipSetDevice(d);
        (void)hipDeviceSynchronize();
        (void)hipFree(W.base);
        freed += W.cap;
        W.base = nullptr; W.cap = 0; W.used = false; W.last = nullptr;
    }
    (void)hipSetDevice(cur);
    return freed;
}

#if S5_FAST
int s5_launch_disk_surface_fast(const s5abi::SurfaceParams& p, const double* tabR, const double* tabH,
#else
int s5_launch_disk_surface_strict(const s5abi::SurfaceParams& p, const double* tabR, const double* tabH,
#endif
                                  const double* alpha, const double* beta, double* P, double* r, double* m,
                                  double* k, int* status, hipStream_t stream)
{
    using namespace S5NS;
    SurfaceWorkspace* g_ws = g_surface_ws;
    std::lock_guard<std::mutex> hold(g_surface_lock);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    SurfaceWorkspace& W = g_ws[dev];
    const size_t n = p.n;
    const size_t b_gd = (sizeof(Geod) * n + 255) & ~size_t(255), b_ca = (sizeof(GeodCache) * n + 255) & ~size_t(255);
    const size_t b_ws = (sizeof(WalkState) * n + 255) & ~size_t(255);
    const size_t need = b_gd + b_ca + b_ws + 256;
    if (W.used && W.last != stream) { if ((e = hipStreamSynchronize(W.last)) != hipSuccess) return (int)e; }
    if (need > W.cap) {
        if (W.base) { if ((e = hipDeviceSynchronize()) != hipSuccess) return (int)e; (void)hipFree(W.base); W.base = nullptr; W.cap = 0; }
        if ((e = hipMalloc((void**)&W.base, need)) != hipSuccess) return (int)e;
        W.cap = need;
    }
    W.last = stream; W.used = true;
    SurfaceWork wk;
    wk.gd = (Geod*)W.base; wk.cache = (GeodCache*)(W.base + b_gd); wk.ws = (WalkState*)(W.base + b_gd + b_ca);
    wk.counters = (unsigned*)(W.base + b_gd + b_ca + b_ws);
    static_assert(N_SURF_COUNTERS * sizeof(unsigned) <= 256, "counter block");
    if ((e = hipMemsetAsync(wk.counters, 0, 256, stream)) != hipSuccess) return (int)e;

    const unsigned blocks = (unsigned)((n + SURF_BLOCK - 1) / SURF_BLOCK);
    const size_t tab_bytes = 2 * sizeof(double) * (size_t)p.n_table;
    const size_t lds_lad = SURF_LADDER_BYTES + tab_bytes, lds_walk = WALK_LADDER_BYTES + tab_bytes;
    auto walk = surface_walk_kernel<WALK_RUNGS, false>;
    auto walk_deep = surface_walk_kernel<LADDER_RUNGS_VALID, true>;
    if (!W.attr_set) {            // more than 64 KB of dynamic LDS has to be allowed once per device
        const int most = (int)(SURF_LADDER_BYTES + 2 * sizeof(double) * (size_t)SURF_MAX_TABLE);
        if ((e = hipFuncSetAttribute((const void*)surface_setup_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, most)) != hipSuccess) return (int)e;
        if ((e = hipFuncSetAttribute((const void*)walk, hipFuncAttributeMaxDynamicSharedMemorySize, most)) != hipSuccess) return (int)e;
        if ((e = hipFuncSetAttribute((const void*)walk_deep, hipFuncAttributeMaxDynamicSharedMemorySize, most)) != hipSuccess) return (int)e;
        if ((e = hipStreamCreateWithFlags(&W.side, hipStreamNonBlocking)) != hipSuccess) return (int)e;
        if ((e = hipEventCreateWithFlags(&W.fork, hipEventDisableTiming)) != hipSuccess) return (int)e;
        if ((e = hipEventCreateWithFlags(&W.join, hipEventDisableTiming)) != hipSuccess) return (int)e;
        W.attr_set = true;
    }
    hipLaunchKernelGGL(surface_setup_kernel, dim3(blocks), dim3(SURF_BLOCK), lds_lad, stream, p, wk, tabR, tabH, alpha, beta);
    for (int round = 0; round < SURF_ROUNDS; ++round) {             // a ray retries at most three times (ref py :258)
        // fork: the few rays with deep ladders walk on the side stream (a handful of waves, latency bound) while the
        // rest walk here; join before the slow steps
        if ((e = hipEventRecord(W.fork, stream)) != hipSuccess) return (int)e;
        if ((e = hipStreamWaitEvent(W.side, W.fork, 0)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL(walk_deep, dim3(blocks), dim3(SURF_BLOCK), lds_lad, W.side, p, wk, tabR, tabH, round);
        if ((e = hipEventRecord(W.join, W.side)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL(walk, dim3(blocks), dim3(SURF_BLOCK), lds_walk, stream, p, wk, tabR, tabH, round);
        if ((e = hipStreamWaitEvent(stream, W.join, 0)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL(surface_slow_kernel, dim3(blocks), dim3(SURF_BLOCK), tab_bytes, stream, p, wk, tabR, tabH, round);
    }
    hipLaunchKernelGGL(surface_finish_kernel, dim3(blocks), dim3(SURF_BLOCK), tab_bytes, stream, p, wk, tabR, tabH, P, r, m, k, status);
    return (int)hipGetLastError();
}
