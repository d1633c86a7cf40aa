// s5_thindisk.hpp -- one image-plane ray of the thin-disk problem, fused for the wave64 machine.
//
// Same algorithm as geodesic_init_inf -> geodesic_find_midplane_crossing -> geodesic_position_rad
// (s5_geod.hpp, i.e. ref: /root/reference/src/sim5kerr-geod.c:42-100, 846-885, 291-357).  Two routines follow the closed-form
// quartic of trace_thin_disk_impl:
//
//  thin_disk_finish_direct -- the reference's sequence.  In the strict variant every value is produced by the same
//    expression and the same R_F / sncndn routines as there (bit-identical results); in the fast variant the same sequence
//    with algebraically equal, cheaper forms (#if S5_FAST blocks).  It is the strict variant's only path; in the fast variant
//    it serves the few rays the routine below hands back (and whole images under SIM5GPU_IMG_DIRECT).  Arranged so that a
//    wave executes ONE copy of each expensive loop whatever mixture of geodesic classes its lanes hold:
//     * the three or four Carlson R_F evaluations a ray needs (radial integral to the turning point, K(mm), cn^-1 of the
//       observer's polar position, and for RC rays with a negative argument the second term of cn^-1) go through one loop
//       over "slots": per slot each lane selects its own arguments, the single inlined R_F body runs once for the whole
//       wave, and each lane scales its own result;
//     * r(P) needs sn (RR) or cn (RC) of different arguments: the lanes select (u, m) and ONE Landen ladder serves both;
//     * the special cases of the inverse Jacobi functions (m within 1e-8 of 0 or 1, z = 0, z = 1: asin, acos, log forms,
//       ref src/sim5elliptic.c:483-503) are flagged per lane and, if any lane of the wave has one, re-evaluated by the
//       generic routine out of line.
//  thin_disk_finish (fast variant only) -- r(P) from the addition theorem of the Jacobi functions instead of the radial
//    integral: see the comment at its head.
//
//  Both: a wave whose lanes all have the same class (the usual case: image neighbours) takes an instantiation with the class
//  as a compile-time constant (<.., KNOWN>); the Landen ladder keeps its rungs in LDS (thin_disk_ladder_column: 256-thread
//  one-dimensional workgroups only), climbed once per ray and descended once per crossing tried (and per ray of a mirrored pair).
//
// The per-ray state that callers need afterwards (polarization, tests) is returned in ThinRay.
#pragma once
#include "s5_disk.hpp"

namespace S5NS {

struct ThinRay {
    int    cls;        // s5abi::PX_*
    int    gtype;      // geodesic class or -1
    int    err;        // GD_* of init_inf
    double r, g, flux; // accepted crossing (r = NaN if none)
    double P;          // position integral of the accepted crossing
    // geodesic quantities (valid when err == 0)
    double a, l, q, beta, Tpp, Tip, rp;
    double dP;         // a number with the sign of Rpc - P at the accepted crossing (the radial direction there, ref :806);
                       // Rpc - P itself where the radial integral was evaluated
};

// generic routines out of line: taken only by lanes in a special case of the inverse functions
static __device__ __noinline__ double inv_sn_cold(double z, double m) { return inv_sn(z, m); }
static __device__ __noinline__ double inv_cn_cold(double z, double m) { return inv_cn(z, m); }
static __device__ __noinline__ double inv_tn_cold(double z, double m) { return inv_tn(z, m); }

// true if inv_sn(z, m) takes the plain z * R_F(1-z^2, 1-m z^2, 1) form
S5_DEV bool isn_plain(double m) { return !(fabs(m - 0.0) < 1e-8) && !(fabs(m - 1.0) < 1e-8); }
// true if inv_cn(z, m) takes the plain sqrt(1-z^2) R_F(z^2, 1-m(1-z^2), 1) [+ second term for z < 0] form
S5_DEV bool icn_plain(double z, double m)
{
    const bool snap = ((z > +1.0) && (z < +1.0 + 1e-8)) || ((z < -1.0) && (z > -1.0 - 1e-8)) ||
                      ((m > +1.0) && (m < +1.0 + 1e-8)) || ((m < 0.0) && (m > 0.0 - 1e-8));
    return !snap && !(z == 0.0) && !(z == 1.0) && !(m == 0.0) && !(m == 1.0);
}

// PRM: the type the job's parameters are read through -- s5abi::ImageParams (a kernel's by-value argument block: the
// compiler keeps what it reads in scalar registers) or a reference into the CONSTANT address space (s5abi::FastJob in the
// kernel-argument segment of the job-list kernel, k_disk_image.hip: every read is a scalar load where it is used, and
// param_reload() below makes the loads of a late phase start there instead of occupying registers from the first instruction)
template <bool WANT_STATE, int KNOWN, bool PAIR, class PRM>
S5_DEV void thin_disk_finish(const PRM& p, ThinRay& out, ThinRay& out2, const double a_in, const double a,
                             const double l, const double q, const double alpha, const double beta, int err, const int type_in,
                             const double ra, const double rb, const double rc_, const double rd_);
template <bool WANT_STATE, int KNOWN, bool PAIR, class PRM>
S5_DEV void thin_disk_finish_direct(const PRM& p, ThinRay& out, ThinRay& out2, const double a_in, const double a,
                                    const double l, const double q, const double alpha, const double beta, int err, const int type_in,
                                    const double ra, const double rb, const double rc_, const double rd_,
                                    const double beta_test0 = NAN, const double beta_test1 = NAN);

// (param_reload(): s5_config.hpp)
// internal class value: the fast routine leaves this ray to the direct one (never stored)
constexpr int PX_COLD_MARK = 100;
// internal error value of the fast routine: the ray's polar range tests are marginal, the direct routine decides (never stored)
constexpr int GD_E_LAST_BIT = 99;
// internal flux value (a flux is never negative): the table of the fast variant does not serve this hit -- within 2e-4 of the
// inner edge in x, beyond x = 16, or no table for this spin (s5_disk.hpp) -- and the closed form is owed.  It is evaluated
// ONCE, by thin_disk_owed_flux at the end of trace_thin_disk_impl, for both rays of a pair and whichever routine found the
// crossing: one copy of the four logarithms in a kernel instead of four (a fifth of its code, and the peak of its scalar
// register pressure: the coefficients of four inlined logarithm kernels hoisted into SGPR pairs).
constexpr double FLUX_OWED = -1.0;

template <bool PAIR, class PRM>
S5_DEV void thin_disk_owed_flux(const PRM& p, ThinRay& out, ThinRay& out2)
{
#if S5_FAST
    const bool f0 = (out.flux < 0.0), f1 = PAIR && (out2.flux < 0.0);
    S5_MARK("owed flux begin");
    if (S5_ANY(f0 || f1)) {
        // constants from the disk model's device block, not from the kernel arguments: referenced here they would sit in
        // ~30 SGPRs of every wave from the first instruction (the launchers always attach the block: capi_core.hip)
        const double* cold = param_reload(p).disk.cold;
#pragma unroll 1
        for (int member = 0; member < (PAIR ? 2 : 1); ++member) {
            const bool f = member ? f1 : f0;
            if (!S5_ANY(f)) continue;
            const double r = member ? out2.r : out.r;
            double F = 0.0;
            if (f) F = cold ? disk_flux_closed_form_mem(cold, r) : NAN;
            if (f) { if (member) out2.flux = F; else out.flux = F; }
        }
    }
#endif
}

// PAIR: the lane traces the ray (alpha, beta) into `out` AND its mirror image (alpha, -beta) into `out2`.  The two
// have the same constants of motion (l, and q through beta^2: ref :76-77), hence the same roots of R(r) and of the
// polar potential, the same three R_F integrals and the same Landen ladder; only the sign in front of cn^-1 in the
// position of the equatorial crossing (ref :868-871) and everything after it -- r(P), g, flux -- differ.  A lane's
// arithmetic for either ray is the arithmetic of the unpaired routine, value for value.
// P_FIRST: the caller is (inlined into) a kernel whose FIRST parameter is this `ImageParams p`, by value.  The direct routine
// behind the fast one then reads the parameters from the kernel's argument segment, where it runs, instead of out of the
// scalar registers that would have to carry them -- spilled to vector lanes -- through the whole fast path (measured on the
// pair kernel: SGPR spills 44 -> 28, -1.2 % time; the same pointer made in the kernel and handed down: +3 %).
// iy: the image row of (alpha, beta_in) when the caller made them with pixel_alpha / pixel_beta, -1 for a caller's own ray.
// Used by the fast variant's cold re-trace alone.  For a height that is not a power of two pixel_beta gives a row of the LOWER
// half minus its mirror row's value -- the reference's own value to the rounding of its quotient (iy + .5) / ny, and that last place decides the polar range
// tests of the rays for which they are marginal (the central column of an odd width: polar_tests_marginal).  So the direct
// routine is handed the reference's own beta of such a row -- beta_test0 for this ray, beta_test1 for the mirror ray of a pair,
// NaN where the ray's beta is the reference's already -- and forms the q of THOSE TESTS from it, per ray; everything else of
// the ray (and of the pair's shared geodesic) keeps beta_in, so paired and plain kernels still give the same bits.
template <class PRM> S5_DEV double pixel_beta_reference(const PRM& p, int iy);
template <bool WANT_STATE, bool PAIR, bool DIRECT = false, bool P_FIRST = false, class PRM>
S5_DEV void trace_thin_disk_impl(const PRM& p, double alpha, double beta_in, ThinRay& out, ThinRay& out2, const int iy = -1,
                                 const double beta_test0 = NAN, const double beta_test1 = NAN)
{
    S5_FPC_QUARTIC
    using namespace s5abi;
    out.cls = PX_ERROR; out.gtype = -1; out.err = GD_OK;
    out.r = NAN; out.g = 0.0; out.flux = 0.0; out.P = NAN;
    if (PAIR) {
        out2.cls = PX_ERROR; out2.gtype = -1; out2.err = GD_OK;
        out2.r = NAN; out2.g = 0.0; out2.flux = 0.0; out2.P = NAN;
    }

    // ---------------- constants of motion and range checks (ref :59-86) ----------------
    const double a_in = p.a;
    int err = GD_OK;
    if ((a_in < 0.0) || (a_in > 1. - 1e-6)) err = GD_E_SPIN;
    else if ((p.incl <= 0.0) || (p.incl >= 1.57079632679)) err = GD_E_INCL;
    const double beta = (beta_in == 0.0) ? +1e-6 : beta_in;
    const double a = fmax(1e-4, a_in);
    const double l = -alpha * p.sin_i;
    const double q = sq(beta) + sq(p.cos_i) * (sq(alpha) - sq(a_in));
    if (err == GD_OK && q == 0.0) err = GD_E_Q_RANGE;

    // ---------------- roots of R(r) (ref :986-1047) ----------------
    const double a2 = a * a, l2 = l * l;
    double A;
    const double C = sq(a - l) + q;
    const double D = 2. / 3. * (q + l2 - a2);
    const double E = 9. / 4. * sq(D) - 12. * a2 * q;
    const double F = -27. / 4. * (D * D * D) - 108. * a2 * q * D + 108. * sq(C);
    const double X = sq(F) - 4. * (E * E * E);
    if (X >= 0) {
        const double sX = msqrt(X);
        A = (F > sX ? +1 : -1) * 1. / 3. * mcbrt(fabs(F - sX) / 2.) +
            (F > -sX ? +1 : -1) * 1. / 3. * mcbrt(fabs(F + sX) / 2.);
    } else {
#if S5_FAST
        // Z^2 = (F^2 - X)/54^2 = 4 E^3/54^2, so Z^(1/3) = sqrt(E)/3: one square root instead of a square
        // root and a cube root; atan2 is scale-free, so the divisions by 54 drop out as well
        const double z = matan2(sqrt_pos(-X), F);                    // X < 0 here, and then E > 0
        A = sqrt_pos(E) * (2. / 3.) * mcos_third(z);
#else
        const double sX = S5_DIVC(msqrt(-X), 54.);
        const double F54 = S5_DIVC(F, 54.);
        const double Z = msqrt(sq(F54) + sq(sX));
        const double z = matan2(sX, F54);
        A = mcbrt(Z) * 2. * mcos(S5_DIVC(z, 3.));
#endif
    }
#if S5_FAST
    double B, rB;
    sqrt_rsqrt_pos(A + D, B, rB);
    const double CB = 4. * C * rB;
#else
    const double B = msqrt(A + D);
    const double CB = mdiv(4. * C, B);
#endif
    const double w_hi = -A + 2. * D - CB;
    const double w_lo = -A + 2. * D + CB;
    const bool hi_real = (w_hi >= 0.0), lo_real = (w_lo >= 0.0);
    const double h_hi = .5 * msqrt(fabs(w_hi)), h_lo = .5 * msqrt(fabs(w_lo));
    const double c_hi = +B / 2., c_lo = -B / 2.;

    // four scalars describe the roots; their meaning depends on the class:
    //   RR: r1 >= r2 >= r3 >= r4      RC: r1 >= r2 real, (u, v) = Re, Im of the complex pair
    //   CC: (b1, a1), (b2, a2) = Re, Im of the two pairs
    double ra, rb, rc_, rd_;
    int type;
    if (hi_real && lo_real) {
        const double p0 = c_hi + h_hi, p1 = c_hi - h_hi, p2 = c_lo + h_lo, p3 = c_lo - h_lo;
        const double t0 = fmin(p0, p2), t3 = fmax(p1, p3);
        ra = fmax(p0, p2); rb = fmax(t0, t3); rc_ = fmin(t0, t3); rd_ = fmin(p1, p3);
        type = T_RR;
        // r0 = DBL_MAX is above r1, so only the double-root test can fire (ref :1023-1036)
        if (err == GD_OK && (DBL_MAX < rc_ || (DBL_MAX > rb && DBL_MAX < ra))) err = GD_E_UNKNOWN;
        if (err == GD_OK && fabs(ra - rb) < 1e-8) { type = T_RR_DBL; err = GD_E_RR_DOUBLE; }
    } else if (hi_real) {
        ra = c_hi + h_hi; rb = c_hi - h_hi; rc_ = c_lo; rd_ = h_lo; type = T_RC;
    } else if (lo_real) {
        ra = c_lo + h_lo; rb = c_lo - h_lo; rc_ = c_hi; rd_ = h_hi; type = T_RC;
    } else {
        ra = c_hi; rb = h_hi; rc_ = c_lo; rd_ = h_lo; type = T_CC;      // (b1, a1, b2, a2)
    }

    // DIRECT: the reference's sequence for every lane (radial integral by R_F, comparisons with Rpc, special cases): the
    // strict variant's only path; in the fast variant the path of the few rays its own routine hands back
    if constexpr (DIRECT || !S5_RPC_ADD) {
        // one instantiation per uniform class (78 % of the rays of the headline image are RR, 22 % RC, and image
        // neighbours share the class), the generic one for mixed waves
        if constexpr (DIRECT) thin_disk_finish_direct<WANT_STATE, -1, PAIR>(p, out, out2, a_in, a, l, q, alpha, beta, err, type, ra, rb, rc_, rd_, beta_test0, beta_test1);
        else if (!wave_any(type != T_RR)) thin_disk_finish_direct<WANT_STATE, T_RR, PAIR>(p, out, out2, a_in, a, l, q, alpha, beta, err, type, ra, rb, rc_, rd_);
        else if (!wave_any(type != T_RC)) thin_disk_finish_direct<WANT_STATE, T_RC, PAIR>(p, out, out2, a_in, a, l, q, alpha, beta, err, type, ra, rb, rc_, rd_);
        else thin_disk_finish_direct<WANT_STATE, -1, PAIR>(p, out, out2, a_in, a, l, q, alpha, beta, err, type, ra, rb, rc_, rd_);
        if constexpr (!DIRECT) thin_disk_owed_flux<PAIR>(p, out, out2);      // (a DIRECT re-trace leaves it to the routine that called it)
        return;
    } else {
#if S5_RPC_ADD
    // (SIM5GPU_IMG_DIRECT, read here and nowhere else: the whole wave is left to the direct routine below)
    if (p.direct) { out.cls = PX_COLD_MARK; if (PAIR) out2.cls = PX_COLD_MARK; }
    else if (!wave_any(type != T_RR)) thin_disk_finish<WANT_STATE, T_RR, PAIR>(p, out, out2, a_in, a, l, q, alpha, beta, err, type, ra, rb, rc_, rd_);
    else if (!wave_any(type != T_RC)) thin_disk_finish<WANT_STATE, T_RC, PAIR>(p, out, out2, a_in, a, l, q, alpha, beta, err, type, ra, rb, rc_, rd_);
    else thin_disk_finish<WANT_STATE, -1, PAIR>(p, out, out2, a_in, a, l, q, alpha, beta, err, type, ra, rb, rc_, rd_);
    // The rays the fast routine left to the direct one: run HERE, inlined, from the pixel's coordinates -- the fast path's state
    // is dead by now, so the copy costs the hot path neither registers nor a call (an out-of-line copy called from inside the
    // routine cost the kernel 8 %: SGPRs saved and restored around the call sites, a stack reserved for every wave; inlined at
    // the routine's end with the roots kept alive for it: 6 %).  It redoes both rays of the lane's pair; the wave waits.
    const bool c0 = (out.cls == PX_COLD_MARK), c1 = PAIR && (out2.cls == PX_COLD_MARK);
    S5_MARK("cold retrace begin");
    if (S5_ANY(c0 || c1)) {
        if (c0 || c1) {
            ThinRay d0, d1;
            // the reference's own beta where pixel_beta's is not it bit for bit (see the routine's head): t0 for this ray (a row of
            // the lower half), t1 for the mirror ray of a pair; NaN = the ray's beta is the reference's
            double t0 = NAN, t1 = NAN;
            {
                const int ny = param_reload(p).ny;
                if (iy >= 0 && (ny & (ny - 1)) != 0) {
                    if (2 * iy + 1 > ny) { t0 = pixel_beta_reference(param_reload(p), iy); if (t0 == beta_in) t0 = NAN; }
                    if (PAIR && 2 * iy + 1 < ny) { t1 = pixel_beta_reference(param_reload(p), ny - 1 - iy); if (t1 == -beta_in) t1 = NAN; }
                }
            }
            if constexpr (P_FIRST) {
                const s5abi::ImageParams* pk = (const s5abi::ImageParams*)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(pk));                    // opaque: its reads are not merged with the kernel's own
                trace_thin_disk_impl<WANT_STATE, PAIR, true>(*pk, alpha, beta_in, d0, PAIR ? d1 : d0, -1, t0, t1);
            } else {
                trace_thin_disk_impl<WANT_STATE, PAIR, true>(param_reload(p), alpha, beta_in, d0, PAIR ? d1 : d0, -1, t0, t1);
            }
            // (the marks are read again here rather than kept: a per-lane flag alive across the routine is a scalar register pair)
            if (out.cls == PX_COLD_MARK) out = d0;
            if (PAIR && out2.cls == PX_COLD_MARK) out2 = d1;
        }
    }
    thin_disk_owed_flux<PAIR>(p, out, out2);
#endif
    }
}

template <bool WANT_STATE, bool P_FIRST = false, class PRM>
S5_DEV void trace_thin_disk(const PRM& p, double alpha, double beta_in, ThinRay& out, const int iy = -1)
{
    trace_thin_disk_impl<WANT_STATE, false, false, P_FIRST>(p, alpha, beta_in, out, out, iy);
}

// this lane's column of the workgroup's ladder block (256-thread one-dimensional workgroups: all callers)
S5_DEV double* thin_disk_ladder_column()
{
    __shared__ alignas(16) double s_ladder[(2 * LADDER_RUNGS + 1) * 256];      // (the extra row: a_{top+1} of ladder_descend_squares; 16 bytes: k_spectrum.hip re-uses the block for double2)
    return &s_ladder[threadIdx.x];
}

// Everything after the class of the ray is known.  KNOWN >= 0 instantiates the routine for a wave whose lanes
// all have that class: `type` is then a compile-time constant and the per-lane class selects, the slots and
// formulas of the other classes and their special cases are pruned by the compiler.  The arithmetic a lane
// performs is the same in every instantiation (same expressions, same order), so its result does not depend on
// which one its wave took -- images stay identical bit for bit whatever the tile shape.
template <bool WANT_STATE, int KNOWN, bool PAIR, class PRM>
S5_DEV void thin_disk_finish_direct(const PRM& p, ThinRay& out, ThinRay& out2, const double a_in, const double a,
                             const double l, const double q, const double alpha, const double beta, int err, const int type_in,
                             const double ra, const double rb, const double rc_, const double rd_, const double beta_test0, const double beta_test1)
{
    S5_FPC_FINISH
    using namespace s5abi;
    const int type = (KNOWN >= 0) ? KNOWN : type_in;
    const double a2 = a * a, l2 = l * l;
    double A = 0.0;                  // RC: |r1 - (u + i v)|, kept for r(P)
    // ---------------- per-class set-up of the radial integral (ref :1051-1100) ----------------
    // Rpc = pre * inverse-Jacobi(zR | mR); sqAB is reused by r(P)
    double mR, zR, pre = 0.0, sqAB, rp;
    if (type == T_RR || type == T_RR_DBL) {
#if S5_FAST
        // 1/sqAB also gives the denominator of the modulus: 1/((ra-rc)(rb-rd)) = (1/sqAB)^2
        sqrt_rsqrt_pos((ra - rc_) * (rb - rd_), sqAB, pre);
        mR = ((rb - rc_) * (ra - rd_)) * (pre * pre);
        pre = pre + pre;
#else
        mR = mdiv((rb - rc_) * (ra - rd_), (rb - rd_) * (ra - rc_));
        sqAB = msqrt((ra - rc_) * (rb - rd_));
        pre = mdiv(2., sqAB);
#endif
        zR = msqrt(mdiv(rb - rd_, ra - rd_));
        rp = ra;
    } else if (type == T_RC) {
        const double Aq = msqrt(sq(ra - rc_) + sq(rd_));
        const double Bq = msqrt(sq(rb - rc_) + sq(rd_));
#if S5_FAST
        sqrt_rsqrt_pos(Aq * Bq, sqAB, pre);
        mR = (sq(Aq + Bq) - sq(ra - rb)) * (0.25 * (pre * pre));       // 1/(A B) = (1/sqrt(A B))^2
#else
        mR = mdiv(sq(Aq + Bq) - sq(ra - rb), 4. * Aq * Bq);
        sqAB = msqrt(Aq * Bq);
        pre = mdiv(1., sqAB);
#endif
        zR = mdiv(Aq - Bq, Aq + Bq);
        rp = ra;
        // keep A, B for r(P): the RC formula needs them again
        A = Aq;                       // kept for r(P); B is recomputed there from the same expression
    } else {
        const double b1 = ra, a1 = rb, b2 = rc_, a2c = rd_;
        const double Aq = msqrt(sq(b1 - b2) + sq(a1 + a2c));
        const double Bq = msqrt(sq(b1 - b2) + sq(a1 - a2c));
        const double g1 = msqrt(mdiv(4. * sq(a1) - sq(Aq - Bq), sq(Aq + Bq) - 4. * sq(a1)));
        mR = mdiv(4. * Aq * Bq, sq(Aq + Bq));
        sqAB = 0.0;
        pre = mdiv(2., Aq + Bq);
        zR = mdiv(-1., g1);
        rp = b1 - a1 * g1;
    }

    // ---------------- roots of the polar potential (ref :1110-1184, host branch: s5_geod.hpp polar_m2_host_rounding) --------
#if S5_FAST
    const double qla = q + l2 - a2;
    const double XT = msqrt(sq(qla) + 4. * q * a2) + qla;
    double m2m = XT * p.inv_2a2;
    double m2p = mdiv(q + q, XT);
    double s_m2p, rs_m2p;                                   // sqrt(m2p) and its reciprocal, used three times
    sqrt_rsqrt_pos(m2p, s_m2p, rs_m2p);
    // the rays whose range tests below are decided by the last bit of m2p (central column: l = 0, m2p = 1 in real arithmetic;
    // central row: beta = 0 -> 1e-6) get the reference's own roundings: its x87 sequence and an IEEE square root
    // (a ray whose reference beta is not the beta it was given bit for bit -- beta_test0 / beta_test1, heights that are not a
    // power of two -- has its OWN q in those tests: err_m[]; where the given beta's roots fail them and a ray's own pass, the
    // shared polar roots are the passing ray's, an ulp away)
    int err_m[2] = {-1, -1};                                // >= 0: this member's own verdict of the polar range tests
    if (S5_ANY(polar_tests_marginal(m2p, s_m2p, p.cos_i, l2, qla, XT))) {
        if (polar_tests_marginal(m2p, s_m2p, p.cos_i, l2, qla, XT)) {
            polar_m2_host_rounding(q, l2, a2, m2m, m2p);
            s_m2p = sqrt(m2p);
            const bool own0 = (beta_test0 == beta_test0), own1 = PAIR && (beta_test1 == beta_test1);
            if ((own0 || own1) && err == GD_OK) {
                const int e_sh = polar_range_error(q, a2, m2m, m2p, s_m2p, p.cos_i);
                err_m[0] = err_m[1] = e_sh;
                double mm_own = m2m, mp_own = m2p;
#pragma unroll
                for (int member = 0; member < (PAIR ? 2 : 1); ++member) {
                    if (!(member ? own1 : own0)) continue;
                    const double q_own = constant_q(member ? beta_test1 : beta_test0, p.cos_i, alpha, a_in);
                    double mm2, mp2;
                    polar_m2_host_rounding(q_own, l2, a2, mm2, mp2);
                    err_m[member] = polar_range_error(q_own, a2, mm2, mp2, sqrt(mp2), p.cos_i);
                    if (err_m[member] == GD_OK) { mm_own = mm2; mp_own = mp2; }
                }
                if (e_sh != GD_OK) { m2m = mm_own; m2p = mp_own; s_m2p = sqrt(m2p); }
            }
            rs_m2p = 1.0 / s_m2p;
        }
    }
#else
    double m2m, m2p;
    polar_m2_host_rounding(q, l2, a2, m2m, m2p);
    const double s_m2p = msqrt(m2p);
#endif
    double mmT = 0.0, mK = 0.0;
    if (err == GD_OK) {
        if ((m2p <= 0.0) || (m2p >= 1.0)) err = GD_E_MUPLUS;
        else if (q > 0.0) {
#if S5_FAST
            // mK = 1/sqrt(a^2 (m2p + m2m)) first; the modulus m2p/(m2p + m2m) is then m2p a^2 mK^2
            const double rk = rsqrt_pos(a2 * (m2p + m2m));
            mmT = (m2p * a2) * (rk * rk);
            if ((mmT < 0.0) || (mmT >= 1.0)) err = GD_E_MM;
            else if (fabs(p.cos_i) > s_m2p) err = GD_E_MU0;
            else mK = rk;
#else
            mmT = mdiv(m2p, m2p + m2m);
            if ((mmT < 0.0) || (mmT >= 1.0)) err = GD_E_MM;
            else if (fabs(p.cos_i) > s_m2p) err = GD_E_MU0;
            else mK = mdiv(1., msqrt(a2 * (m2p + m2m)));
#endif
        } else if (q < 0.0) {
            mmT = mdiv(m2p + m2m, m2p);
            if ((mmT < 0.0) || (mmT >= 1.0)) err = GD_E_MM;
            else if ((fabs(p.cos_i) > s_m2p) || (fabs(p.cos_i) < msqrt(-m2m))) err = GD_E_MU0;
            else mK = mdiv(1., msqrt(a2 * m2p));
        } else {
            err = GD_E_Q_RANGE;
        }
    }
    out.err = err;
    const bool ok = (err == GD_OK);
#if S5_FAST
    const bool ok_m0 = ok && !(err_m[0] > GD_OK), ok_m1 = ok && !(err_m[1] > GD_OK);       // (a member failing its own range tests)
    const double u_i = p.cos_i * rs_m2p;
#else
    const double u_i = mdiv(p.cos_i, s_m2p);
#endif

    // ---------------- the R_F evaluations of this ray, one shared loop (see header) ----------------
    // slot 0: radial integral   slot 1: K(mmT)   slot 2: cn^-1(u_i | mmT)   slot 3: RC, zR < 0: second term
    double zT = zR;
    if (wave_any(type == T_CC)) {                                                     // tn^-1 -> sn^-1 (ref :527)
        if (type == T_CC) zT = msqrt(mdiv(zR * zR, 1. + zR * zR));
    }
    const bool plain0 = (type == T_RC) ? icn_plain(zR, mR)
                      : (type == T_CC) ? (!(mR == 0.0) && !(mR == 1.0) && isn_plain(mR))
                                       : isn_plain(mR);
    const bool plain2 = icn_plain(u_i, mmT);
    const bool need3 = ok && (type == T_RC) && plain0 && !(zR > 0.0);
    double res0 = 0.0, res1 = 0.0, res2 = 0.0, res3 = 0.0;
    // unrolled: three inlined R_F bodies (slot 1 is the table, slot 3 rare).  Rolled into one body it once saved the kernel
    // from 256 VGPRs and spills; at today's 108 VGPRs the copies cost nothing and the rolled loop costs 4 % (measured)
#pragma unroll
    for (int slot = 0; slot < 4; ++slot) {
#if S5_F_AGMK
        if (slot == 1) continue;                         // K(mmT) comes from the AGM below
#endif
        if (slot == 3 && !wave_any(need3)) break;
        double x, y, mult;
        if (slot == 0) {
            const double z2 = zT * zT;
            if (type == T_RC) { x = z2; y = 1.0 - mR * (1. - z2); mult = sqrt_pos(1. - z2); }     // plain lanes: z2 < 1
            else { x = 1.0 - z2; y = 1.0 - mR * zT * zT; mult = zT; }      // (m z) z, as ref :485
        } else if (slot == 1) {
            x = 0.0; y = 1.0 - mmT; mult = 1.0;
        } else if (slot == 2) {
            const double z2 = u_i * u_i;
            x = z2; y = 1.0 - mmT * (1. - z2); mult = sqrt_pos(1. - z2);
        } else {
            double m3 = mdiv(mR, mR - 1.);              // modulus of the second term (ref :513, :280)
            if (m3 == 1.0) m3 = 0.99999999;
            const double s = -zR, s2 = s * s;
            x = 1. - s2; y = 1.0 - s2 * m3; mult = s;
        }
        // plain lanes have x, y > 0 by construction (z^2 < 1, moduli in [0,1)); the others are redone out of line
        const double v = mult * carlson_rf_positive(x, y, 1.0);
        if (slot == 0) res0 = v; else if (slot == 1) res1 = v; else if (slot == 2) res2 = v; else res3 = v;
    }
    // assemble as the generic routines do
    double Rint = res0;
    if (type == T_RC && need3) Rint = mdiv(2., msqrt(1. - mR)) * res3 + res0;
#if S5_F_AGMK
    {
        // K(mm): from the table (kernels.hpp KT_*: 128 polynomials of degree 7 on [0, 0.9], 2e-16) where it reaches,
        // by the arithmetic-geometric mean elsewhere (the lanes' wave with them)
        const bool tab = (p.ktab != nullptr) && (mmT >= 0.0) && (mmT < KT_MMAX);
        if (tab) {
            const double u = mmT * ((double)KT_N / KT_MMAX);
            int i = (int)u;
            i = i < KT_N - 1 ? i : KT_N - 1;
            const double tau = 2.0 * (u - (double)i) - 1.0;
            const double* c = p.ktab + (size_t)i * (KT_DEG + 1);
            double acc = c[KT_DEG];
#pragma unroll
            for (int k = KT_DEG - 1; k >= 0; --k) acc = __builtin_fma(acc, tau, c[k]);
            res1 = acc;
        }
        if (wave_any(!tab)) {
            if (!tab) res1 = ell_K(mmT);
        }
    }
#endif
    double K = res1;
    double icn_i = res2;
    // special cases, out of line
    if (wave_any(ok && !plain0)) {
        if (ok && !plain0)
            Rint = (type == T_RC) ? inv_cn_cold(zR, mR) : (type == T_CC) ? inv_tn_cold(zR, mR) : inv_sn_cold(zR, mR);
    }
    if (wave_any(ok && !plain2)) {
        if (ok && !plain2) icn_i = inv_cn_cold(u_i, mmT);
    }
    const double Rpc = pre * Rint;

    if (WANT_STATE) {
        out.a = a; out.l = l; out.q = q; out.beta = beta; out.rp = rp; out.dP = NAN;
        out.Tpp = 2. * (mK * K); out.Tip = mK * icn_i;
        if (PAIR) {
            out2.a = a; out2.l = l; out2.q = q; out2.beta = -beta; out2.rp = rp; out2.dP = NAN;
            out2.Tpp = out.Tpp; out2.Tip = out.Tip;
        }
    }
    if (PAIR) out2.err = err;
    if (!ok) return;
    out.gtype = type;
    out.cls = PX_MISS;
    if (PAIR) { out2.gtype = type; out2.cls = PX_MISS; }
#if S5_FAST
    if (!ok_m0) { out.err = err_m[0]; out.gtype = -1; out.cls = PX_ERROR; }
    if (PAIR && !ok_m1) { out2.err = err_m[1]; out2.gtype = -1; out2.cls = PX_ERROR; }
#endif

    // ---------------- equatorial crossings and r(P) (ref :846-885, :291-357) ----------------
    const bool q_pos = (q > 0.0);
    double uu = u_i;
    bool u_bad = (uu < -1.0 - 1e-4) || (uu > +1.0 + 1e-4);
    if (uu < -1.0) uu = -1.0;
    if (uu > +1.0) uu = +1.0;
    double icn_u = icn_i;
    if (wave_any(uu != u_i && !u_bad && q_pos)) {              // clamped by the slack rule: re-evaluate
        if (uu != u_i && !u_bad && q_pos) icn_u = inv_cn_cold(uu, mmT);
    }
    // r(P) needs sn (RR) or cn (RC) of modulus mR: the rungs of its Landen ladder are climbed ONCE per ray -- they serve
    // every crossing order and both rays of a pair -- and kept in LDS; lanes that cannot use them climb a short dummy
    LadderLds lad{thin_disk_ladder_column()};
    LadderState lst;
    const bool ladder_class = (type == T_RR) || (type == T_RC);
    const bool may_cross = q_pos && !u_bad;
    if (wave_any(ladder_class && may_cross)) ladder_climb(lad, (ladder_class && may_cross) ? mR : 0.5, lst);
    bool cf0 = false, cf1 = false;
    // two inlined passes rather than a run-time loop: as a loop the compiler predicates the pass on per-lane state and the
    // lanes used fall from 97 % to 91 % (measured: +6.5 % VALU instructions, +4.5 % time)
#pragma unroll
    for (int member = 0; member < (PAIR ? 2 : 1); ++member) {
        const double beta_m = (member == 0) ? beta : -beta;
        int cls_m = PX_MISS;
        double r_m = NAN, P_m = NAN, g_m = 0.0, flux_m = 0.0, dP_m = NAN;
        bool cf_m = false;                      // fast variant: the flux of this ray is owed by the closed form (below)
#if S5_FAST
        if (!((member == 0) ? ok_m0 : ok_m1)) continue;              // this member failed its own range tests: PX_ERROR stands
#endif
        bool done = false;
#pragma unroll 1
        for (int order = 0; order < p.max_order; ++order) {
            if (!wave_any(!done)) break;
            if (!done) {
                double P;
                if (!may_cross) P = NAN;
                else {
                    if (beta_m > 0.0) P = mK * ((2. * (double)order + 1.) * K + icn_u);
                    else if (beta_m < 0.0) P = mK * ((2. * (double)order + 1.) * K - icn_u);
                    else P = mK * ((2. * (double)order + 1.) * K);
                    if (P > 2. * Rpc) P = NAN;
                }
                if (isnan(P)) { cls_m = (order == 0) ? PX_NAN0 : PX_NAN1; done = true; }
                else {
                    // r(P): RR through sn, RC through cn, one ladder for both
                    double r;
                    const bool in_range = !((P <= 0.0) || (P >= 2. * Rpc));
                    const bool at_peri = (P == Rpc);
                    const bool rr = (type == T_RR), rcx = (type == T_RC) && !(P > Rpc);
                    const bool use_ladder = in_range && !at_peri && (rr || rcx);
                    double su = 0.0;
                    if (rr) su = 0.5 * fabs(P - Rpc) * sqAB;
                    else if (rcx) su = sqAB * (Rpc - P);
                    double sn = 0.0, cn = 1.0, dn = 1.0;
                    if (wave_any(use_ladder)) {
                        if (use_ladder) ladder_descend(lad, lst, su, sn, cn, dn);
                    }
                    if (!in_range) r = NAN;
                    else if (at_peri) r = rp;
                    else if (rr) {
                        S5_FPC_RADIUS
                        const double sn2 = sn * sn;
                        r = mdiv(ra * (rb - rd_) - rb * (ra - rd_) * sn2, rb - rd_ - (ra - rd_) * sn2);
                    } else if (rcx) {
                        S5_FPC_RADIUS
                        const double Aq = A;
                        const double Bq = msqrt(sq(rb - rc_) + sq(rd_));
                        r = mdiv(rb * Aq - ra * Bq - (rb * Aq + ra * Bq) * cn, (Aq - Bq) - (Aq + Bq) * cn);
                    } else r = NAN;
                    if (r >= p.rms) {
                        cls_m = (order == 0) ? PX_HIT0 : PX_HIT1;
                        r_m = r; P_m = P; dP_m = Rpc - P;
                        done = true;
                    }
                }
            }
        }
        if (cls_m == PX_HIT0 || cls_m == PX_HIT1) {
#if S5_FAST
            double x, rx;                                 // sqrt(r) and its reciprocal serve the g-factor and the flux
            sqrt_rsqrt_pos(r_m, x, rx);                   // r >= rms > 0
            g_m = gfactor_kepler_x(r_m, x, a_in, l);
            flux_m = disk_flux_table(p.disk, r_m, x, rx, cf_m);
#else
            g_m = gfactor_kepler(r_m, a_in, l);
            flux_m = disk_flux(p.disk, r_m);
#endif
        }
        (void)cf0; (void)cf1;
        if (cf_m) flux_m = FLUX_OWED;           // the closed form, once, at the end of trace_thin_disk_impl (thin_disk_owed_flux)
        if (!PAIR || member == 0) { out.cls = cls_m; out.r = r_m; out.P = P_m; out.g = g_m; out.flux = flux_m; if (WANT_STATE) out.dP = dP_m; }
        else { out2.cls = cls_m; out2.r = r_m; out2.P = P_m; out2.g = g_m; out2.flux = flux_m; if (WANT_STATE) out2.dP = dP_m; }
    }
}


#if S5_RPC_ADD
// ---------------------------------------------------------------------------------------------------------------------------
// FAST VARIANT.  Same sequence, with two differences in how r(P) is reached (everything else is the arithmetic of the
// routine above, #if S5_FAST branches taken):
//
//  * THE RADIAL INTEGRAL Rpc IS NOT EVALUATED.  r(P) takes sn or cn of c (Rpc - P) = F0 - w, where w = c P and F0 = c Rpc is
//    an inverse Jacobi function of an ALGEBRAIC argument: sn(F0) = zR (RR), cn(F0) = zR (RC) -- the very argument the R_F
//    of the radial integral would be called with.  So sn, cn, dn of w come from the ladder, those of F0 from zR, and the
//    addition theorem gives sn(w - F0) or cn(F0 - w): no inverse function, one R_F (of three) less per ray pair.  The
//    comparisons of P with Rpc and 2 Rpc (ref :303-309, :336, :881) become sign tests on the same quantities; they need
//    w < 2 K(mR) (RR) or 4 K(mR) (RC) to be unambiguous, and K(mR) = pi / (2 c_N) is the last mean of the ladder that is
//    climbed anyway -- as a bound on the angle w c_N the descent starts from.  The descent stays in the fraction form
//    (ladder_descend_fractions): sn^2 = Pn / Q and cn = X / Y enter the r(P) formulas without having been divided.
//  * Rays the addition theorem does not serve are MARKED (cls = PX_COLD_MARK) and traced by thin_disk_finish_direct, which the
//    caller (trace_thin_disk_impl) runs after this routine from the pixel's coordinates: geodesics with complex roots only
//    (CC: r(P) is NaN, but the class of the pixel depends on P > 2 Rpc), the special cases of the inverse functions (modulus
//    within 1e-8 of 0 or 1, ...), and ILL-CONDITIONED sums -- the denominator 1 - m sn^2(w) sn^2(F0) of the theorem cancels
//    when m, sn(w) and sn(F0) are all close to 1 (rays that wind around the photon orbit of a fast hole), so below 1e-3 of
//    its terms the ray is handed to the direct evaluation.
// A ray's result still depends on its own arguments only (which path a lane takes is decided by its own values).
// ---------------------------------------------------------------------------------------------------------------------------
enum : int { CROSS_FORMULA = 0, CROSS_BEYOND = 1, CROSS_NONE = 2 };

template <bool WANT_STATE, int KNOWN, bool PAIR, class PRM>
S5_DEV void thin_disk_finish(const PRM& p_in, ThinRay& out, ThinRay& out2, const double a_in, const double a,
                             const double l, const double q, const double alpha, const double beta, int err, const int type_in,
                             const double ra, const double rb, const double rc_, const double rd_)
{
    S5_FPC_FINISH
    // constant-address-space parameters are loaded from HERE: each class instantiation issues its own scalar loads of the few
    // it reads instead of holding them -- spilled to vector lanes -- from the kernel's first instruction
    const auto& p = param_reload(p_in);
    using namespace s5abi;
    constexpr double S5_PI = 3.14159265358979323846;
    const int type = (KNOWN >= 0) ? KNOWN : type_in;
    const double a2 = a * a, l2 = l * l;
    // ---------------- per-class set-up of the radial motion (ref :1051-1100) ----------------
    // modulus mR and argument zR of the inverse function in Rpc = pre * inverse-Jacobi(zR | mR); sqAB scales P in r(P)
    double mR, zR, sqAB;              // sqAB: CC keeps the prefactor of Rpc here (it has no r(P))
    double Aq = 0.0, Bq = 0.0;       // RC: |r1 - (u + i v)|, |r2 - (u + i v)|, kept for r(P);  CC: Aq holds the pericentre rp
                                     // (RR and RC: rp = r1; one register pair for the class-specific constant)
    if (type == T_RR || type == T_RR_DBL) {
        // 1/sqAB also gives the denominator of the modulus: 1/((ra-rc)(rb-rd)) = (1/sqAB)^2
        double pre;
        sqrt_rsqrt_pos((ra - rc_) * (rb - rd_), sqAB, pre);
        mR = ((rb - rc_) * (ra - rd_)) * (pre * pre);
        zR = msqrt(mdiv(rb - rd_, ra - rd_));
    } else if (type == T_RC) {
        Aq = msqrt(sq(ra - rc_) + sq(rd_));
        Bq = msqrt(sq(rb - rc_) + sq(rd_));
        double pre;
        sqrt_rsqrt_pos(Aq * Bq, sqAB, pre);
        mR = (sq(Aq + Bq) - sq(ra - rb)) * (0.25 * (pre * pre));       // 1/(A B) = (1/sqrt(A B))^2
        zR = mdiv(Aq - Bq, Aq + Bq);
    } else {
        const double b1 = ra, a1 = rb, b2 = rc_, a2c = rd_;
        const double Ac = msqrt(sq(b1 - b2) + sq(a1 + a2c));
        const double Bc = msqrt(sq(b1 - b2) + sq(a1 - a2c));
        const double g1 = msqrt(mdiv(4. * sq(a1) - sq(Ac - Bc), sq(Ac + Bc) - 4. * sq(a1)));
        mR = mdiv(4. * Ac * Bc, sq(Ac + Bc));
        sqAB = mdiv(2., Ac + Bc);    // the prefactor of Rpc
        zR = mdiv(-1., g1);
        Aq = b1 - a1 * g1;           // rp
    }
#define S5_THIN_RP ((type == T_CC) ? Aq : ra)

    // ---------------- roots of the polar potential (ref :1110-1184, device branch) ----------------
    const double qla = q + l2 - a2;
    const double XT = msqrt(sq(qla) + 4. * q * a2) + qla;
    const double m2m = XT * p.inv_2a2;
    const double m2p = mdiv(q + q, XT);
    double s_m2p, rs_m2p;                                   // sqrt(m2p) and its reciprocal, used three times
    sqrt_rsqrt_pos(m2p, s_m2p, rs_m2p);
    // The two range tests on m2p (ref :1140, :1153) can only fail by rounding: in real arithmetic m2p <= 1 (equal for l = 0:
    // the central column of an odd width) and |cos i| <= sqrt(m2p) (equal for beta = 0: the observer on the polar turning
    // point; the central row of an odd height, where the reference sets beta = 1e-6).  So whoever comes within 1e-12 of
    // either threshold is left to the direct routine, which forms m2p with the reference's own roundings (s5_geod.hpp
    // polar_m2_host_rounding) -- as an error code of its own: the lane idles through this routine like any rejected ray, both
    // rays of a pair (they share the polar roots).  One subtraction more than the exact tests.
    // l = 0 exactly (alpha = 0: the central column) is left to the direct routine whatever m2p came out as: m2p = 1 there in
    // real arithmetic, but with |q| << a^2 the sum XT cancels and the computed m2p is rounding noise well away from 1 (found by
    // the campaign of round 6: 2 pixels in 40 000 jobs with beta^2 ~ a^2 cos^2 i on that column -- the reference rejects the
    // ray, this routine called it a miss; the image is the same, the class plane was not).
    // The margin is 1e-12 times max(1, |qla| / |XT|): where XT cancels (qla < 0, |q| << a^2) the computed m2p carries that much
    // more rounding noise (s5_geod.hpp polar_tests_marginal).
    double mmT = 0.0, mK = 0.0;
    if (err == GD_OK) {
        const double margin = 1e-12 * fmax(1.0, fabs(qla) * fabs(mdiv(1.0, XT)));
        const double s_near = s_m2p - margin;
        if (l2 == 0.0) err = GD_E_LAST_BIT;
        else if (m2p <= 0.0) err = GD_E_MUPLUS;
        else if (m2p >= 1.0 - margin) err = GD_E_LAST_BIT;
        else if (q > 0.0) {
            // mK = 1/sqrt(a^2 (m2p + m2m)) first; the modulus m2p/(m2p + m2m) is then m2p a^2 mK^2
            const double rk = rsqrt_pos(a2 * (m2p + m2m));
            mmT = (m2p * a2) * (rk * rk);
            if ((mmT < 0.0) || (mmT >= 1.0)) err = GD_E_MM;
            else if (fabs(p.cos_i) > s_near) err = GD_E_LAST_BIT;
            else mK = rk;
        } else if (q < 0.0) {
            mmT = mdiv(m2p + m2m, m2p);
            if ((mmT < 0.0) || (mmT >= 1.0)) err = GD_E_MM;
            else if (fabs(p.cos_i) > s_near) err = GD_E_LAST_BIT;
            else if (fabs(p.cos_i) < msqrt(-m2m)) err = GD_E_MU0;
            else mK = mdiv(1., msqrt(a2 * m2p));
        } else {
            err = GD_E_Q_RANGE;
        }
    }
    out.err = err;
    const bool ok = (err == GD_OK);
    const double u_i = p.cos_i * rs_m2p;

    // ---------------- what the crossing search needs to know about the ray ----------------
    // "no special case of the inverse function": CONSERVATIVE forms of isn_plain / icn_plain (a handful of comparisons
    // instead of their dozen: whatever they exclude besides the special cases only takes the generic routine, which is right
    // for every argument)
    const bool plain0 = (mR > 1e-8) && (mR < 1.0 - 1e-8) && ((type != T_RC) || ((fabs(zR) < 1.0) && (zR != 0.0)));   // (CC rays never take the addition path)
    const bool plain2 = (u_i > 0.0) && (u_i < 1.0) && (mmT > 0.0) && (mmT < 1.0);
    const bool q_pos = (q > 0.0);
    double uu = u_i;
    const bool u_bad = (uu < -1.0 - 1e-4) || (uu > +1.0 + 1e-4);
    if (uu < -1.0) uu = -1.0;
    if (uu > +1.0) uu = +1.0;
    const bool ladder_class = (type == T_RR) || (type == T_RC);
    const bool may_cross = q_pos && !u_bad;
    // r(P) needs sn (RR) or cn (RC) of modulus mR: the rungs of its Landen ladder are climbed ONCE per ray -- they serve
    // every crossing order and both rays of a pair -- and kept in LDS; lanes that cannot use them climb a short dummy
    LadderLds lad{thin_disk_ladder_column()};
    LadderState lst{};
    if (wave_any(ladder_class && may_cross)) ladder_climb<LadderLds, LADDER_RUNGS, true>(lad, (ladder_class && may_cross) ? mR : 0.5, lst);
    const bool by_add = ok && plain0 && ladder_class && may_cross && !lst.flipped && !lst.degenerate && !lst.incomplete;

    // ---------------- the polar integrals: cn^-1(u_i | mmT) by R_F, K(mmT) from the table ----------------
    double icn_i;
    {
        // x = u_i^2 was formed as a square, its root is |u_i| (first pass of the duplication with one square root);
        // plain lanes have x, y > 0 by construction (u^2 < 1, modulus in [0,1)); the others are redone out of line
        const double z2 = u_i * u_i;
        icn_i = sqrt_pos(1. - z2) * carlson_rf_root_x(fabs(u_i), z2, 1.0 - mmT * (1. - z2));
    }
    double K;
    {
        // K(mm): from the table (kernels.hpp KT_*: 128 polynomials of degree 7 on [0, 0.9], 2e-16) where it reaches,
        // by the arithmetic-geometric mean elsewhere (the lanes' wave with them)
        const bool tab = (p.ktab != nullptr) && (mmT >= 0.0) && (mmT < KT_MMAX);
        K = 0.0;
        if (tab) {
            const double u = mmT * ((double)KT_N / KT_MMAX);
            int i = (int)u;
            i = i < KT_N - 1 ? i : KT_N - 1;
            const double tau = 2.0 * (u - (double)i) - 1.0;
            const double* c = p.ktab + (size_t)i * (KT_DEG + 1);
            double acc = c[KT_DEG];
#pragma unroll
            for (int k = KT_DEG - 1; k >= 0; --k) acc = __builtin_fma(acc, tau, c[k]);
            K = acc;
        }
        if (S5_ANY_MISC(!tab)) {
            if (!tab) K = ell_K(mmT);
        }
    }
    if (S5_ANY_MISC(ok && !plain2)) {
        if (ok && !plain2) icn_i = inv_cn_cold(u_i, mmT);
    }
    if (WANT_STATE) {
        out.a = a; out.l = l; out.q = q; out.beta = beta; out.rp = S5_THIN_RP; out.dP = NAN;
        out.Tpp = 2. * (mK * K); out.Tip = mK * icn_i;
        if (PAIR) {
            out2.a = a; out2.l = l; out2.q = q; out2.beta = -beta; out2.rp = out.rp; out2.dP = NAN;
            out2.Tpp = out.Tpp; out2.Tip = out.Tip;
        }
    }
    if (PAIR) out2.err = err;
    if (!ok) {
        if (err == GD_E_LAST_BIT) { out.cls = PX_COLD_MARK; if (PAIR) out2.cls = PX_COLD_MARK; }
        return;
    }
    out.gtype = type;
    out.cls = PX_MISS;
    if (PAIR) { out2.gtype = type; out2.cls = PX_MISS; }

    // ---------------- equatorial crossings and r(P) (ref :846-885, :291-357) ----------------
    double icn_u = icn_i;
    if (S5_ANY_MISC(uu != u_i && !u_bad && q_pos)) {              // clamped by the slack rule: re-evaluate
        if (uu != u_i && !u_bad && q_pos) icn_u = inv_cn_cold(uu, mmT);
    }
    // mK distributed over the sum in P (two products formed once per ray, not three factors kept per crossing)
    const double mKK = mK * K, mKi = mK * icn_u;
    // constants of the addition theorem: sn, cn, dn of F0 as products.  Few, and the cheap ones are re-formed where they
    // are used: every double kept across the crossing loop is two of the kernel's 128 registers.
    double add_s = 0.0, add_d = 0.0;             // RC: sn(F0), dn(F0);  RR: add_s = cn(F0) dn(F0)
    if (S5_ANY_MISC(by_add)) {
        if (type == T_RC) {
            const double s2 = 1. - zR * zR;                     // sn^2(F0)
            add_s = sqrt_pos(s2); add_d = sqrt_pos(1. - mR * s2);
        } else {
            const double z2 = zR * zR;                          // sn^2(F0)
            add_s = sqrt_pos((1. - z2) * (1. - mR * z2));       // cn(F0) dn(F0)
        }
    }
    constexpr int MEMBERS = PAIR ? 2 : 1;
    // a ray that may cross but is not served by the addition theorem goes the reference's way, after the loops
    bool cold[2] = {may_cross && !by_add, may_cross && !by_add};
    // two inlined passes rather than a run-time loop: as a loop the compiler predicates the pass on per-lane state and the
    // lanes used fall from 97 % to 91 % (measured: +6.5 % VALU instructions, +4.5 % time)
#pragma unroll
    for (int member = 0; member < MEMBERS; ++member) {
        const double beta_m = (member == 0) ? beta : -beta;
        int cls_m = PX_MISS;
        double r_m = NAN, P_m = NAN, dP_m = NAN;
        bool done = cold[member];
#pragma unroll 1
        for (int order = 0; order < p.max_order; ++order) {
            if (!S5_ANY_ORDER(!done)) break;
            if (!done) {
                double P;
                if (!may_cross) P = NAN;
                else {
                    if (beta_m > 0.0) P = (2. * (double)order + 1.) * mKK + mKi;
                    else if (beta_m < 0.0) P = (2. * (double)order + 1.) * mKK - mKi;
                    else P = (2. * (double)order + 1.) * mKK;
                }
                const double wc = (((type == T_RC) ? sqAB : 0.5 * sqAB) * P) * lst.c;    // the angle the descent starts from
                // beyond 2 Rpc for sure: RR (F0 < K) from w = 2 K(mR), RC (F0 < 2 K) from w = 4 K(mR), i.e. w c_N from pi, 2 pi
                if (may_cross && !(wc < ((type == T_RC) ? 2. * S5_PI : S5_PI))) P = NAN;
                if (isnan(P)) { cls_m = (order == 0) ? PX_NAN0 : PX_NAN1; done = true; }
                else {
                    // sn^2(w - F0) = Pn / Q (RR), cn(F0 - w) = X / Y (RC)
                    double Pn = 0.0, Q = 1.0, X = 1.0, Y = 1.0, dP = NAN, r = NAN;
                    int code = CROSS_FORMULA;
                    {
                        double s0, c0, C, ga, N, D;
                        // (from the node table where the launcher attached one -- every launcher of the fast variant does, so all
                        // its image kernels form the same numbers: s5_trig.hpp msincos_tab; a wave-uniform test)
                        if (p.sctab) msincos_tab(p.sctab, wc, s0, c0); else msincos(wc, s0, c0);
                        ladder_descend_squares(lad, lst, s0, c0, C, ga, N, D);
                        // numerators of sn(w) and cn(w) over rho (the signs as ladder_descend assigns them)
                        const double S = (s0 >= 0.0) ? fabs(ga) : -fabs(ga);
                        const double Cc = ((ga >= 0.0) == (s0 >= 0.0)) ? C : -C;
                        const double rho2 = C * C + ga * ga;
                        // 1 - m sn^2(w) sn^2(F0) over rho^2: the theorem's denominator; where it cancels, the direct way
                        const double mz2 = mR * ((type == T_RC) ? 1. - zR * zR : zR * zR);
                        const double den0 = rho2 - mz2 * (S * S);
                        if (!(den0 > 1e-3 * rho2)) { cold[member] = true; done = true; code = CROSS_NONE; }
                        if (type == T_RC) {
                            // cn(F0 - w) = rho (zR Cc D + sn dn(F0) S N) / (D den0).  F0 - w lies in (-4K, 2K): for
                            // w < 2K the sign of sn(F0 - w), i.e. of (sn(F0) Cc N - zR S dn(F0) D) D, says whether
                            // P <= Rpc (for w >= 2K > F0 it is not); past Rpc the ray is still inside 2 Rpc while
                            // cn(F0 - w) > cn(F0) = zR (|F0 - w| < 4K - F0 here)
                            const double rho = sqrt_pos(rho2);
                            X = rho * (zR * Cc * D + (add_s * add_d) * S * N);
                            Y = D * den0;
                            dP = (add_s * Cc * N - zR * S * add_d * D) * D;
                            if (code == CROSS_FORMULA && (!(dP >= 0.0) || !(wc < S5_PI)))
                                code = ((X - zR * Y) * Y > 0.0) ? CROSS_NONE : CROSS_BEYOND;
                        } else {
                            // sn(w - F0) = rho (S cn dn(F0) D - zR Cc N) / (D den0); its sign is that of P - Rpc
                            const double num = S * add_s * D - zR * Cc * N;
                            const double den = D * den0;
                            Pn = rho2 * (num * num);
                            Q = den * den;
                            dP = -(num * D);
                        }
                    }
                    if (code == CROSS_FORMULA) {
                        if (type == T_RR) {
                            // ref :320 divided through by r1 - r4: (r2 - r4)/(r1 - r4) is zR^2 = sn^2(F0)
                            S5_FPC_RADIUS
                            const double z2q = (zR * zR) * Q;
                            const double dnm = z2q - Pn;
                            r = mdiv(ra * z2q - rb * Pn, dnm);
                            if (!(dnm > 0.0)) { code = CROSS_BEYOND; r = NAN; }       // sn^2(w - F0) >= sn^2(F0): w >= 2 F0
                        } else if (type == T_RC) {
                            S5_FPC_RADIUS
                            r = mdiv((rb * Aq - ra * Bq) * Y - (rb * Aq + ra * Bq) * X, (Aq - Bq) * Y - (Aq + Bq) * X);
                        }
                    }
                    if (code == CROSS_BEYOND) { cls_m = (order == 0) ? PX_NAN0 : PX_NAN1; done = true; }
                    if (r >= p.rms) {
                        cls_m = (order == 0) ? PX_HIT0 : PX_HIT1;
                        r_m = r; P_m = P; dP_m = dP;
                        done = true;
                    }
                }
            }
        }
        if (!PAIR || member == 0) { out.cls = cls_m; out.r = r_m; out.P = P_m; if (WANT_STATE) out.dP = dP_m; }
        else { out2.cls = cls_m; out2.r = r_m; out2.P = P_m; if (WANT_STATE) out2.dP = dP_m; }
    }
    // ---------------- g-factor and flux of the accepted crossings ----------------
    const auto& pd = param_reload(p);            // (constant-address-space parameters: the disk's are loaded from here on)
#pragma unroll
    for (int member = 0; member < MEMBERS; ++member) {
        ThinRay& o = (member == 0) ? out : out2;
        if (o.cls == PX_HIT0 || o.cls == PX_HIT1) {
            bool cf = false;
            double x, rx;                                 // sqrt(r) and its reciprocal serve the g-factor and the flux
            sqrt_rsqrt_pos(o.r, x, rx);                   // r >= rms > 0
            o.g = gfactor_kepler_x(o.r, x, a_in, l);
            o.flux = disk_flux_table(pd.disk, o.r, x, rx, cf);
            if (cf) o.flux = FLUX_OWED;          // the closed form, once, at the end of trace_thin_disk_impl (thin_disk_owed_flux)
        }
    }
    // ---------------- the rays left to the direct evaluation ----------------
    // marked for the caller (trace_thin_disk_impl), which runs the direct routine for them from the pixel's coordinates:
    // nothing of this routine's state has to stay alive for it
    if (cold[0]) out.cls = PX_COLD_MARK;
    if (MEMBERS > 1 && cold[1]) out2.cls = PX_COLD_MARK;
}
#undef S5_THIN_RP
#endif   // S5_RPC_ADD

// Image-plane coordinates of a pixel (ref disk-image.c:57-58).  Rows iy and ny - 1 - iy must get beta values that are each
// other's negatives bit for bit: that is what lets the mirrored kernel give the very image of the plain one.  For image sizes
// that are powers of two the fast variant's products with the reciprocals are the reference's numbers exactly; for other
// sizes see the two routines.
template <class PRM>
S5_DEV double pixel_alpha(const PRM& p, int ix)
{
#if S5_FAST
    // the product with 1/nx is the reference's quotient bit for bit when nx is a power of two (every BASELINE size); for the
    // other widths the quotient itself (IEEE division, once per lane): alpha is then the reference's number for every pixel
    if ((p.nx & (p.nx - 1)) == 0) return (((double)(ix) + .5) * p.inv_nx - 0.5) * 2.0 * p.rmax;
    return (((double)(ix) + .5) / (double)(p.nx) - 0.5) * 2.0 * p.rmax;
#else
    return (((double)(ix) + .5) / (double)(p.nx) - 0.5) * 2.0 * p.rmax;
#endif
}

// the reference's expression for every row (ref disk-image.c:58)
template <class PRM>
S5_DEV double pixel_beta_reference(const PRM& p, int iy)
{
    return (((double)(iy) + .5) / (double)(p.ny) - 0.5) * 2.0 * p.rmax * ((double)p.ny / (double)p.nx);
}

template <class PRM>
S5_DEV double pixel_beta(const PRM& p, int iy)
{
#if S5_FAST
    // ny a power of two (every BASELINE size): the product below IS the reference's expression, bit for bit, and antisymmetric.
    // Other heights: the reference's own expression for the rows of the upper half, and for a row of the lower half MINUS the
    // value of its mirror row -- the two rows of a mirrored pair must get opposite beta exactly (the reference's quotients of
    // rows iy and ny-1-iy are not always each other's complement in the last bit), so the upper half has the reference's
    // number and the lower half has it to the rounding of the quotient: (iy + .5) / ny - 0.5 cancels, so the two rows' values
    // round apart by up to 2^-53 / |iy / ny - 0.5| relative (5 and 15 units in the last place in the two cases looked at).  What
    // that costs: r(beta) of the REFERENCE has steps of 1e-9 .. 2e-7 where the iteration count of its own sn / cn changes
    // (CA = 1e-8, src/sim5elliptic.c:544); a lower-half pixel whose beta falls across such a step from the reference's differs
    // from it by the step.  Seen at 3 pixels in 80 000 random jobs (2e9 pixels): 1.7e-7 .. 2.3e-7, DESIGN.md section 5.
    if ((p.ny & (p.ny - 1)) == 0) return ((double)(2 * iy + 1 - p.ny) * (0.5 * p.inv_ny)) * 2.0 * p.rmax * p.ny_over_nx;
    const bool lower = (2 * iy + 1 > p.ny);
    const int jy = lower ? p.ny - 1 - iy : iy;
    const double b = (((double)(jy) + .5) / (double)(p.ny) - 0.5) * 2.0 * p.rmax * p.ny_over_nx;
    return lower ? -b : b;
#else
    return pixel_beta_reference(p, iy);
#endif
}

} // namespace S5NS
