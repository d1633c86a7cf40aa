// k_spectrum.hip -- observed spectrum of a thin disk: image x energy grid in one pass (gfx950).
//
// What the reference does in Python per ray (python/sim5diskraytrace.py:96-123): find the disk crossing
// (first order only, :239-242), T_eff = (F/sigma)^1/4 (python/sim5diskmodel.py:47), local frame of the
// disk surface (tetrad_surface with Omega from the disk's specific angular momentum, :340-348), redshift
// g = k_t / (k.U) (:353-361), emission cosine mu_e = (k.N)/(k.U) (:377-390), and add
// I_nu(E_j / g) g^3 dOmega to every energy bin j with the black-body radiance of
// python/sim5diskspectrum.py:54-88 (colour hardening f, limb darkening 1/2 + 3/4 mu_e).
//
// Kernel: a workgroup traces a 32 x 8 pixel tile with the fused thin-disk routine, each lane leaves
// (T, g, limb-darkening factor) of its pixel in LDS, then the workgroup is re-used TRANSPOSED: thread t owns
// energy bin j = t mod EB and pixel sub-set t / EB, loops over the staged pixels (LDS broadcast reads, no
// conflicts), and accumulates its bin in a register -- no cross-lane reduction per energy.  The sub-set
// partial sums are combined through LDS in a fixed order and the workgroup writes one partial spectrum;
// a second kernel sums the partial spectra over workgroups in index order.  The result is deterministic
// (no atomics).  Output: sum over pixels of I_nu(E_j/g) g^3 in erg cm^-2 s^-1 keV^-1 srad^-1 per pixel; the
// caller multiplies by the pixel solid angle.
#include "s5_thindisk.hpp"
#include "kernels.hpp"

namespace S5NS {

using namespace s5abi;

constexpr int SPEC_TILE_W = 32, SPEC_TILE_H = 8;

#if !S5_FAST
S5_DEV double planck_python(double T, double limbf, double f, double E)
{
    // python/sim5diskspectrum.py:72-86 (its own constants; note kev2freq twice, not 1/freq2kev)
    const double planck_h = 6.626069e-27, kev2freq = 2.417990e+17, c2 = 8.987554e+20, kB = 1.380650e-16;
    const double nu = kev2freq * E;
    return limbf * 2.0 * planck_h * (nu * nu * nu) / c2 / (f * f * f * f) *
           1. / (exp((planck_h * kev2freq * E) / (kB * f * T)) - 1.0) * kev2freq;
}
#endif

// (T, g, limb-darkening factor) of one traced ray (python/sim5diskraytrace.py:96-123, 340-390); g = 0: contributes nothing
template <class PRM>
S5_DEV void spectrum_pixel(const PRM& p, const SpectrumParams& sp, const ThinRay& t, double& T, double& g, double& limbf)
{
    T = 0.0; g = 0.0; limbf = 0.0;
    if (t.cls == PX_HIT0 && t.flux != 0.0) {
        double k[4], U[4], N[4];
        const double e0[4] = { 1.0, 0.0, 0.0, 0.0 }, e2[4] = { 0.0, 0.0, 1.0, 0.0 };
        photon_momentum(p.a, t.r, 0.0, t.l, t.q, t.dP, 1.0, k);           // ref py :250
        Metric mt;
        kerr_metric(p.a, t.r, 0.0, mt);
        Tetrad tt;
        tetrad_surface(mt, omega_from_ell(disk_ell(p.disk, t.r), mt), 0.0, 0.0, tt);
        on2bl(e0, U, tt);
        on2bl(e2, N, tt);
        const double kU = dot(k, U, mt);
        const double gg = mdiv(k[0] * mt.g00 + k[3] * mt.g03, kU);
        double mue = mdiv(dot(k, N, mt), kU);
        if ((mue < 0.0) && (mue > -1e-2)) mue = 1e-3;                            // ref py :387
        if (gg > 0.0) {
            g = gg;
            T = sqrt(sqrt(t.flux / 5.670400e-05));
            limbf = (sp.limb_darkening > 0) ? ((mue >= 0.0) ? 0.5 + 0.75 * mue : 1.0) : 1.0;
        }
    }
}

#if S5_FAST
// What the loop over the energies needs of a pixel -- x1 = log2(e) h kev2freq / (kB f T g) and the amplitude limb(mu_e) amp0 -- for a
// crossing of the EQUATORIAL PLANE by a disk on circular orbits (m = 0, v_r = 0, dH/dR = 0: the thin disk of this job), in closed
// form; (1, 0) for a dark pixel (no branch in the loop over the pixels).  With k_t = -1, k_phi = l (the constants of motion;
// photon_momentum normalises to them), U = A (1, 0, 0, Omega) with A^-2 = -(g00 + 2 Omega g03 + Omega^2 g33) (ref
// src/sim5kerr.c:871-875) and the surface normal N = (0, 0, -1/sqrt(g22), 0) (ref :898-902):
//     k.U = A (-1 + Omega l)            g     = k_t / (k.U)         = 1 / (A (1 - Omega l))
//     k.N = -sqrt(g22) k^theta          mu_e  = (k.N) / (k.U)       = g sqrt(q) / r        (k^theta = +sqrt(q) / r^2 at m = 0, ref :1179-1207)
// -- what the tetrad chain of spectrum_pixel evaluates with ~600 operations (photon_momentum, the contravariant metric, three
// normalised tetrad legs, two on2bl, two dot products).  Omega is the reference's own mix: the Keplerian l(r) of the DISK MODEL,
// whose spin is a float static (ref src/sim5disk-nt.c:27-28, :260-266), turned into an angular velocity by the metric of the
// hole's double spin (ref :1101-1111) -- 3e-8 from the Omega_K of either spin, and x = E / (kT g) carries that into the Wien tail
// multiplied by x, so the trace's own g-factor (gfactorK of the double spin) is NOT used here.  Written without the intermediate
// quotients: with l(r) = Nl / Dl, r g00 = 2 - r, r g03 = -2a, r g33 = r (r^2 + a^2) + 2 a^2 (m = 0),
//     Omega = No / Do,   No = -(r g03 Dl + Nl r g00),   Do = r g33 Dl + Nl r g03
//     A^-2  = P / (r Do^2),   P = -(r g00 Do^2 + 2 No Do r g03 + No^2 r g33)
//     g     = sign(Do) sqrt(P / r) / (Do - No l)
// and g, 1 / (g T) from ONE reciprocal of sqrt(P / r) (Do - No l) T: four square roots and two reciprocals per pixel (the chain of
// quotients: four and five).  Agreement with the tetrad chain: rounding (tests/test_py_diskraytrace.py, 1e-6 of every bin against
// the strict kernel and against the reference's Python classes).
template <class PRM>
S5_DEV void spectrum_stage_equatorial(const PRM& p, const SpectrumParams& sp, const ThinRay& t, double l, double q, double sqrt_q,
                                      double x_scale, double amp0, double& x1, double& amp)
{
    x1 = 1.0; amp = 0.0;
    // The reference forms the photon's momentum at the crossing (photon_momentum, ref src/sim5kerr.c:1175-1195) and drops the
    // pixel when that is not a number (g = NaN fails `g > 0`, ref python/sim5diskraytrace.py:113): R(r) = (r^2 + a^2 - a l)^2 -
    // Delta ((l - a)^2 + q) below -1e-8.  In real arithmetic R >= 0 on a geodesic; a crossing AT the ray's pericentre (P = Rpc:
    // r = rp exactly, ref src/sim5kerr-geod.c:309) has R = 0 and the reference's sum of terms of 1e8 comes out at +-1e-8 -- a
    // pixel it keeps or drops by rounding.  The closed form below needs no k^r, but the same pixels must count: the
    // reference's expression, its operations in its order (round 6: tests/tools/fuzz_spectrum.py 1500 6301 uniform, case 575 --
    // the strict kernel without the pixel at r = rp = 99.43, this one with it: 2.8e-5 of the spectrum).
    bool momentum_is_a_number = true;
    {
        const double a = p.a, r = t.r;
        const double a2 = a * a, r2 = r * r;
        const double D = r2 - 2. * r + a2;
        double R = sq(r2 + a2 - a * l) - D * (sq(l - a) + q);
        if ((R < 0.0) && (-R < 1e-8)) R = 0.0;
        double M = q;                                                          // (m = 0: M = q - l^2 m^2 / (1 - m^2) + a^2 m^2)
        if ((M < 0.0) && (-M < 1e-8)) M = 0.0;
        momentum_is_a_number = !(R < 0.0) && !(M < 0.0);
    }
    if (t.cls == PX_HIT0 && t.flux != 0.0 && momentum_is_a_number) {
        const double a = p.a, af = p.disk.a, r = t.r;
        const double rl = fmax(p.disk.rms, r);                                 // disk_ell: l(r) of the inner edge below it
        const double x = sqrt_pos(rl);
        const double Nl = rl * rl - 2. * af * x + af * af, Dl = x * rl - 2. * x + af;
        const double G00 = 2. - r, G03 = -2. * a, G33 = r * (r * r + a * a) + 2. * (a * a);
        const double No = -(G03 * Dl + Nl * G00), Do = G33 * Dl + Nl * G03;
        const double P = -(G00 * (Do * Do) + 2. * (No * Do) * G03 + (No * No) * G33);
        const double rr = mrcp(r);
        const double s = msqrt(P * rr);                                        // (P < 0: NaN, the ray contributes nothing, as there)
        double den = Do - No * l;
        if (Do < 0.0) den = -den;
        const double u = t.flux * (1.0 / 5.670400e-05);                        // T^4
        const double T = sqrt_pos(sqrt_pos(u));
        const double R = mrcp((s * den) * T);
        const double gg = (s * s) * (T * R);
        if ((gg > 0.0) && !(T < 1e2)) {                                        // ref py :76
            double mue = (gg * sqrt_q) * rr;
            if ((mue < 0.0) && (mue > -1e-2)) mue = 1e-3;                      // ref py :387
            const double limbf = (sp.limb_darkening > 0) ? ((mue >= 0.0) ? 0.5 + 0.75 * mue : 1.0) : 1.0;
            x1 = x_scale * ((den * den) * R);
            amp = limbf * amp0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// FAST VARIANT.  Three changes against the kernel below (which stays the strict variant's):
//  * The local frame of a crossing in closed form (spectrum_stage_equatorial above).
//  * A row set symmetric about the middle of the image is traced in MIRRORED PAIRS (s5_thindisk.hpp: a lane traces (alpha, beta)
//    and (alpha, -beta), which share the geodesic) at four waves per SIMD: a workgroup stages 512 pixels, not 256.  The staging
//    arrays live in the Landen-ladder block of the trace, which is dead by then (LDS stays at 34 KB: four workgroups per CU).
//  * The Planck factor 1 / (e^x - 1) of a (pixel, energy) pair costs 15.25 issue slots instead of ~45 (a full-precision exp and
//    a Newton division): planck_sum below.  The bar on the spectrum is 1e-6 (tests/test_py_diskraytrace.py::test_fused_spectrum_kernel,
//    against python/sim5diskspectrum.py:54-88).  Per pixel: log2(e) h kev2freq / (kB f T g) and the amplitude; per energy bin
//    E^3 is applied once, after the loop over the pixels.
// ---------------------------------------------------------------------------------------------------------------------------
#define S5_SPEC_WAVES 4
constexpr int FAST_TILE_W = 16, FAST_TILE_H = 16;        // a wave: a 16 x 4 patch, as in the image kernels (image neighbours share class and trip counts)

// sum over the staged pixels of amp / (e^x - 1) for this lane's energy, x log2(e) = E sX[q] = n + f.  15.25 issue slots per pair
// (16 in the form for fewer than 64 energies per pass):
//  * t + M with M = 1.5 2^52 in ONE fma (the sum is rounded to an integer: that is n, to nearest even, and its two's
//    complement sits in the low word of the result: no conversion), n = (t + M) - M, f = fma(E, sX, -n) -- exact;
//  * 2^f = 1 + f (c1 + f (c2 + ... + f c6)): (2^f - 1) / f on [-1/2, 1/2] by the degree-5 polynomial of least maximal RELATIVE
//    error (tests/tools/exp2_coefficients.py: 1.07e-8; the subtraction of 1 below is exact to 1e-16 / x -- the property of the
//    reference's own exp(x) - 1.0, ref python/sim5diskspectrum.py:84 -- so small x keeps its accuracy) -- five Horner steps
//    with the constants in scalar registers and one fma;
//  * 2^n by v_ldexp on the low word.  Run-time stride (GROUPS == 0): 2^n 2^f - 1 in one fma (n beyond the exponent range gives
//    infinity, a reciprocal of 0 and a term of 0: no cap on x needed), the 26-bit reciprocal seed (four slots), one fma for the sum.
//    Compile-time stride: the same with t = -x log2(e): u = 2^n 2^f = e^-x in (0, 1] (an n below the exponent range gives 0 and a
//    term of 0), w = 1 - u, b = amp u, and EIGHT terms b_i / w_i share one reciprocal: N / D from a tree of seven (N, D) pairs
//    (N = N1 D2 + N2 D1, D = D1 D2: three slots each; every w is in (0, 1], so D cannot overflow) -- 12 + (21 + 4 + 1) / 8 slots.
// The low word is n only while |t| < 2^31.  CLAMP (a workgroup with a pixel whose sX times the largest energy is beyond 2^30:
// T g of a few kelvin) bounds t first, a few more slots; 2^(2^30) is as infinite as 2^t.
// The pixels are staged as (sX, amplitude) pairs: one 16-byte LDS read per term.
//
// GROUPS > 0: the stride over the pixels is that compile-time number and `first` is the same in every lane of the wave (256 /
// GROUPS >= 64 energies per pass: a wave is 64 energies of ONE pixel sub-set) -- the loop counter lives on the scalar unit
// and eight terms are read at immediate offsets from one address: no vector instruction of the loop is bookkeeping.
// GROUPS == 0: stride and start at run time (fewer than 64 energies per pass).
template <bool CLAMP, int GROUPS>
S5_DEV double planck_sum(const double2* __restrict__ sXA, int first, int step, int npix, double E)
{
    constexpr double C1 = 0.6931471879266856, C2 = 0.2402264979496441, C3 = 0.05550357433648187, C4 = 0.009618237494183314,
                     C5 = 0.0013390735475399872, C6 = 0.00015403512618661003;
    constexpr double M = 6755399441055744.0;                             // 1.5 2^52
    // C6 stays in a vector register pair (a VOP3 instruction takes one scalar operand: the other constants)
    double c6 = C6;
    asm volatile("" : "+v"(c6));
    // one pair: ~14 instructions of ONE dependent chain; four chains interleaved by hand (the compiler does not unroll a loop
    // of run-time trip count) so that an instruction's latency is covered by its twins, four sums added at the end
    auto term = [&](const double2 xa, double acc) -> double {
        double tm, f;
        if (CLAMP) {
            double t = E * xa.x;
            if (t > 1073741824.0) t = 1073741824.0;                      // (not fmin: a NaN stays one)
            tm = t + M;
            f = t - (tm - M);
        } else {
            tm = __builtin_fma(E, xa.x, M);
            f = __builtin_fma(E, xa.x, -(tm - M));
        }
        double e = hfmac(f, c6, C5);
        e = hfmac(f, e, C4);
        e = hfmac(f, e, C3);
        e = hfmac(f, e, C2);
        e = hfmac(f, e, C1);
        e = __builtin_fma(f, e, 1.0);                                   // 2^f
        const double two_n = __builtin_amdgcn_ldexp(1.0, __double2loint(tm));
        const double den = __builtin_fma(two_n, e, -1.0);
        return __builtin_fma(xa.y, __builtin_amdgcn_rcp(den), acc);
    };
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    if (GROUPS > 0) {
        // eight terms over ONE reciprocal (see above): u = e^-x = 2^(-t), term = a u / (1 - u), the sum of eight as N / D
        const double nE = -E;
        auto uw = [&](const double2 xa, double& b, double& w) {
            double tm, f;
            if (CLAMP) {
                double t = nE * xa.x;
                if (t < -1073741824.0) t = -1073741824.0;
                tm = t + M;
                f = t - (tm - M);
            } else {
                tm = __builtin_fma(nE, xa.x, M);
                f = __builtin_fma(nE, xa.x, -(tm - M));
            }
            double e = hfmac(f, c6, C5);
            e = hfmac(f, e, C4);
            e = hfmac(f, e, C3);
            e = hfmac(f, e, C2);
            e = hfmac(f, e, C1);
            e = __builtin_fma(f, e, 1.0);
            const double u = __builtin_amdgcn_ldexp(e, __double2loint(tm));
            w = 1.0 - u;
            b = xa.y * u;
        };
        const int start = __builtin_amdgcn_readfirstlane(first);
        for (int q = start; q < npix; q += 8 * GROUPS) {
            const double2* const at = sXA + q;
            double b0, b1, b2, b3, b4, b5, b6, b7, w0, w1, w2, w3, w4, w5, w6, w7;
            uw(at[0 * GROUPS], b0, w0); uw(at[1 * GROUPS], b1, w1); uw(at[2 * GROUPS], b2, w2); uw(at[3 * GROUPS], b3, w3);
            uw(at[4 * GROUPS], b4, w4); uw(at[5 * GROUPS], b5, w5); uw(at[6 * GROUPS], b6, w6); uw(at[7 * GROUPS], b7, w7);
            const double N01 = __builtin_fma(b0, w1, b1 * w0), D01 = w0 * w1;
            const double N23 = __builtin_fma(b2, w3, b3 * w2), D23 = w2 * w3;
            const double N45 = __builtin_fma(b4, w5, b5 * w4), D45 = w4 * w5;
            const double N67 = __builtin_fma(b6, w7, b7 * w6), D67 = w6 * w7;
            const double Na = __builtin_fma(N01, D23, N23 * D01), Da = D01 * D23;
            const double Nb = __builtin_fma(N45, D67, N67 * D45), Db = D45 * D67;
            const double N = __builtin_fma(Na, Db, Nb * Da), D = Da * Db;
            acc0 = __builtin_fma(N, __builtin_amdgcn_rcp(D), acc0);
        }
    } else {
        int q = first;
        for (; q + 3 * step < npix; q += 4 * step) {
            acc0 = term(sXA[q], acc0); acc1 = term(sXA[q + step], acc1); acc2 = term(sXA[q + 2 * step], acc2); acc3 = term(sXA[q + 3 * step], acc3);
        }
        for (; q < npix; q += step) acc0 = term(sXA[q], acc0);
    }
    return (acc0 + acc1) + (acc2 + acc3);
}

// UNIFORM ENERGY GRID (round 6): E_j = E_0 + j dE.  Then u_j = e^-x_j = 2^(-E_j sX) of a pixel obeys u_(j+1) = u_j tau with
// tau = 2^(-dE sX): ONE multiplication where the general loop spends eleven slots on an exponential.  Lanes own RUNS of eight
// consecutive energies (thread t: run t mod RP, pixel sub-set t / RP); a lane takes eight of its pixels at a time -- u at the
// head of its run by the exponential of planck_sum (1e-8, as there; the energy is the grid's own value, not E_0 + j dE), tau
// from the staged column (made once per pixel by the exponential of s5_trig.hpp: the eight-fold product must not carry eight
// times 1e-8) -- and walks the eight energies: per energy w = 1 - b / amp for the eight pixels (one fma each), their eight terms b / w
// over ONE reciprocal (the tree of planck_sum), b *= tau.  2 + 3.25 slots per (pixel, energy) pair + 11 / 8 for the head of the
// run: ~6.6 against 15.25.  acc[k] = the lane's sum for energy k of its run.
template <int RP>
S5_DEV void planck_runs_uniform(const double2* __restrict__ sXA, const double2* __restrict__ sTR, int subset, int npix, double E_head, double acc[8])
{
    constexpr double C1 = 0.6931471879266856, C2 = 0.2402264979496441, C3 = 0.05550357433648187, C4 = 0.009618237494183314,
                     C5 = 0.0013390735475399872, C6 = 0.00015403512618661003;
    constexpr double M = 6755399441055744.0;                             // 1.5 2^52
    constexpr int SUBSETS = 256 / RP;
    double c6 = C6;
    asm volatile("" : "+v"(c6));
    const double nE = -E_head;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.0;
    for (int q = subset; q < npix; q += 8 * SUBSETS) {
        // what is carried along a run is b = amp u (the numerator of the term); the denominator 1 - u = 1 - b / amp is ONE fma
        // with the staged reciprocal of the amplitude (0 for a dark pixel: b = 0, w = 1, a term of 0) -- two slots per pair
        // before the tree instead of three (u *= tau, w = 1 - u, b = amp u)
        double b[8], tau[8], ramp[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double2 xa = sXA[q + i * SUBSETS];
            const double2 tr = sTR[q + i * SUBSETS];
            tau[i] = tr.x; ramp[i] = tr.y;
            const double tm = __builtin_fma(nE, xa.x, M);
            const double f = __builtin_fma(nE, xa.x, -(tm - M));
            double e = hfmac(f, c6, C5);
            e = hfmac(f, e, C4);
            e = hfmac(f, e, C3);
            e = hfmac(f, e, C2);
            e = hfmac(f, e, C1);
            e = __builtin_fma(f, e, 1.0);
            b[i] = xa.y * __builtin_amdgcn_ldexp(e, __double2loint(tm));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = __builtin_fma(-b[i], ramp[i], 1.0);
            const double N01 = __builtin_fma(b[0], w[1], b[1] * w[0]), D01 = w[0] * w[1];
            const double N23 = __builtin_fma(b[2], w[3], b[3] * w[2]), D23 = w[2] * w[3];
            const double N45 = __builtin_fma(b[4], w[5], b[5] * w[4]), D45 = w[4] * w[5];
            const double N67 = __builtin_fma(b[6], w[7], b[7] * w[6]), D67 = w[6] * w[7];
            const double Na = __builtin_fma(N01, D23, N23 * D01), Da = D01 * D23;
            const double Nb = __builtin_fma(N45, D67, N67 * D45), Db = D45 * D67;
            const double N = __builtin_fma(Na, Db, Nb * Da), D = Da * Db;
            acc[k] = __builtin_fma(N, __builtin_amdgcn_rcp(D), acc[k]);
            if (k < 7) {
#pragma unroll
                for (int i = 0; i < 8; ++i) b[i] *= tau[i];
            }
        }
    }
}

template <bool CLAMP>
S5_DEV double planck_sum_for(const double2* __restrict__ sXA, int first, int groups, int npix, double E)
{
    switch (groups) {
        case 1: return planck_sum<CLAMP, 1>(sXA, first, 1, npix, E);
        case 2: return planck_sum<CLAMP, 2>(sXA, first, 2, npix, E);
        case 4: return planck_sum<CLAMP, 4>(sXA, first, 4, npix, E);
        default: return planck_sum<CLAMP, 0>(sXA, first, groups, npix, E);
    }
}

// The workgroups' partial spectra are written ENERGY-MAJOR -- partial[j * nblocks + workgroup] -- so that ONE more launch adds
// them up: spectrum_sum_kernel, a workgroup per energy reading its row contiguously, every value added at a fixed place of a
// fixed tree (deterministic, no floating-point atomics).  (The strict variant keeps the two-level tree of spectrum_reduce_kernel
// launches; an in-kernel tree by the last workgroup to arrive was built and measured: the device-scope fences it needs write
// back and invalidate the L2 of every XCD -- the job went from 0.14 to 0.27 ms.)
template <bool PAIR>
__global__ __launch_bounds__(256, S5_SPEC_WAVES)
void disk_spectrum_fast_kernel(ImageParams p, SpectrumParams sp, const double* __restrict__ energies,
                               double* __restrict__ partial)
{
    const int tid = threadIdx.x;
    const int lane_x = tid % FAST_TILE_W, lane_y = tid / FAST_TILE_W;
    const int ix = blockIdx.x * FAST_TILE_W + lane_x;
    const int lr = blockIdx.y * FAST_TILE_H + lane_y;                       // PAIR: local row in the upper half
    const int half = PAIR ? (p.nrows + 1) / 2 : p.nrows;
    // what this lane stages for its (two) pixels: (1, 0) = dark
    double x0 = 1.0, amp_0 = 0.0, x1 = 1.0, amp_1 = 0.0;
    if (ix < p.nx && lr < half) {
        const int iy = p.y0 + lr;
        const double alpha = pixel_alpha(p, ix), beta = pixel_beta(p, iy);
        ThinRay t, t2;
        if (PAIR) trace_thin_disk_impl<false, true, false, true>(p, alpha, beta, t, t2, iy);     // max_order = 1, rms = 0: first crossing, any radius
        else trace_thin_disk<false, true>(p, alpha, beta, t, iy);
        const bool second = PAIR && (p.nrows - 1 - lr != lr);                // (an odd middle row is its own mirror)
        // the constants of motion of the pair (ref src/sim5kerr-geod.c:76-77) for the frame: l, sqrt(q)
        const double l = -alpha * p.sin_i;
        const double b = (beta == 0.0) ? +1e-6 : beta;
        const double q = constant_q(b, p.cos_i, alpha, p.a);                                        // ref src/sim5kerr-geod.c:77 (the caller's spin; its roundings: s5_geod.hpp)
        const double sqrt_q = msqrt(q);                                                             // (q < 0: NaN, no limb darkening, as with the tetrads)
        const double planck_h = 6.626069e-27, kev2freq = 2.417990e+17, c2 = 8.987554e+20, kB = 1.380650e-16;
        const double f = sp.hardening;
        const double amp0 = mdiv(2.0 * planck_h * (kev2freq * kev2freq * kev2freq) * kev2freq, c2 * (f * f * f * f));
        const double x_scale = mdiv(1.44269504088896340736 * (planck_h * kev2freq), kB * f);
        spectrum_stage_equatorial(p, sp, t, l, q, sqrt_q, x_scale, amp0, x0, amp_0);
        if (second) spectrum_stage_equatorial(p, sp, t2, l, q, sqrt_q, x_scale, amp0, x1, amp_1);
    }
    // the largest |energy| of the job, by every wave for itself (a few loads and six lane exchanges; no LDS)
    double e_max = 0.0;
    for (int j = tid % 64; j < sp.n_energies; j += 64) e_max = fmax(e_max, fabs(energies[j]));
    for (int w = 32; w > 0; w >>= 1) e_max = fmax(e_max, __shfl_xor(e_max, w));
    // a UNIFORM grid E_j = E_0 + j dE of at least 64 energies (every wave looks at all of them: the four agree): the recurrence
    // of planck_runs_uniform.  The tolerance, 1e-13 of the largest energy, is what grids made by different expressions of the
    // same step differ by; in the spectrum it is x 1e-13 <= 1e-10.
    double dE = 0.0;
    bool uniform = false;
    if (sp.n_energies >= 64) {
        const double E0 = energies[0];
        dE = (energies[sp.n_energies - 1] - E0) / (double)(sp.n_energies - 1);
        bool off = !(dE > 0.0) || !(E0 > 0.0);
        for (int j = tid % 64; j < sp.n_energies; j += 64) off = off || !(fabs(energies[j] - (E0 + (double)j * dE)) <= 1e-13 * e_max);
        uniform = __builtin_amdgcn_ballot_w64(off) == 0ull;
    }
    // the staging arrays take over the ladder block of the trace (every lane is through with it)
    double* const lds = thin_disk_ladder_column() - threadIdx.x;
    // [512] pairs: log2(e) h kev2freq / (kB f T g) (a harmless 1 for a dark pixel), amplitude (0 for a dark pixel)
    double2* const sXA = reinterpret_cast<double2*>(lds);                    // (the ladder block starts on a 16-byte boundary: s5_thindisk.hpp)
    double* const sAcc = lds + 1024;                                         // [256]
    __syncthreads();
    sXA[tid] = make_double2(x0, amp_0);
    sXA[tid + 256] = make_double2(x1, amp_1);
    double2* const sTR = reinterpret_cast<double2*>(lds + 1024);             // [512] uniform grid: (2^(-dE sX), 1 / amplitude) of every staged pixel
    double* const sRed = lds + 2048;                                         // [2048] uniform grid: the lanes' sums, [energy of the pass][sub-set]
    if (uniform) {
        // e^(-dE x) to full precision (s5_trig.hpp mexp: 1.7e-16); an argument below -700 is 0 for every purpose of the loop
        const double ln2 = 0.693147180559945309417;
        sTR[tid] = make_double2(mexp(fmax(-(dE * x0) * ln2, -700.0)), (amp_0 > 0.0) ? 1.0 / amp_0 : 0.0);
        sTR[tid + 256] = make_double2(mexp(fmax(-(dE * x1) * ln2, -700.0)), (amp_1 > 0.0) ? 1.0 / amp_1 : 0.0);
    }
    const bool beyond = !(x0 * e_max < 1073741824.0) || !(x1 * e_max < 1073741824.0);
    // (the barrier the staged pixels need anyway)  any pixel whose exponent could leave the 32-bit range: planck_sum<true>
    const bool clamp = __syncthreads_or(beyond) != 0;

    if (uniform && !clamp) {
        // transposed phase of the uniform grid: RP runs of eight energies x (256 / RP) pixel sub-sets per pass
        const int npix_u = PAIR ? 512 : 256;
        const int runs = (sp.n_energies + 7) / 8;
        const size_t nblk = (size_t)gridDim.x * gridDim.y, blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        for (int r0 = 0, rp = 16; r0 < runs; r0 += rp) {
            rp = (runs - r0 > 8) ? 16 : 8;                                    // (a last pass of up to eight runs: twice the pixel sub-sets)
            const int run = r0 + tid % rp, subset = tid / rp;
            const int jh = 8 * run;
            double acc[8];
            const bool live = run < runs;
            const double Eh = live ? energies[jh] : 1.0;
            if (rp == 16) planck_runs_uniform<16>(sXA, sTR, subset, npix_u, Eh, acc);
            else planck_runs_uniform<8>(sXA, sTR, subset, npix_u, Eh, acc);
            const int subsets = 256 / rp;
#pragma unroll
            for (int k = 0; k < 8; ++k) sRed[((tid % rp) * 8 + k) * subsets + subset] = acc[k];
            __syncthreads();
            if (tid < rp * 8) {
                const int j = 8 * r0 + tid;
                if (j < sp.n_energies) {
                    double tot = 0.0;
                    for (int g2 = 0; g2 < subsets; ++g2) tot += sRed[tid * subsets + g2];
                    const double E = energies[j];
                    partial[(size_t)j * nblk + blk] = tot * (E * E * E);
                }
            }
            __syncthreads();
        }
        return;
    }

    // transposed phase: EB energy bins x (256 / EB) pixel sub-sets
    const int EB = sp.bins_per_pass;                     // power of two, <= 256
    const int groups = 256 / EB;
    const int jj = tid % EB, grp = tid / EB;
    const int npix = PAIR ? 512 : 256;
    for (int j0 = 0; j0 < sp.n_energies; j0 += EB) {
        const int j = j0 + jj;
        double acc = 0.0;
        if (j < sp.n_energies) {
            const double E = energies[j];
            acc = (clamp ? planck_sum_for<true>(sXA, grp, groups, npix, E) : planck_sum_for<false>(sXA, grp, groups, npix, E)) * (E * E * E);
        }
        sAcc[tid] = acc;
        __syncthreads();
        if (grp == 0 && j < sp.n_energies) {
            double tot = 0.0;
            for (int s = 0; s < groups; ++s) tot += sAcc[s * EB + jj];
            const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            partial[(size_t)j * ((size_t)gridDim.x * gridDim.y) + blk] = tot;
        }
        __syncthreads();
    }
}

// spectrum[j] = sum over the workgroups of partial[j * nblocks + b]: thread t adds b = t, t + 256, ... in order, the 256 sums
// are added pairwise through LDS in a fixed pattern
__global__ __launch_bounds__(256)
void spectrum_sum_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ spectrum)
{
    __shared__ double s[256];
    const double* row = partial + (size_t)blockIdx.x * (size_t)nblocks;
    double tot = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) tot += row[b];
    s[threadIdx.x] = tot;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) spectrum[blockIdx.x] = s[0];
}
#endif   // S5_FAST

#if !S5_FAST
__global__ __launch_bounds__(256, 2)
void disk_spectrum_kernel(ImageParams p, SpectrumParams sp, const double* __restrict__ energies,
                          double* __restrict__ partial)
{
    __shared__ double sT[256], sG[256], sL[256];
    __shared__ double sAcc[256];
    const int tid = threadIdx.x;
    const int lane_x = tid % SPEC_TILE_W, lane_y = tid / SPEC_TILE_W;
    const int ix = blockIdx.x * SPEC_TILE_W + lane_x;
    const int lr = blockIdx.y * SPEC_TILE_H + lane_y;
    double T = 0.0, g = 0.0, limbf = 0.0;
    if (ix < p.nx && lr < p.nrows) {
        const int iy = p.y0 + lr;
        const double alpha = (((double)(ix) + .5) / (double)(p.nx) - 0.5) * 2.0 * p.rmax;
        const double beta = (((double)(iy) + .5) / (double)(p.ny) - 0.5) * 2.0 * p.rmax *
                            ((double)p.ny / (double)p.nx);
        ThinRay t;
        trace_thin_disk<true>(p, alpha, beta, t);            // max_order = 1, rms = 0: first crossing, any radius
        spectrum_pixel(p, sp, t, T, g, limbf);
    }
    sT[tid] = T; sG[tid] = g; sL[tid] = limbf;
    __syncthreads();

    // transposed phase: EB energy bins x (256 / EB) pixel sub-sets
    const int EB = sp.bins_per_pass;                     // power of two, <= 256
    const int groups = 256 / EB;
    const int jj = tid % EB, grp = tid / EB;
    for (int j0 = 0; j0 < sp.n_energies; j0 += EB) {
        const int j = j0 + jj;
        double acc = 0.0;
        if (j < sp.n_energies) {
            const double E = energies[j];
            for (int q = grp; q < 256; q += groups) {
                const double gq = sG[q];
                if (gq > 0.0) {
                    const double Tq = sT[q];
                    if (!(Tq < 1e2))                         // ref py :76
                        acc += planck_python(Tq, sL[q], sp.hardening, mdiv(E, gq)) * (gq * gq * gq);
                }
            }
        }
        sAcc[tid] = acc;
        __syncthreads();
        if (grp == 0 && j < sp.n_energies) {
            double tot = 0.0;
            for (int s = 0; s < groups; ++s) tot += sAcc[s * EB + jj];
            const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            partial[blk * (size_t)sp.n_energies + j] = tot;
        }
        __syncthreads();
    }
}

#endif   // !S5_FAST

// one level of the (deterministic) tree sum over workgroup partials: row c of dst = sum of rows
// [c * SPEC_FAN, (c + 1) * SPEC_FAN) of src, energies across the lanes (coalesced)
constexpr int SPEC_FAN = 64;
__global__ __launch_bounds__(256)
void spectrum_reduce_kernel(const double* __restrict__ src, size_t rows, int n_energies, double* __restrict__ dst)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_energies) return;
    const size_t r0 = (size_t)blockIdx.y * SPEC_FAN;
    const size_t r1 = (r0 + SPEC_FAN < rows) ? r0 + SPEC_FAN : rows;
    double tot = 0.0;
    for (size_t b = r0; b < r1; ++b) tot += src[b * (size_t)n_energies + j];
    dst[(size_t)blockIdx.y * n_energies + j] = tot;
}

} // namespace S5NS

#if S5_FAST
int s5_launch_disk_spectrum_fast(const s5abi::ImageParams& p, const s5abi::SpectrumParams& sp,
                                 const double* energies, double* partial, double* spectrum, hipStream_t stream)
#else
int s5_launch_disk_spectrum_strict(const s5abi::ImageParams& p, const s5abi::SpectrumParams& sp,
                                   const double* energies, double* partial, double* spectrum, hipStream_t stream)
#endif
{
    using namespace S5NS;
#if S5_FAST
    // a row set symmetric about the middle of the image: mirrored pairs, 32 x (8 + 8) pixels per workgroup
    const bool pair = (p.y0 + p.y1 == p.ny) && p.nrows >= 2;
    const int tile_rows = pair ? (p.nrows + 1) / 2 : p.nrows;
    const dim3 grid((p.nx + FAST_TILE_W - 1) / FAST_TILE_W, (tile_rows + FAST_TILE_H - 1) / FAST_TILE_H);
    if (pair) hipLaunchKernelGGL(disk_spectrum_fast_kernel<true>, grid, dim3(256), 0, stream, p, sp, energies, partial);
    else hipLaunchKernelGGL(disk_spectrum_fast_kernel<false>, grid, dim3(256), 0, stream, p, sp, energies, partial);
    hipError_t e1 = hipGetLastError();
    if (e1 != hipSuccess) return (int)e1;
    hipLaunchKernelGGL(spectrum_sum_kernel, dim3((unsigned)sp.n_energies), dim3(256), 0, stream, partial, (int)(grid.x * grid.y), spectrum);
    return (int)hipGetLastError();
#else
    const dim3 grid((p.nx + SPEC_TILE_W - 1) / SPEC_TILE_W, (p.nrows + SPEC_TILE_H - 1) / SPEC_TILE_H);
    hipLaunchKernelGGL(disk_spectrum_kernel, grid, dim3(256), 0, stream, p, sp, energies, partial);
#endif
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    // tree sum, fan-in 64 per level, ping-pong between the partial rows and the scratch rows behind them
    // (workspace = (nblocks + ceil(nblocks / 64)) rows); the last level writes the spectrum
    const size_t nblocks = (size_t)grid.x * grid.y;
    double* bufs[2] = { partial, partial + nblocks * (size_t)sp.n_energies };
    size_t rows = nblocks;
    int cur = 0;
    for (;;) {
        const size_t out_rows = (rows + SPEC_FAN - 1) / SPEC_FAN;
        double* dst = (out_rows == 1) ? spectrum : bufs[cur ^ 1];
        hipLaunchKernelGGL(spectrum_reduce_kernel, dim3((sp.n_energies + 255) / 256, (unsigned)out_rows), dim3(256), 0, stream,
                           bufs[cur], rows, sp.n_energies, dst);
        if ((e = hipGetLastError()) != hipSuccess) return (int)e;
        if (out_rows == 1) break;
        rows = out_rows; cur ^= 1;
    }
    return 0;
}
