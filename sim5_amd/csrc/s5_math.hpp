// s5_math.hpp -- FP64 primitives of the device code, in the two build variants of s5_config.hpp.
//
// gfx950 has no FP64 divide or square-root instruction: both are software sequences around the
// quarter-rate v_rcp_f64 / v_rsq_f64 seeds (~26 good bits).  The compiler's IEEE sequences carry
// operand scaling for the denormal range, a last correctly-rounding step and special-value fix-ups
// (17 instructions for sqrt, 11 + hazard no-ops for a division).  The values on this path are
// O(1e-6 .. 1e6), so the fast variant refines the seed directly:
//   msqrt : rsq, one coupled Newton step (g = sqrt, h = 1/(2 sqrt)), one residual correction
//           -> 8 instructions + zero/negative guard, error < 1 ulp
//   mrcp  : rcp, two Newton steps                              -> 5 instructions, error < 1 ulp
//   mdiv  : mrcp, quotient, one residual correction            -> 8 instructions, error < 1 ulp
// The strict variant is plain `sqrt` and `/` (correctly rounded).
#pragma once
#include "s5_config.hpp"

namespace S5NS {

S5_DEV bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// "if (S5_ANY(c)) { if (c) ... }" and "if (!S5_ANY(live)) break;": where a wave may skip work none of its lanes needs.  The
// plain divergent branch does that already -- s_and_saveexec sets the execution mask and s_cbranch_execz skips the block --
// on scalar instructions alone, while the vote of a flag that was not compared in the same basic block goes through a
// vector register and back (v_cndmask 0/1, v_cmp_ne: two vector instructions each; 180 of them in the pair kernel of
// rounds 1-3).  wave_any stays where a wave-uniform DECISION is needed (which instantiation a wave takes).
// (a translation unit may keep the explicit votes -- `#define S5_WAVE_VOTES 1` before its includes, in the SOURCE, not a build
// flag: k_surface.hip does, its walk kernel needs two registers fewer that way and sits on its cap of 168)
#ifdef S5_WAVE_VOTES
#define S5_ANY(c) wave_any(c)
#else
#define S5_ANY(c) (c)
#endif
#define S5_ANY_ORDER(c) S5_ANY(c)
#define S5_ANY_MISC(c) S5_ANY(c)

S5_DEV double sq(double x) { return x * x; }
// A product the back end must not fuse into the sum that follows (no instruction is emitted).  The march kernel's fast build
// is compiled with -ffp-contract=fast, which lets the back end fuse any a*b+c of the translation unit whatever a pragma says;
// the constants of motion and the operands of the polar range tests must carry the reference's own roundings in EVERY build
// (the last bit of q decides the class of a ray with l = 0: s5_geod.hpp).
S5_DEV double rounded_product(double x) { asm("" : "+v"(x)); return x; }
S5_DEV double max3abs(double a, double b, double c) { return fmax(fmax(fabs(a), fabs(b)), fabs(c)); }

// Horner step a*b + c as one instruction (the build runs with -ffp-contract=off; the approximation kernels
// of the fast variant ask for the fused form explicitly)
S5_DEV double hfma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// a*b + C for a literal constant C.  The compiler's own choice for this shape is v_fmac_f64 with C copied into
// the destination VGPR pair first -- two v_mov_b32 per coefficient, i.e. a Horner step costs three VALU slots.
// Written out as the three-address form with C in an SGPR pair (two s_mov_b32 on the scalar unit) it is one.
// Measured on the 4096^2 image: 1 694 -> see DESIGN.md VALU instructions per ray.
S5_DEV double hfmac(double a, double b, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}

#if S5_F_SQRTDIV

// sqrt for x known to be positive, finite and normal
S5_DEV double sqrt_pos(double x)
{
    // seed accurate to 2^-24.2 (measured on gfx950, tests/tools/seedacc.hip); one coupled Goldschmidt step takes g
    // to ~4e-15, the residual step to < 1 ulp.  h only scales the residual, so the seed's accuracy is enough.
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    const double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    const double d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

// sqrt(x) and 1/sqrt(x) together for positive normal x: 11 instructions (a sqrt plus a division would be 16)
S5_DEV void sqrt_rsqrt_pos(double x, double& s, double& rs)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    r = __builtin_fma(-h, g, 0.5);
    h = __builtin_fma(h, r, h);
    s = g;
    rs = h + h;
}

S5_DEV double rsqrt_pos(double x) { double s, rs; sqrt_rsqrt_pos(x, s, rs); return rs; }

// general sqrt: +-0 -> itself, negative / NaN -> NaN (as IEEE), +inf is not expected on this path
S5_DEV double msqrt(double x)
{
    const double g = sqrt_pos(x);
    return (x == 0.0) ? x : g;          // rsq(0) = inf makes g NaN; rsq(<0) = NaN propagates
}

S5_DEV double mrcp(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    return __builtin_fma(r, e, r);
}

S5_DEV double mdiv(double a, double b)
{
    // v_rcp_f64 is accurate to 2^-24.4 (measured); one Newton step gives 2e-15, and the residual step below
    // squares that once more -- a second Newton step on r would change nothing in q
    double r = __builtin_amdgcn_rcp(b);
    const double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = a * r;
    const double rem = __builtin_fma(-b, q, a);
    return __builtin_fma(rem, r, q);
}

#else

S5_DEV double sqrt_pos(double x) { return sqrt(x); }
S5_DEV void sqrt_rsqrt_pos(double x, double& s, double& rs) { s = sqrt(x); rs = 1.0 / s; }
S5_DEV double rsqrt_pos(double x) { return 1.0 / sqrt(x); }
S5_DEV double msqrt(double x) { return sqrt(x); }
S5_DEV double mrcp(double b) { return 1.0 / b; }
S5_DEV double mdiv(double a, double b) { return a / b; }

#endif

#if S5_F_LIBM
// x^(1/3) for x >= 0 (the reference writes pow(x, 1./3.); 1./3. is not exactly one third, the two
// differ by ln(x) * 1.85e-17 relative)
S5_DEV double mcbrt(double x) { return cbrt(x); }
// division by a compile-time constant: multiply by the folded reciprocal
#define S5_DIVC(a, c) ((a) * (1.0 / (c)))
#else
S5_DEV double mcbrt(double x) { return pow(x, 1. / 3.); }
#define S5_DIVC(a, c) ((a) / (c))
#endif

} // namespace S5NS
