// s5_config.hpp -- build variant of the device math.
//
// Every device header is compiled in one of two variants, selected per translation unit:
//
//   S5_FAST=0  ("strict", namespace s5):  the reference's parameters and operation order, IEEE
//              correctly rounded sqrt and division, no FMA contraction (-ffp-contract=off).  Used by
//              the batch forms of the per-ray API, the step-wise integrator, and by the image kernels
//              when the job asks for SIM5GPU_IMG_STRICT.
//   S5_FAST=1  ("fast", namespace s5f):   same algorithms tuned for the FP64 VALU -- Newton-refined
//              v_rsq_f64 / v_rcp_f64 without the denormal-range scaling and special-value fix-ups, a
//              7th-order Carlson series with Carlson's division-free stopping rule (~4 passes instead of
//              ~6.5), cbrt, bounded-range sincos/acos/log, K(m) by the AGM, folded constant reciprocals.  Still no FMA contraction (measured:
//              no speed-up, 1e5x larger worst-pixel error).  Every primitive stays within 1 ulp; against
//              the reference r and g agree to < 1e-12 on the headline image (bar: 1e-6) and the hit/miss
//              class maps are identical on every golden image.  Ablation on MI355X, 4096^2 headline image:
//              all on 1.71 ms; without the sqrt/div sequences 2.08 ms; without the 7th-order series
//              2.23 ms; without cbrt/sincos 1.88 ms; strict 2.85 ms.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>

#ifndef S5_FAST
#define S5_FAST 0
#endif

#if S5_FAST
#define S5NS s5f
#else
#define S5NS s5
#endif

// individual features of the fast variant (what the ablation builds of rounds 1-3 switched off one at a time: header comment)
#define S5_F_SQRTDIV S5_FAST      // Newton-refined rsq/rcp sequences instead of IEEE sqrt and divide
#define S5_F_RF7 S5_FAST          // 7th-order Carlson series with Carlson's division-free stopping rule
#define S5_F_AGMK S5_FAST         // K(m) by the arithmetic-geometric mean instead of R_F(0, 1-m, 1): -7 % time
#define S5_F_LIBM S5_FAST         // cbrt for x^(1/3), fused sincos, folded constant reciprocals

// Selective FMA contraction in the fast variant.  The translation units are compiled with -ffp-contract=off (the
// reference build has no FMA); a region named below lets the compiler fuse a*b+c when its bit is set in S5_FPC_MASK.
//   1 quartic (constants of motion, closed-form roots)     2 class set-up, polar roots, crossing logic
//   4 the quotients of r(P)                                8 Carlson R_F loop and series
//  16 Landen ladder (climb, descent)                      32 g-factor and flux table
// Bisected on MI355X against the CPU checker, 4096^2 headline image, one region at a time (round 3): regions
// 2 .. 32 leave the worst pixel where it was (r 3e-13 .. 7e-13, g 7e-13, flux 6e-8; class map identical), region 1 alone
// takes it to r 1.4e-7, g 7e-7, flux 2.3e-5 -- the discriminant X = F^2 - 4 E^3 of the resolvent cubic cancels to rounding
// noise near the double-root locus, and only the reference's own two roundings (F^2 rounded, then the difference) put the
// same rays on the same side of X = 0.  So everything but the quartic contracts: 0.377 -> 0.366 ms on the headline image.
#define S5_FPC_MASK (S5_FAST ? 62 : 0)
// contract(on), not (fast): a*b+c is fused only inside ONE source expression, decided by the front end (llvm.fmuladd, always
// v_fma_f64 on gfx950), so a routine rounds the same way in every kernel and template instantiation it is inlined into.
// With (fast) the back end fuses across statements where it finds it profitable, which differs from one instantiation
// to the next: the pairing kernel and the plain kernel then disagree in the last bit (caught by
// tests/test_gpu_images.py::test_mirrored_pairs_give_the_plain_image).
#define S5_FPC_PRAGMA_ON  _Pragma("clang fp contract(on)")
#define S5_FPC_PRAGMA_OFF _Pragma("clang fp contract(off)")
#if S5_FAST && (S5_FPC_MASK & 1)
#define S5_FPC_QUARTIC S5_FPC_PRAGMA_ON
#else
#define S5_FPC_QUARTIC
#endif
#if S5_FAST && (S5_FPC_MASK & 2)
#define S5_FPC_FINISH S5_FPC_PRAGMA_ON
#else
#define S5_FPC_FINISH
#endif
#if S5_FAST && (S5_FPC_MASK & 4)
#define S5_FPC_RADIUS S5_FPC_PRAGMA_ON
#else
#define S5_FPC_RADIUS S5_FPC_PRAGMA_OFF
#endif
#if S5_FAST && (S5_FPC_MASK & 8)
#define S5_FPC_RF S5_FPC_PRAGMA_ON
#else
#define S5_FPC_RF
#endif
#if S5_FAST && (S5_FPC_MASK & 16)
#define S5_FPC_LADDER S5_FPC_PRAGMA_ON
#else
#define S5_FPC_LADDER
#endif
#if S5_FAST && (S5_FPC_MASK & 32)
#define S5_FPC_GFLUX S5_FPC_PRAGMA_ON
#else
#define S5_FPC_GFLUX
#endif

// fast variant: the image kernels' rays do not evaluate the radial integral Rpc; r(P) comes from the addition theorem
// (s5_thindisk.hpp).
#if S5_FAST
#define S5_RPC_ADD 1
#else
#define S5_RPC_ADD 0
#endif

#define S5_DEV __device__ __forceinline__
// a comment in the generated assembly (to find a source region in the ISA; no instruction)
#ifdef S5_ISA_MARKS
#define S5_MARK(text) asm volatile("; S5MARK " text)
#else
#define S5_MARK(text) do {} while (0)
#endif
// constant address space (kernel-argument segment, read-only tables): loads through such a pointer are scalar loads
#define S5_AS4 __attribute__((address_space(4)))

// A floating-point literal as a value in SCALAR registers.  A 64-bit literal cannot be an operand of a vector instruction; left
// to itself the compiler often materialises it in a vector register pair right before its use (two v_mov_b32, then v_fmac with
// the constant as the tied accumulator): three vector issue slots for one fused multiply-add in a Horner step, in kernels that
// are bound by vector issue.  Through this (pure, CSE-able) asm the constant is two s_mov_b32 -- the scalar unit, which has slack
// -- and the multiply-add reads it as its one scalar operand.  The VALUE is the literal: results are the same bits.
__device__ __forceinline__ double sconst(double c)
{
    asm("" : "+s"(c));
    return c;
}

// Parameters behind a constant-address-space reference: the same object through a pointer the optimiser cannot see through,
// so the loads that follow are issued from here on (not hoisted to the kernel's head and held -- spilled -- in SGPRs).  A
// by-value argument block passes through unchanged.
template <class T> S5_DEV const T& param_reload(const T& p) { return p; }
template <class T> S5_DEV const S5_AS4 T& param_reload(const S5_AS4 T& p)
{
    const S5_AS4 T* q = &p;
    asm volatile("" : "+s"(q));
    return *q;
}
