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

// individual features of the fast variant (overridable for ablation builds, -DS5_F_xxx=0)
#ifndef S5_F_SQRTDIV
#define S5_F_SQRTDIV S5_FAST      // Newton-refined rsq/rcp sequences instead of IEEE sqrt and divide
#endif
#ifndef S5_F_RF7
#define S5_F_RF7 S5_FAST          // 7th-order Carlson series with Carlson's division-free stopping rule
#endif
#ifndef S5_F_AGMK
#define S5_F_AGMK S5_FAST         // K(m) by the arithmetic-geometric mean instead of R_F(0, 1-m, 1): -7 % time
#endif                            // in the fused image kernel (1.365 -> 1.27 ms), same results
#ifndef S5_F_LIBM
#define S5_F_LIBM S5_FAST         // cbrt for x^(1/3), fused sincos, folded constant reciprocals
#endif

#define S5_DEV __device__ __forceinline__
