// s5_x87.hpp -- the x87 double-extended arithmetic of ONE statement group of the reference, bit for bit.
//
// The reference's host build carries the polar roots through `long double` (ref: /root/reference/src/sim5kerr-geod.c:1125-1131):
// with gcc on x86-64 that is the 80-bit x87 format -- a 64-bit significand, round to nearest even -- and every result is
// rounded a SECOND time when it is stored to a double.  For a ray with l = 0 (central column of an odd-width image) the
// outer root m2p equals 1 exactly in real arithmetic, so the reference's range test `m2p >= 1.0` (ref :1140) is decided by
// those very roundings; for beta = 0 -> 1e-6 (central row) `|cos i| > sqrt(m2p)` (ref :1153) is.  A double-only evaluation
// (the reference's own CUDA branch, ref :1133-1138) lands on the other side for a quarter of such pixels.  So the sequence is
// reproduced here in integer arithmetic: products, sums and quotients of 64-bit significands with one rounding to 64 bits,
// then one to 53.  What gcc 11 -O3 emits for the statement group (checked in the disassembly of the reference library built here:
// fmul / faddl / fstpl / sqrtsd / faddl / fdivl / fdivrl) is restated in `polar_roots_x87` below.
//
// No device float instruction takes part, so host (g++: tests/c/x87_check.cpp compares every operation with the CPU's real
// long double on random operands) and device give the same bits.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define S5X_FN __host__ __device__ inline
#else
#define S5X_FN inline
#endif

namespace s5x87 {

// value = (-1)^s * m * 2^e; m has bit 63 set unless the value is zero (m == 0)
struct X80 { uint64_t m; int e; int s; };

S5X_FN int clz64(uint64_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)v);
#else
    return __builtin_clzll(v);
#endif
}

S5X_FN void mul64x64(uint64_t a, uint64_t b, uint64_t& hi, uint64_t& lo)
{
#if defined(__HIP_DEVICE_COMPILE__)
    hi = __umul64hi(a, b);
    lo = a * b;
#else
    const unsigned __int128 p = (unsigned __int128)a * b;
    hi = (uint64_t)(p >> 64);
    lo = (uint64_t)p;
#endif
}

// a finite, non-zero double (normal or subnormal) as an X80, exactly
S5X_FN X80 from_double(double d)
{
    union { double d; uint64_t u; } c; c.d = d;
    X80 r;
    r.s = (int)(c.u >> 63);
    const int ex = (int)((c.u >> 52) & 0x7ff);
    uint64_t f = c.u & 0xfffffffffffffull;
    if (ex == 0) {
        if (f == 0) { r.m = 0; r.e = 0; return r; }
        const int sh = clz64(f);
        r.m = f << sh;
        r.e = -1074 - sh;
    } else {
        r.m = (f | (1ull << 52)) << 11;
        r.e = ex - 1075 - 11;
    }
    return r;
}

// (h * 2^64 + l) * 2^e, plus "some more, less than one unit of l" when sticky: normalised and rounded to 64 bits, ties to even
S5X_FN X80 round_pack(int s, uint64_t h, uint64_t l, int e, bool sticky)
{
    X80 r; r.s = s;
    if (h == 0 && l == 0) { r.m = 0; r.e = 0; return r; }            // (sticky alone cannot occur: callers keep >= 2 guard bits)
    if (h == 0) { h = l; l = 0; e -= 64; }
    const int sh = clz64(h);
    if (sh) { h = (h << sh) | (l >> (64 - sh)); l <<= sh; e -= sh; }
    const bool guard = (l >> 63) != 0;
    const bool rest = ((l << 1) != 0) || sticky;
    if (guard && (rest || (h & 1))) {
        h += 1;
        if (h == 0) { h = 1ull << 63; e += 1; }
    }
    r.m = h; r.e = e + 64;
    return r;
}

S5X_FN X80 mul(const X80& a, const X80& b)
{
    uint64_t h, l;
    mul64x64(a.m, b.m, h, l);
    return round_pack(a.s ^ b.s, h, l, a.e + b.e, false);
}

S5X_FN X80 add(X80 a, X80 b)
{
    if (a.m == 0) return b;
    if (b.m == 0) return a;
    if (a.e < b.e || (a.e == b.e && a.m < b.m)) { const X80 t = a; a = b; b = t; }     // |a| >= |b| from here on

    const int E = a.e - 63;
    const uint64_t ah = a.m >> 1, al = a.m << 63;
    uint64_t bh = b.m >> 1, bl = b.m << 63;
    bool sticky = false;
    const int d = a.e - b.e;
    if (d >= 128) { bh = 0; bl = 0; sticky = true; }
    else if (d >= 64) { sticky = (bl != 0) || (d > 64 && (bh << (128 - d)) != 0); bl = (d == 64) ? bh : (bh >> (d - 64)); bh = 0; }
    else if (d > 0) { sticky = (bl << (64 - d)) != 0; bl = (bl >> d) | (bh << (64 - d)); bh >>= d; }
    uint64_t h, l;
    if (a.s == b.s) {
        l = al + bl;
        h = ah + bh + (l < al ? 1 : 0);
    } else {
        l = al - bl;
        h = ah - bh - (al < bl ? 1 : 0);
        if (sticky) {                                   // the exact value lies just under: borrow one unit, keep the rest as sticky
            if (l == 0) h -= 1;
            l -= 1;
        }
    }
    X80 r = round_pack(a.s, h, l, E, sticky);
    if (r.m == 0) r.s = 0;                              // x - x = +0 when rounding to nearest
    return r;
}

S5X_FN X80 div(const X80& n, const X80& dd)
{
    // 66 quotient bits of n.m / dd.m by shift and subtract, the remainder as sticky
    uint64_t rem = n.m, qh = 0, ql = 0;
    bool carry = false;
    for (int i = 0; i < 66; ++i) {
        const bool ge = carry || rem >= dd.m;
        if (ge) rem -= dd.m;
        qh = (qh << 1) | (ql >> 63);
        ql = (ql << 1) | (ge ? 1 : 0);
        carry = (rem >> 63) != 0;
        rem <<= 1;
    }
    return round_pack(n.s ^ dd.s, qh, ql, n.e - dd.e - 65, carry || rem != 0);
}

// the second rounding: 64 -> 53 bits, ties to even (what `fstpl` does).  Results outside the normal range of a double do not
// occur for the operands this header is used on (the caller falls back to plain double arithmetic for such input).
S5X_FN double to_double(const X80& x)
{
    if (x.m == 0) return x.s ? -0.0 : 0.0;
    uint64_t m = x.m >> 11;
    const uint64_t low = x.m & 0x7ff;
    int e = x.e + 11;
    if (low > 0x400 || (low == 0x400 && (m & 1))) {
        m += 1;
        if (m == (1ull << 53)) { m >>= 1; e += 1; }
    }
    const double v = ldexp((double)m, e);
    return x.s ? -v : v;
}

// m2m and m2p of the polar potential as the reference's host build forms them (ref :1125-1131), from the doubles
// qla = q + l^2 - a^2, c4 = 4 q a^2, a^2 + a^2 and q + q that its SSE code hands to the x87 unit:
//     fld qla; fmul st0,st0; faddl c4; fstpl -> double; sqrtsd; faddl (qla still on the stack) = X
//     X / dbla -> fstpl m2m;   dblq / X -> fstpl m2p
// Returns false (nothing written) when an operand is zero, not finite, or the radicand negative: the caller's plain double
// sequence then gives the reference's NaN / infinity behaviour.
struct PolarM2 { double m2m, m2p; int ok; };
#if defined(__HIP_DEVICE_COMPILE__)
// ONE copy per code object, called: the routine is ~1 300 instructions and sits on cold paths only (three inlined copies in
// the image kernels' direct routine cost the production kernel 0.6 % through code size alone).  Arguments and result by
// value, in registers: a result through references would put the caller's variables on a stack.
__device__ __attribute__((noinline))
#else
inline
#endif
PolarM2 polar_roots_x87_value(double qla, double c4, double dbla, double dblq)
{
    PolarM2 out = {0.0, 0.0, 0};
    if (!(fabs(qla) < 1e150 && fabs(c4) < 1e300 && fabs(dbla) < 1e300 && fabs(dblq) < 1e300)) return out;
    if (dbla == 0.0 || dblq == 0.0 || (qla != 0.0 && fabs(qla) < 1e-150) || fabs(dbla) < 1e-300 || fabs(dblq) < 1e-300) return out;
    const X80 xq = from_double(qla);
    const double rad = to_double(add(mul(xq, xq), from_double(c4)));
    if (!(rad >= 0.0)) return out;
    const X80 X = add(xq, from_double(sqrt(rad)));
    if (X.m == 0) return out;
    out.m2m = to_double(div(X, from_double(dbla)));
    out.m2p = to_double(div(from_double(dblq), X));
    out.ok = 1;
    return out;
}

S5X_FN bool polar_roots_x87(double qla, double c4, double dbla, double dblq, double& m2m, double& m2p)
{
    const PolarM2 r = polar_roots_x87_value(qla, c4, dbla, dblq);
    if (!r.ok) return false;
    m2m = r.m2m; m2p = r.m2p;
    return true;
}

}  // namespace s5x87
