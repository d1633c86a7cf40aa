// x87_check.cpp -- sim5_amd/csrc/s5_x87.hpp (integer emulation of the x87 double-extended operations) against the CPU's
// real `long double` on random operands, and polar_roots_x87 against the statement group it restates
// (ref: /root/reference/src/sim5kerr-geod.c:1125-1131) written out in long double.  Plain g++, x86-64 only; test
// infrastructure (tests/test_host_shim.py builds and runs it).  Prints "ok <n>" or the first mismatch.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <dlfcn.h>
#include "../../sim5_amd/csrc/s5_x87.hpp"

#if LDBL_MANT_DIG != 64
#error "this check needs the x87 80-bit long double"
#endif

using namespace s5x87;

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static double rnd_double(int spread)
{
    union { double d; uint64_t u; } c;
    const uint64_t frac = rnd() & 0xfffffffffffffull;
    const int ex = 1023 + (int)(rnd() % (2 * spread + 1)) - spread;
    c.u = ((rnd() & 1) << 63) | ((uint64_t)ex << 52) | frac;
    if (rnd() % 8 == 0) c.u &= ~((1ull << (rnd() % 52)) - 1);        // trailing zeros: exact cases and ties
    return c.d;
}
static long double to_ld(const X80& x)
{
    if (x.m == 0) return 0.0L;
    long double v = ldexpl((long double)x.m, x.e);
    return x.s ? -v : v;
}
static X80 from_ld(long double v)
{
    X80 r; r.s = v < 0; if (v < 0) v = -v;
    if (v == 0) { r.m = 0; r.e = 0; return r; }
    int e; long double f = frexpl(v, &e);                               // v = f 2^e, f in [0.5, 1)
    r.m = (uint64_t)ldexpl(f, 64); r.e = e - 64;
    return r;
}
static int fail(const char* what, long double a, long double b, long double got, long double want)
{
    printf("MISMATCH %s: a=%La b=%La got=%La want=%La\n", what, a, b, got, want);
    return 1;
}

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 2000000;
    long checked = 0;
    for (long i = 0; i < n; ++i) {
        const double a = rnd_double(40), b = rnd_double(i % 3 ? 40 : 3);
        // extended operands with all 64 bits in use: products of doubles
        const volatile long double ea = (long double)a * (long double)rnd_double(2), eb = (long double)b * (long double)rnd_double(2);
        const X80 xa = from_ld(ea), xb = from_ld(eb);
        if (to_ld(xa) != ea || to_ld(from_double(a)) != (long double)a) return fail("convert", ea, a, to_ld(xa), ea);
        volatile long double w;
        w = ea * eb;  if (to_ld(mul(xa, xb)) != w) return fail("mul", ea, eb, to_ld(mul(xa, xb)), w);
        w = ea + eb;  if (to_ld(add(xa, xb)) != w) return fail("add", ea, eb, to_ld(add(xa, xb)), w);
        w = ea - eb;  { X80 nb = xb; nb.s ^= 1; if (to_ld(add(xa, nb)) != w) return fail("sub", ea, eb, to_ld(add(xa, nb)), w); }
        w = ea / eb;  if (to_ld(div(xa, xb)) != w) return fail("div", ea, eb, to_ld(div(xa, xb)), w);
        w = (long double)a + (long double)b; if (to_ld(add(from_double(a), from_double(b))) != w) return fail("add53", a, b, 0, w);
        { volatile double d = (double)ea; if (to_double(xa) != d) return fail("to_double", ea, 0, to_double(xa), d); }
        // near-cancelling operands
        { const double c = -a * (1.0 + (double)(rnd() % 5) * DBL_EPSILON);
          w = (long double)a + (long double)c; if (to_ld(add(from_double(a), from_double(c))) != w) return fail("cancel", a, c, 0, w); }
        checked += 8;
    }
    // the statement group itself, on constants of motion of the kind the image kernels see (l = 0 column included)
    for (long i = 0; i < n / 4; ++i) {
        const double a = (i % 5 == 0) ? 1e-4 : 1e-4 + (rnd() % 1000000) * 9.99e-7;
        const double beta = ((double)(rnd() % 2000001) - 1e6) * 2e-5, alpha = (i % 2) ? 0.0 : ((double)(rnd() % 2000001) - 1e6) * 2e-5;
        const double ci = (double)(rnd() % 1000000) * 1e-6, si = sqrt(1 - ci * ci);
        const double l = -alpha * si, q = beta * beta + ci * ci * (alpha * alpha - a * a);
        if (q == 0.0) continue;
        const double a2 = a * a, l2 = l * l;
        volatile long double qla = q + l2 - a2;
        volatile long double X = sqrt((double)(qla * qla + (long double)(4. * q * a2))) + qla;     // sqrt(double) of the narrowed sum (no tgmath in the reference)
        volatile long double dbla = a2 + a2, dblq = q + q;
        volatile double m2m = X / dbla, m2p = dblq / X;
        double e_m2m = -1, e_m2p = -1;
        const bool okx = polar_roots_x87(q + l2 - a2, 4. * q * a2, a2 + a2, q + q, e_m2m, e_m2p);
        const bool finite = (double)(qla * qla + (long double)(4. * q * a2)) >= 0.0;
        if (okx != finite) { printf("MISMATCH roots: fallback decision a=%.17g q=%.17g l=%.17g\n", a, q, l); return 1; }
        if (okx && (memcmp((void*)&m2m, &e_m2m, 8) || memcmp((void*)&m2p, &e_m2p, 8))) {
            printf("MISMATCH roots: a=%.17g q=%.17g l=%.17g m2m %.17g / %.17g m2p %.17g / %.17g\n", a, q, l, (double)m2m, e_m2m, (double)m2p, e_m2p);
            return 1;
        }
        checked += 1;
    }
    // against the unmodified reference itself (oracle/_ref/libsim5ref.so, when given): m2m, m2p of geodesic_init_inf's record
    if (argc > 2) {
        void* h = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
        if (!h) { printf("cannot load %s\n", argv[2]); return 2; }
        typedef int (*init_fn)(double, double, double, double, void*, int*);
        init_fn init = (init_fn)dlsym(h, "geodesic_init_inf");
        if (!init) { printf("no geodesic_init_inf\n"); return 2; }
        long ran = 0, decided_by_rounding = 0;
        for (long i = 0; i < n / 4; ++i) {
            const double a = (i % 7 == 0) ? 0.0 : (double)(rnd() % 999999) * 1e-6;
            const double inc = 0.02 + (double)(rnd() % 1000000) * 1.5e-6;
            const double beta = (i % 11 == 0) ? 0.0 : ((double)(rnd() % 2000001) - 1e6) * 2e-5;
            const double alpha = (i % 2) ? 0.0 : ((double)(rnd() % 2000001) - 1e6) * 2e-5;
            double g[30]; int err = -1;
            init(inc, a, alpha, beta, g, &err);
            if (!(err == 0 || err == 8 || err == 9 || err == 10)) continue;           // the polar roots were reached
            const double b = (beta == 0.0) ? 1e-6 : beta, ac = fmax(1e-4, a);
            const double l = -alpha * sin(inc), q = b * b + cos(inc) * cos(inc) * (alpha * alpha - a * a);
            if (memcmp(&l, &g[5], 8) || memcmp(&q, &g[6], 8)) { printf("MISMATCH l, q of the reference\n"); return 1; }
            const double a2 = ac * ac, l2 = l * l;
            double m2m = -1, m2p = -1;
            if (!polar_roots_x87(q + l2 - a2, 4. * q * a2, a2 + a2, q + q, m2m, m2p)) continue;
            if (memcmp(&m2p, &g[16], 8) || memcmp(&m2m, &g[17], 8)) {
                printf("MISMATCH reference: i=%.17g a=%.17g alpha=%.17g beta=%.17g m2p %.17g / %.17g m2m %.17g / %.17g\n", inc, a, alpha, beta, g[16], m2p, g[17], m2m);
                return 1;
            }
            const double X = sqrt((q + l2 - a2) * (q + l2 - a2) + 4. * q * a2) + (q + l2 - a2);
            if (((q + q) / X >= 1.0) != (m2p >= 1.0)) ++decided_by_rounding;
            ++ran;
        }
        printf("reference records %ld (range test m2p >= 1 decided differently by plain double on %ld)\n", ran, decided_by_rounding);
    }
    printf("ok %ld\n", checked);
    return 0;
}
