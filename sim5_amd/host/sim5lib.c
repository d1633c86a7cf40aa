/*
 * sim5lib.c -- host side of the SIM5 scalar API over libsim5gpu.so (see sim5lib.h).
 *
 * Plain C (builds with the reference's own example Makefile flags: gcc -O3 -w -fgnu89-inline ... -lm).
 * Each SIM5 function = one batch call with n = 1 through function pointers resolved by dlopen/dlsym
 * on first use (glibc >= 2.34 has dlopen in libc, so no -ldl is needed).  No ray arithmetic happens in
 * this file.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stddef.h>
#include "sim5lib.h"

#ifndef SIM5GPU_LIB_DEFAULT
#define SIM5GPU_LIB_DEFAULT ""
#endif

static void *s5_handle;

static void s5_die(const char *what)
{
    /* same stderr convention as the reference's error(), but a missing GPU path is fatal */
    fprintf(stderr, "ERROR: sim5lib (MI355X): %s\n", what);
    exit(EXIT_FAILURE);
}

static pthread_once_t s5_handle_once = PTHREAD_ONCE_INIT;
static void s5_load(void)
{
    {
        const char *env = getenv("SIM5GPU_LIB");
        char fromfile[4096];
        const char *cand[4];
        int n = 0;
        if (env && *env) cand[n++] = env;
        cand[n++] = "libsim5gpu.so";
        if (SIM5GPU_LIB_DEFAULT[0]) cand[n++] = SIM5GPU_LIB_DEFAULT;
        /* in-tree build next to this source file, when __FILE__ carries a directory */
        {
            const char *f = __FILE__;
            const char *slash = strrchr(f, '/');
            if (slash && (size_t)(slash - f) + 32 < sizeof fromfile) {
                memcpy(fromfile, f, (size_t)(slash - f));
                strcpy(fromfile + (slash - f), "/../lib/libsim5gpu.so");
                cand[n++] = fromfile;
            }
        }
        for (int i = 0; i < n && !s5_handle; i++) s5_handle = dlopen(cand[i], RTLD_NOW | RTLD_LOCAL);
        if (!s5_handle) s5_die("cannot load libsim5gpu.so (set SIM5GPU_LIB); there is no CPU fallback");
    }
}

static void *s5_sym(const char *name)
{
    pthread_once(&s5_handle_once, s5_load);         /* the library is opened once, whichever thread comes first */
    void *p = dlsym(s5_handle, name);
    if (!p) s5_die(name);
    return p;
}

/* A failed call: the reference's convention is error() -- print "ERROR: ..." to stderr and return (ref src/sim5utils.c:41-54);
 * nothing exits, the caller gets the NaN / FALSE the wrapper initialised its result with.  Only a machine without a usable
 * GPU ends the program: there is no CPU path behind this API, and every later call would fail the same way. */
static int s5_check(int rc, const char *fn)
{
    if (rc != 0) {
        const char *(*last)(void) = (const char *(*)(void))s5_sym("sim5gpu_last_error");
        fprintf(stderr, "ERROR: sim5lib (MI355X): %s failed (%d): %s\n", fn, rc, last());
        if (rc == -1) exit(EXIT_FAILURE);               /* SIM5GPU_E_NO_DEVICE */
    }
    return rc;
}

/* resolve once per call site (threads may race to resolve the same pointer: the slot is read and written atomically, and
 * every racer writes the same value) */
#define S5_FN(type, var, name) static type var##_slot; type var = __atomic_load_n(&var##_slot, __ATOMIC_ACQUIRE); \
    if (!var) { var = (type)s5_sym(name); __atomic_store_n(&var##_slot, var, __ATOMIC_RELEASE); }

typedef int (*fn_geod_init_inf)(size_t, const double *, const double *, const double *, const double *, geodesic *, int *, int *);
typedef int (*fn_geod_init_src)(size_t, const double *, const double *, const double *, const double *, const int *, geodesic *, int *, int *);
typedef int (*fn_geod_P)(size_t, const geodesic *, const double *, double *);
typedef int (*fn_geod_order)(size_t, const geodesic *, const int *, double *);
typedef int (*fn_geod_Pint)(size_t, const geodesic *, const double *, const int *, double *);
typedef int (*fn_geod_mom)(size_t, const geodesic *, const double *, const double *, const double *, double *);
typedef int (*fn_geod_azm)(size_t, const geodesic *, const double *, const double *, const double *, double *);
typedef int (*fn_geod_delay)(size_t, const geodesic *, const double *, const double *, const double *, const double *, const double *, const double *, double *);
typedef int (*fn_geod_follow)(size_t, const geodesic *, const double *, double *, double *, double *, int *);
typedef int (*fn_d1)(size_t, const double *, double *);
typedef int (*fn_d2)(size_t, const double *, const double *, double *);
typedef int (*fn_d3)(size_t, const double *, const double *, const double *, double *);

/* One round trip per ray for the caller loop of ref examples/04-disk-image-eqplane/disk-image.c:62-100.  geodesic_init_inf
 * asks the GPU for the geodesic AND for what that loop asks next (sim5gpu_geodesic_init_inf_chain: crossings of orders 0 and
 * 1, the radii there, gfactorK and disk_nt_flux at those radii -- each by the device routine of the single call).  The record
 * is kept per thread next to a copy of the geodesic; geodesic_find_midplane_crossing, geodesic_position_rad, gfactorK and
 * disk_nt_flux answer from it when -- and only when -- their arguments are bit for bit the ones the record was made for (the
 * whole 240-byte struct, P, r, a, l, and the same disk set-up); any other call goes to the GPU as before.  The record is made
 * by the STRICT routines -- the ones the single calls run -- so a function's value never depends on whether a record
 * answered it (round 4 made it in the fast arithmetic by default).  SIM5_SHIM_NO_CHAIN=1 switches the record off
 * (every call a round trip: the tests compare the two).
 *
 * LOOK-AHEAD (round 5).  Both callers of the reference walk an image in raster order (disk-image.c:53-58,
 * python/sim5diskraytrace.py:163-165): a row is one (i, a, beta) with the SAME sequence of alpha values as the row before.
 * So the shim remembers the alphas of the row it is being shown; when a new row starts with the first alpha of the previous
 * one, it asks for the records of the WHOLE row -- the remembered alphas, bit for bit, with the new beta -- in ONE batch call,
 * and answers the following geodesic_init_inf calls from them: each only after its four arguments have been compared, bit for
 * bit, with the ones its record was made for.  The first call that does not match (another order, another image) drops the
 * remaining records and goes to the GPU on its own, as before; nothing is ever answered from a record made for other
 * arguments, so a caller in any order gets the values of the single calls.  One launch per row instead of one per ray.
 * SIM5_SHIM_NO_LOOKAHEAD=1 switches it off. */
typedef struct {
    double P[2], r[2], g[2], flux[2];
    double a, l;
    int have_r[2];
    int valid, flux_valid;
} s5_chain;
typedef int (*fn_geod_chain)(size_t, const double *, const double *, const double *, const double *, geodesic *, int *, int *, s5_chain *);
static __thread struct { int live; unsigned long disk_gen; geodesic g; s5_chain c; } s5_last;
/* Which disk model a record's flux belongs to: the generation counter lives INSIDE libsim5gpu (bumped by every successful
 * sim5gpu_disk_nt_setup, whoever calls it -- this shim, sim5_amd/capi.py, DiskModel_ThinDisk, a C program beside the shim:
 * ADVICE r5), and a record carries the value read BEFORE the batch call that made it. */
static unsigned long s5_disk_generation(void)
{
    typedef unsigned long (*fn)(void);
    S5_FN(fn, f, "sim5gpu_disk_nt_generation");
    return f();
}
/* record / look-ahead switches: read from the environment ONCE, both before any thread can see either (pthread_once) */
static int s5_chain_mode = 2;                     /* 0 off, 2 on (strict routines) */
static int s5_lookahead = 1;                      /* 0 off, 1 on */
static pthread_once_t s5_modes_once = PTHREAD_ONCE_INIT;
static void s5_modes_init(void)
{
    const char *e = getenv("SIM5_SHIM_NO_CHAIN"), *la = getenv("SIM5_SHIM_NO_LOOKAHEAD");
    const int chain = (e && *e && *e != '0') ? 0 : 2;
    s5_lookahead = (chain && !(la && *la && *la != '0')) ? 1 : 0;
    s5_chain_mode = chain;
}

static int s5_same_bits(double x, double y) { return memcmp(&x, &y, sizeof x) == 0; }
static int s5_record_for(const geodesic *g) { return s5_last.live && memcmp(g, &s5_last.g, sizeof *g) == 0; }

#define S5_ROW_MAX 16384                          /* longest row remembered */
#define S5_AHEAD_MAX 32768                        /* rays asked for in one look-ahead call (11 MB of records) */
typedef struct {
    /* the row being shown */
    double i, a, beta;
    double *alpha; int n, cap;
    /* the row before it, complete */
    double *tmpl; int tmpl_n, tmpl_cap;
    double tmpl_i, tmpl_a, tmpl_beta;
    int rows_seen;                 /* complete rows of this (i, a) with this alpha sequence, the template's included */
    /* the callers' pixel formula, once its parameters reproduce what was shown bit for bit (s5_fit_*): alpha_x =
     * ((x + .5) / nx - .5) * 2 * R and beta_y = ((y + .5) / ny - .5) * 2 * R [* (ny / nx)]  (ref disk-image.c:57-58,
     * python/sim5diskraytrace.py:163-164); 0 = not known */
    int fit_nx, fit_ny, fit_scaled, fit_tried;
    double fit_R;
    /* records made ahead: entry k was made for (spec_i, spec_a, spec_alpha[k], spec_beta[k]); `cursor` = the entry the next
     * call should ask for */
    int spec_n, cursor, spec_cap;
    double spec_i, spec_a;
    double *spec_alpha, *spec_beta, *arg_i, *arg_a;
    geodesic *g; int *err, *ok; s5_chain *c;
    unsigned long disk_gen;
} s5_ahead;
static __thread s5_ahead s5_la;
/* the buffers of a thread's look-ahead (up to ~11 MB) go when the thread does */
static pthread_key_t s5_la_key;
static pthread_once_t s5_la_key_once = PTHREAD_ONCE_INIT;
static void s5_la_free(void *p)
{
    s5_ahead *L = (s5_ahead *)p;
    if (!L) return;
    free(L->alpha); free(L->tmpl); free(L->spec_alpha); free(L->spec_beta); free(L->arg_i); free(L->arg_a);
    free(L->g); free(L->err); free(L->ok); free(L->c);
    memset(L, 0, sizeof *L);
}
static void s5_la_key_make(void) { (void)pthread_key_create(&s5_la_key, s5_la_free); }
static void s5_la_register(void)
{
    static __thread int done;
    if (done) return;
    done = 1;
    pthread_once(&s5_la_key_once, s5_la_key_make);
    (void)pthread_setspecific(s5_la_key, &s5_la);     /* (the destructor runs while the thread's TLS block is still there) */
}

static int s5_grow(double **p, int *cap, int need)
{
    if (need <= *cap) return 1;
    int nc = *cap ? *cap : 256;
    while (nc < need) nc *= 2;
    double *q = (double *)realloc(*p, (size_t)nc * sizeof(double));
    if (!q) return 0;
    *p = q; *cap = nc;
    return 1;
}

/* the callers' expression for a pixel coordinate, operation for operation (volatile: no contraction, no excess precision) */
static double s5_pixel(int x, int n, double R, double scale)
{
    volatile double t = ((double)x + .5) / (double)n;
    t = t - 0.5;
    t = t * 2.0;
    t = t * R;
    if (scale != 1.0) t = t * scale;
    return t;
}

/* n and R of the formula from its first two values v0, v1 (pixels 0 and 1), or 0: tried with R a few units in the last
 * place around v0 / t0, accepted only if BOTH values come out bit for bit */
static int s5_fit_axis(double v0, double v1, double scale, double R_hint, double *R_out)
{
    const double d = v1 - v0;
    if (!(d > 0.0) || !(v0 < 0.0)) return 0;
    const double nf = 1.0 - 2.0 * v0 / d;
    if (!(nf >= 2.0 && nf <= 1e6)) return 0;
    const int n = (int)(nf + 0.5);
    if (fabs(nf - (double)n) > 1e-6 * nf) return 0;
    double R0 = R_hint;
    if (!(R0 > 0.0)) R0 = v0 / ((((double)0 + .5) / (double)n - 0.5) * 2.0 * scale);
    double R = R0;
    for (int k = 0; k < 9; k++) {                    /* R0, then its neighbours alternately up and down */
        if (R > 0.0 && s5_same_bits(s5_pixel(0, n, R, scale), v0) && s5_same_bits(s5_pixel(1, n, R, scale), v1)) { *R_out = R; return n; }
        if (R_hint > 0.0) break;                      /* the other axis fixed R already */
        R = R0;
        for (int j = 0; j <= k / 2; j++) R = nextafter(R, (k & 1) ? 0.0 : INFINITY);
    }
    return 0;
}

static void s5_forget_fit(void) { s5_la.fit_nx = s5_la.fit_ny = s5_la.fit_scaled = s5_la.fit_tried = 0; s5_la.fit_R = 0.0; }

/* The caller shows (i, a, alpha, beta): keep the book of rows.  Returns what could be made ahead now:
 *   1  this call opens a row that repeats the remembered one (same i, a, first alpha)
 *   2  this is the second call of the FIRST row of an image and the pixel formula reproduces both alphas: the rest of the row
 *   0  nothing */
static int s5_row_note(double i, double a, double alpha, double beta)
{
    s5_ahead *L = &s5_la;
    if (L->n > 0 && s5_same_bits(i, L->i) && s5_same_bits(a, L->a) && s5_same_bits(beta, L->beta)) {
        if (L->n < S5_ROW_MAX && s5_grow(&L->alpha, &L->cap, L->n + 1)) L->alpha[L->n++] = alpha;
        else L->n = S5_ROW_MAX + 1;                           /* too long to remember: never becomes a template */
        if (L->n == 2 && L->tmpl_n == 0 && !L->fit_tried) {
            L->fit_tried = 1;
            L->fit_nx = s5_fit_axis(L->alpha[0], L->alpha[1], 1.0, 0.0, &L->fit_R);
            if (L->fit_nx > 2 && L->fit_nx <= S5_ROW_MAX) return 2;
            L->fit_nx = 0;
        }
        return 0;
    }
    /* a new row: the finished one becomes the template if it was a row at all */
    const int same_image = L->n >= 2 && L->n <= S5_ROW_MAX && s5_same_bits(i, L->i) && s5_same_bits(a, L->a);
    const int repeats = same_image && L->tmpl_n == L->n && memcmp(L->tmpl, L->alpha, (size_t)L->n * sizeof(double)) == 0;
    if (same_image) {
        const double beta_prev = L->beta;
        double *t = L->tmpl; int tc = L->tmpl_cap;
        L->tmpl = L->alpha; L->tmpl_cap = L->cap; L->tmpl_n = L->n; L->tmpl_i = L->i; L->tmpl_a = L->a;
        L->alpha = t; L->cap = tc;
        L->rows_seen = repeats ? L->rows_seen + 1 : 1;
        /* two complete-or-begun rows: the vertical axis of the formula (needs the horizontal one: R and nx) */
        if (L->rows_seen == 1 && L->fit_nx == L->tmpl_n && L->fit_ny == 0) {
            double R = 0.0;
            int ny = s5_fit_axis(beta_prev, beta, 1.0, L->fit_R, &R);                       /* python caller: square image */
            if (ny > 1) { L->fit_ny = ny; L->fit_scaled = 0; }
            else {
                /* ref disk-image.c:58: ... * 2 * rmax * ((double)ny / (double)nx) */
                const double d = beta - beta_prev;
                const double nf = (d > 0.0) ? 1.0 - 2.0 * beta_prev / d : 0.0;
                const int nyc = (nf >= 2.0 && nf <= 1e6) ? (int)(nf + 0.5) : 0;
                if (nyc > 1) {
                    const double sc = (double)nyc / (double)L->fit_nx;
                    if (s5_same_bits(s5_pixel(0, nyc, L->fit_R, sc), beta_prev) && s5_same_bits(s5_pixel(1, nyc, L->fit_R, sc), beta)) { L->fit_ny = nyc; L->fit_scaled = 1; }
                }
            }
        }
        L->tmpl_beta = beta_prev;
    } else { L->tmpl_n = 0; L->rows_seen = 0; s5_forget_fit(); }
    L->i = i; L->a = a; L->beta = beta; L->n = 0;
    if (s5_grow(&L->alpha, &L->cap, 1)) L->alpha[L->n++] = alpha;
    return (L->tmpl_n >= 2 && s5_same_bits(alpha, L->tmpl[0])) ? 1 : 0;
}

static int s5_spec_room(int n)
{
    s5_ahead *L = &s5_la;
    if (n <= L->spec_cap) return 1;
    int nc = L->spec_cap ? L->spec_cap : 256;
    while (nc < n) nc *= 2;
    double *sa = (double *)realloc(L->spec_alpha, (size_t)nc * sizeof(double));
    if (sa) L->spec_alpha = sa;
    double *sb = (double *)realloc(L->spec_beta, (size_t)nc * sizeof(double));
    if (sb) L->spec_beta = sb;
    double *ai = (double *)realloc(L->arg_i, (size_t)nc * sizeof(double));
    if (ai) L->arg_i = ai;
    double *aa = (double *)realloc(L->arg_a, (size_t)nc * sizeof(double));
    if (aa) L->arg_a = aa;
    geodesic *g = (geodesic *)realloc(L->g, (size_t)nc * sizeof(geodesic));
    if (g) L->g = g;
    int *e = (int *)realloc(L->err, (size_t)nc * sizeof(int));
    if (e) L->err = e;
    int *o = (int *)realloc(L->ok, (size_t)nc * sizeof(int));
    if (o) L->ok = o;
    s5_chain *c = (s5_chain *)realloc(L->c, (size_t)nc * sizeof(s5_chain));
    if (c) L->c = c;
    if (!sa || !sb || !ai || !aa || !g || !e || !o || !c) { L->spec_cap = 0; return 0; }
    memset(L->g, 0, (size_t)nc * sizeof(geodesic));
    L->spec_cap = nc;
    return 1;
}

/* Records made ahead in ONE batch call; 0 if that was not possible (the caller goes on alone).
 *   what = 1: the template row with this call's beta -- and, where the vertical axis of the formula is known and reproduces
 *             this beta, the rows after it too, up to S5_AHEAD_MAX rays
 *   what = 2: pixels 2 .. nx-1 of the first row by the formula (this call is pixel 1: it is answered on its own) */
static int s5_make_ahead(fn_geod_chain fc, int what, double i, double a, double beta)
{
    s5_ahead *L = &s5_la;
    int n = 0;
    L->spec_n = 0;
    if (what == 2) {
        n = L->fit_nx - 2;
        if (n < 1 || !s5_spec_room(n)) return 0;
        for (int k = 0; k < n; k++) { L->spec_alpha[k] = s5_pixel(k + 2, L->fit_nx, L->fit_R, 1.0); L->spec_beta[k] = beta; }
    } else {
        const int nx = L->tmpl_n;
        int rows = 1;
        int y = L->rows_seen;                                  /* this row's index if the image started with the first template */
        const double sc = L->fit_scaled ? (double)L->fit_ny / (double)L->fit_nx : 1.0;
        /* (the next image of the same size and view starts over at row 0) */
        if (L->fit_ny > 0 && L->fit_nx == nx && s5_same_bits(s5_pixel(0, L->fit_ny, L->fit_R, sc), beta)) { y = 0; L->rows_seen = 0; }
        if (L->fit_ny > 0 && L->fit_nx == nx && y < L->fit_ny && s5_same_bits(s5_pixel(y, L->fit_ny, L->fit_R, sc), beta)) {
            rows = L->fit_ny - y;
            if (rows > S5_AHEAD_MAX / nx) rows = S5_AHEAD_MAX / nx;
            if (rows < 1) rows = 1;
        }
        n = rows * nx;
        if (!s5_spec_room(n)) return 0;
        for (int r = 0; r < rows; r++) {
            const double b = (r == 0) ? beta : s5_pixel(y + r, L->fit_ny, L->fit_R, sc);
            for (int k = 0; k < nx; k++) { L->spec_alpha[r * nx + k] = L->tmpl[k]; L->spec_beta[r * nx + k] = b; }
        }
    }
    for (int k = 0; k < n; k++) { L->arg_i[k] = i; L->arg_a[k] = a; }
    const unsigned long gen = s5_disk_generation();           /* before the call: a set-up during it leaves the records stale, not wrong */
    const int rc = fc((size_t)n, L->arg_i, L->arg_a, L->spec_alpha, L->spec_beta, L->g, L->err, L->ok, L->c);
    if (rc != 0) { s5_check(rc, "geodesic_init_inf (look-ahead)"); return 0; }
    L->spec_n = n; L->cursor = 0;
    L->spec_i = i; L->spec_a = a;
    L->disk_gen = gen;
    return 1;
}

/* the record made ahead for exactly these arguments, if it is the next one; -1 otherwise (what was made ahead is dropped) */
static int s5_row_take(double i, double a, double alpha, double beta)
{
    s5_ahead *L = &s5_la;
    if (!L->spec_n) return -1;
    if (L->cursor < L->spec_n && s5_same_bits(alpha, L->spec_alpha[L->cursor]) && s5_same_bits(beta, L->spec_beta[L->cursor]) &&
        s5_same_bits(i, L->spec_i) && s5_same_bits(a, L->spec_a)) return L->cursor++;
    if (L->cursor < L->spec_n) s5_forget_fit();               /* the caller left the predicted order: no formula any more */
    L->spec_n = 0;
    return -1;
}

int geodesic_init_inf(double i, double a, double alpha, double beta, geodesic *g, int *error)
{
    int err = 0, ok = 0, rc = 0;
    pthread_once(&s5_modes_once, s5_modes_init);
    if (s5_chain_mode) {
        S5_FN(fn_geod_chain, fc, "sim5gpu_geodesic_init_inf_chain");
        s5_last.live = 0;
        int k = -1;
        if (s5_lookahead) {
            s5_la_register();
            k = s5_row_take(i, a, alpha, beta);
            const int what = s5_row_note(i, a, alpha, beta);
            if (k < 0 && what == 1 && s5_make_ahead(fc, 1, i, a, beta)) k = s5_row_take(i, a, alpha, beta);
            else if (k < 0 && what == 2) (void)s5_make_ahead(fc, 2, i, a, beta);         /* (this call itself goes alone) */
        }
        if (k >= 0) {
            const s5_ahead *L = &s5_la;
            /* what geodesic_init_inf writes, and nothing else: dmdp_inf, k[4] and p (ref src/sim5kerr-geod.h:59,66-67) are never
             * set by it -- the caller's bytes stay there, as they do when the call goes to the GPU alone */
            memcpy(g, &L->g[k], offsetof(geodesic, dmdp_inf));
            memcpy(&g->Rpc, &L->g[k].Rpc, offsetof(geodesic, k) - offsetof(geodesic, Rpc));
            err = L->err[k]; ok = L->ok[k];
            if (ok) { memcpy(&s5_last.g, g, sizeof *g); s5_last.c = L->c[k]; s5_last.disk_gen = L->disk_gen; s5_last.live = 1; }
        } else {
            const unsigned long gen = s5_disk_generation();
            rc = s5_check(fc(1, &i, &a, &alpha, &beta, g, &err, &ok, &s5_last.c), "geodesic_init_inf");
            if (ok && !rc) { memcpy(&s5_last.g, g, sizeof *g); s5_last.disk_gen = gen; s5_last.live = 1; }
        }
    } else {
        S5_FN(fn_geod_init_inf, f, "sim5gpu_geodesic_init_inf");
        rc = s5_check(f(1, &i, &a, &alpha, &beta, g, &err, &ok), "geodesic_init_inf");
    }
    /* a failed batch call: FALSE with a non-zero status, like every failure of the reference (ref src/sim5kerr-geod.c:59-98) */
    if (rc) { ok = 0; if (!err) err = GD_ERROR_UNKNOWN_SOLUTION; }
    if (error) *error = err;
    return ok ? TRUE : FALSE;
}

int geodesic_init_src(double a, double r, double m, double k[4], int ppc, geodesic *g, int *error)
{
    S5_FN(fn_geod_init_src, f, "sim5gpu_geodesic_init_src");
    int err = 0, ok = 0;
    s5_check(f(1, &a, &r, &m, k, &ppc, g, &err, &ok), "geodesic_init_src");
    if (error) *error = err;
    return ok ? TRUE : FALSE;
}

double geodesic_P_int(geodesic *g, double r, int ppc)
{
    S5_FN(fn_geod_Pint, f, "sim5gpu_geodesic_P_int");
    double P = NAN;
    s5_check(f(1, g, &r, &ppc, &P), "geodesic_P_int");
    return P;
}

double geodesic_position_rad(geodesic *g, double P)
{
    if (s5_record_for(g)) {
        for (int k = 0; k < 2; k++) if (s5_last.c.have_r[k] && s5_same_bits(P, s5_last.c.P[k])) return s5_last.c.r[k];
    }
    S5_FN(fn_geod_P, f, "sim5gpu_geodesic_position_rad");
    double r = NAN;
    s5_check(f(1, g, &P, &r), "geodesic_position_rad");
    return r;
}

double geodesic_position_pol(geodesic *g, double P)
{
    S5_FN(fn_geod_P, f, "sim5gpu_geodesic_position_pol");
    double m = NAN;
    s5_check(f(1, g, &P, &m), "geodesic_position_pol");
    return m;
}

double geodesic_dm_sign(geodesic *g, double P)
{
    S5_FN(fn_geod_P, f, "sim5gpu_geodesic_dm_sign");
    double s = NAN;
    s5_check(f(1, g, &P, &s), "geodesic_dm_sign");
    return s;
}

void geodesic_momentum(geodesic *g, double P, double r, double m, double k[])
{
    S5_FN(fn_geod_mom, f, "sim5gpu_geodesic_momentum");
    s5_check(f(1, g, &P, &r, &m, k), "geodesic_momentum");
}

/* ref: src/sim5kerr-geod.c:463-556 */
double geodesic_position_azm(geodesic *g, double r, double m, double P)
{
    S5_FN(fn_geod_azm, f, "sim5gpu_geodesic_position_azm");
    double phi = NAN;
    s5_check(f(1, g, &r, &m, &P, &phi), "geodesic_position_azm");
    return phi;
}

/* ref: src/sim5kerr-geod.c:560-664 */
double geodesic_timedelay(geodesic *g, double P1, double r1, double m1, double P2, double r2, double m2)
{
    S5_FN(fn_geod_delay, f, "sim5gpu_geodesic_timedelay");
    double dt = NAN;
    s5_check(f(1, g, &P1, &r1, &m1, &P2, &r2, &m2, &dt), "geodesic_timedelay");
    return dt;
}

double geodesic_find_midplane_crossing(geodesic *g, int order)
{
    if ((order == 0 || order == 1) && s5_record_for(g)) return s5_last.c.P[order];
    S5_FN(fn_geod_order, f, "sim5gpu_geodesic_find_midplane_crossing");
    double P = NAN;
    s5_check(f(1, g, &order, &P), "geodesic_find_midplane_crossing");
    return P;
}

void geodesic_follow(geodesic *g, double step, double *P, double *r, double *m, int *status)
{
    S5_FN(fn_geod_follow, f, "sim5gpu_geodesic_follow");
    int st = 0;
    s5_check(f(1, g, &step, P, r, m, &st), "geodesic_follow");
    if (status) *status = st;
}

void kerr_metric(double a, double r, double m, sim5metric *metric)
{
    typedef int (*fn)(size_t, const double *, const double *, const double *, sim5metric *);
    S5_FN(fn, f, "sim5gpu_kerr_metric");
    s5_check(f(1, &a, &r, &m, metric), "kerr_metric");
}

void kerr_connection(double a, double r, double m, double G[4][4][4])
{
    typedef int (*fn)(size_t, const double *, const double *, const double *, double *);
    S5_FN(fn, f, "sim5gpu_kerr_connection");
    s5_check(f(1, &a, &r, &m, &G[0][0][0]), "kerr_connection");
}

double dotprod(double V1[4], double V2[4], sim5metric *m)
{
    typedef int (*fn)(size_t, const double *, const double *, const sim5metric *, double *);
    S5_FN(fn, f, "sim5gpu_dotprod");
    double out = NAN;
    s5_check(f(1, V1, V2, m, &out), "dotprod");
    return out;
}

void tetrad_zamo(sim5metric *m, sim5tetrad *t)
{
    typedef int (*fn)(size_t, const sim5metric *, sim5tetrad *);
    S5_FN(fn, f, "sim5gpu_tetrad_zamo");
    s5_check(f(1, m, t), "tetrad_zamo");
}

void tetrad_azimuthal(sim5metric *m, double Omega, sim5tetrad *t)
{
    typedef int (*fn)(size_t, const sim5metric *, const double *, sim5tetrad *);
    S5_FN(fn, f, "sim5gpu_tetrad_azimuthal");
    s5_check(f(1, m, &Omega, t), "tetrad_azimuthal");
}

void tetrad_surface(sim5metric *m, double Omega, double V, double dhdr, sim5tetrad *t)
{
    typedef int (*fn)(size_t, const sim5metric *, const double *, const double *, const double *, sim5tetrad *);
    S5_FN(fn, f, "sim5gpu_tetrad_surface");
    s5_check(f(1, m, &Omega, &V, &dhdr, t), "tetrad_surface");
}

void bl2on(double Vin[4], double Vout[4], sim5tetrad *t)
{
    typedef int (*fn)(size_t, const double *, double *, const sim5tetrad *);
    S5_FN(fn, f, "sim5gpu_bl2on");
    s5_check(f(1, Vin, Vout, t), "bl2on");
}

void on2bl(double Vin[4], double Vout[4], sim5tetrad *t)
{
    typedef int (*fn)(size_t, const double *, double *, const sim5tetrad *);
    S5_FN(fn, f, "sim5gpu_on2bl");
    s5_check(f(1, Vin, Vout, t), "on2bl");
}

double r_bh(double a) { S5_FN(fn_d1, f, "sim5gpu_r_bh"); double r = NAN; s5_check(f(1, &a, &r), "r_bh"); return r; }
double r_ms(double a) { S5_FN(fn_d1, f, "sim5gpu_r_ms"); double r = NAN; s5_check(f(1, &a, &r), "r_ms"); return r; }
double r_mb(double a) { S5_FN(fn_d1, f, "sim5gpu_r_mb"); double r = NAN; s5_check(f(1, &a, &r), "r_mb"); return r; }
double r_ph(double a) { S5_FN(fn_d1, f, "sim5gpu_r_ph"); double r = NAN; s5_check(f(1, &a, &r), "r_ph"); return r; }
double OmegaK(double r, double a) { S5_FN(fn_d2, f, "sim5gpu_OmegaK"); double o = NAN; s5_check(f(1, &r, &a, &o), "OmegaK"); return o; }
double ellK(double r, double a) { S5_FN(fn_d2, f, "sim5gpu_ellK"); double o = NAN; s5_check(f(1, &r, &a, &o), "ellK"); return o; }

double Omega_from_ell(double ell, sim5metric *m)
{
    typedef int (*fn)(size_t, const double *, const sim5metric *, double *);
    S5_FN(fn, f, "sim5gpu_Omega_from_ell");
    double o = NAN;
    s5_check(f(1, &ell, m, &o), "Omega_from_ell");
    return o;
}

double gfactorK(double r, double a, double l)
{
    if (s5_last.live && s5_same_bits(a, s5_last.c.a) && s5_same_bits(l, s5_last.c.l)) {
        for (int k = 0; k < 2; k++) if (s5_last.c.have_r[k] && !isnan(r) && s5_same_bits(r, s5_last.c.r[k])) return s5_last.c.g[k];
    }
    S5_FN(fn_d3, f, "sim5gpu_gfactorK");
    double g = NAN;
    s5_check(f(1, &r, &a, &l, &g), "gfactorK");
    return g;
}

void photon_momentum(double a, double r, double m, double l, double q, double r_sign, double m_sign, double k[4])
{
    typedef int (*fn)(size_t, const double *, const double *, const double *, const double *, const double *,
                      const double *, const double *, double *);
    S5_FN(fn, f, "sim5gpu_photon_momentum");
    s5_check(f(1, &a, &r, &m, &l, &q, &r_sign, &m_sign, k), "photon_momentum");
}

void photon_motion_constants(double a, double r, double m, double k[4], double *L, double *Q)
{
    typedef int (*fn)(size_t, const double *, const double *, const double *, const double *, double *, double *);
    S5_FN(fn, f, "sim5gpu_photon_motion_constants");
    s5_check(f(1, &a, &r, &m, k, L, Q), "photon_motion_constants");
}

double photon_carter_const(double k[4], sim5metric *metric)
{
    typedef int (*fn)(size_t, const double *, const sim5metric *, double *);
    S5_FN(fn, f, "sim5gpu_photon_carter_const");
    double Q = NAN;
    s5_check(f(1, k, metric, &Q), "photon_carter_const");
    return Q;
}

void raytrace_prepare(double bh_spin, double x[4], double k[4], double presision_factor, int options, raytrace_data *rtd)
{
    typedef int (*fn)(size_t, const double *, const double *, const double *, const double *, const int *, raytrace_data *);
    S5_FN(fn, f, "sim5gpu_raytrace_prepare");
    s5_check(f(1, &bh_spin, x, k, &presision_factor, &options, rtd), "raytrace_prepare");
}

/* raytrace() one call at a time (ref README.md:184-193, src/sim5unittests.c:116-127: while (...) { dl = cap; raytrace(x, k, &dl,
 * &rtd); }).  A single call is a kernel launch of one lane (~20 us); the loop above feeds every call the previous call's results
 * and the same cap, so the shim asks the GPU for the NEXT calls as well -- sim5gpu_raytrace_record: K consecutive calls of the
 * same ray in one launch, every intermediate state kept -- and answers the following calls from those records: each ONLY after
 * its x, k, *step and the 144 bytes of *rtd have been compared, bit for bit, with what its record was made from (the previous
 * record's output, the same cap).  Any other call (a caller that changes the cap, the state, the ray) drops the records and goes
 * to the GPU; two misses in a row switch the look-ahead off for the next 256 calls.  K doubles from 8 to 64 while the records are
 * being used up.  SIM5_SHIM_NO_LOOKAHEAD=1: one launch per call.  (A ray alone advances one call per ~6 us on the GPU -- the
 * dependent chain of one lane -- so this loop stays an order of magnitude behind one CPU core whatever the shim does:
 * INTEGRATION.md 1; whole images of rays belong to sim5gpu_torus_image.) */
typedef struct { double x[4], k[4], step; raytrace_data rtd; } s5_rt_step;          /* = sim5gpu_raytrace_step */
#define S5_RT_MAX 64
static __thread struct {
    int n, cursor, K, misses, off_for;
    double cap;
    double x0[4], k0[4]; raytrace_data rtd0;                 /* what record 0 was made from */
    s5_rt_step rec[S5_RT_MAX];
} s5_rt;

static int s5_rt_matches(const double x[4], const double k[4], double cap, const raytrace_data *rtd)
{
    if (s5_rt.cursor >= s5_rt.n || !s5_same_bits(cap, s5_rt.cap)) return 0;
    const double *px = s5_rt.cursor ? s5_rt.rec[s5_rt.cursor - 1].x : s5_rt.x0;
    const double *pk = s5_rt.cursor ? s5_rt.rec[s5_rt.cursor - 1].k : s5_rt.k0;
    const raytrace_data *pr = s5_rt.cursor ? &s5_rt.rec[s5_rt.cursor - 1].rtd : &s5_rt.rtd0;
    return memcmp(x, px, 4 * sizeof(double)) == 0 && memcmp(k, pk, 4 * sizeof(double)) == 0 && memcmp(rtd, pr, sizeof *rtd) == 0;
}

void raytrace(double x[4], double k[4], double *step, raytrace_data *rtd)
{
    pthread_once(&s5_modes_once, s5_modes_init);
    if (s5_lookahead) {
        if (s5_rt.n && s5_rt_matches(x, k, *step, rtd)) {
            const s5_rt_step *o = &s5_rt.rec[s5_rt.cursor++];
            memcpy(x, o->x, sizeof o->x); memcpy(k, o->k, sizeof o->k); *step = o->step; memcpy(rtd, &o->rtd, sizeof *rtd);
            s5_rt.misses = 0;
            return;
        }
        /* not the call the records were made for (or none left) */
        const int used_up = s5_rt.n > 0 && s5_rt.cursor == s5_rt.n;
        if (s5_rt.n > 0 && s5_rt.cursor <= 1 && !used_up && ++s5_rt.misses >= 2) { s5_rt.off_for = 256; s5_rt.misses = 0; }
        s5_rt.K = used_up ? (s5_rt.K * 2 > S5_RT_MAX ? S5_RT_MAX : s5_rt.K * 2) : 8;
        s5_rt.n = 0;
        if (s5_rt.off_for > 0) s5_rt.off_for--;
        else {
            typedef int (*fnr)(const double *, const double *, double, const raytrace_data *, int, s5_rt_step *);
            S5_FN(fnr, fr, "sim5gpu_raytrace_record");
            memcpy(s5_rt.x0, x, sizeof s5_rt.x0); memcpy(s5_rt.k0, k, sizeof s5_rt.k0); memcpy(&s5_rt.rtd0, rtd, sizeof *rtd);
            s5_rt.cap = *step;
            if (s5_check(fr(x, k, *step, rtd, s5_rt.K, s5_rt.rec), "raytrace (look-ahead)") == 0) {
                s5_rt.n = s5_rt.K; s5_rt.cursor = 1;
                const s5_rt_step *o = &s5_rt.rec[0];
                memcpy(x, o->x, sizeof o->x); memcpy(k, o->k, sizeof o->k); *step = o->step; memcpy(rtd, &o->rtd, sizeof *rtd);
                return;
            }
        }
    }
    typedef int (*fn)(size_t, double *, double *, double *, raytrace_data *, int);
    S5_FN(fn, f, "sim5gpu_raytrace");
    s5_check(f(1, x, k, step, rtd, 1), "raytrace");
}

double raytrace_error(double x[4], double k[4], raytrace_data *rtd)
{
    typedef int (*fn)(size_t, const double *, const double *, const raytrace_data *, double *);
    S5_FN(fn, f, "sim5gpu_raytrace_error");
    double e = NAN;
    s5_check(f(1, x, k, rtd, &e), "raytrace_error");
    return e;
}

/* the arguments of the last set-up, as the reference keeps them (floats), for the header of disk_nt_dump only */
static float s5_disk_M = 10.0, s5_disk_a = 0.0, s5_disk_alpha = 0.1;
static int s5_disk_options = 0;

int disk_nt_setup(double M, double a, double mdot_or_L, double alpha, int options)
{
    typedef int (*fn)(double, double, double, double, int);
    S5_FN(fn, f, "sim5gpu_disk_nt_setup");
    s5_check(f(M, a, mdot_or_L, alpha, options), "disk_nt_setup");
    s5_disk_M = M; s5_disk_a = a; s5_disk_alpha = alpha; s5_disk_options = options;
    /* (records made for the previous model no longer answer disk_nt_flux: the library bumped its generation counter) */
    return 0;
}

void disk_nt_done(void) {}

double disk_nt_r_min(void)
{
    typedef int (*fn)(double *);
    S5_FN(fn, f, "sim5gpu_disk_nt_r_min");
    double r = NAN;
    s5_check(f(&r), "disk_nt_r_min");
    return r;
}

static double disk_nt_r_min_f(void) { return (float)disk_nt_r_min(); }     /* the float static disk_nt_disk_rms */
double disk_nt_vr(double r) { (void)r; return 0.0; }                          /* ref src/sim5disk-nt.c:253-303 */
double disk_nt_h(double r) { (void)r; return 0.0; }
double disk_nt_dhdr(double r) { (void)r; return 0.0; }

double disk_nt_flux(double r)
{
    if (s5_last.live && s5_last.c.flux_valid && s5_last.disk_gen == s5_disk_generation()) {
        for (int k = 0; k < 2; k++) if (s5_last.c.have_r[k] && !isnan(r) && s5_same_bits(r, s5_last.c.r[k])) return s5_last.c.flux[k];
    }
    S5_FN(fn_d1, f, "sim5gpu_disk_nt_flux"); double o = NAN; s5_check(f(1, &r, &o), "disk_nt_flux"); return o;
}
double disk_nt_ell(double r) { S5_FN(fn_d1, f, "sim5gpu_disk_nt_ell"); double o = NAN; s5_check(f(1, &r, &o), "disk_nt_ell"); return o; }
double disk_nt_sigma(double r) { S5_FN(fn_d1, f, "sim5gpu_disk_nt_sigma"); double o = NAN; s5_check(f(1, &r, &o), "disk_nt_sigma"); return o; }

double disk_nt_mdot(void)
{
    typedef int (*fn)(double *);
    S5_FN(fn, f, "sim5gpu_disk_nt_mdot");
    double v = NAN;
    s5_check(f(&v), "disk_nt_mdot");
    return v;
}

double disk_nt_lumi(void)
{
    typedef int (*fn)(double *);
    S5_FN(fn, f, "sim5gpu_disk_nt_lumi");
    double v = NAN;
    s5_check(f(&v), "disk_nt_lumi");
    return v;
}

/* radial profile dump in the reference's layout (ref src/sim5disk-nt.c:310-360): header with the model parameters,
 * then one line per radius from the inner edge to 2000 r_g in steps of 5 %; the whole profile is two batch calls */
void disk_nt_dump(char *filename)
{
    FILE *stream = stdout;
    if (filename) stream = fopen(filename, "w");
    if (!stream) { fprintf(stderr, "disk_nt_dump: cannot open output (%s)\n", filename); return; }
    const float disk_rmax = 2000.;
    double rr[512], fl[512], sg[512], el[512];
    size_t n = 0;
    double r;
    for (r = (float)disk_nt_r_min_f(); r < disk_rmax && n < 512; r *= 1.05) rr[n++] = r;
    { S5_FN(fn_d1, f, "sim5gpu_disk_nt_flux"); s5_check(f(n, rr, fl), "disk_nt_flux"); }
    { S5_FN(fn_d1, f, "sim5gpu_disk_nt_sigma"); s5_check(f(n, rr, sg), "disk_nt_sigma"); }
    { S5_FN(fn_d1, f, "sim5gpu_disk_nt_ell"); s5_check(f(n, rr, el), "disk_nt_ell"); }
    fprintf(stream, "# (sim5disk-nt) dump\n");
    fprintf(stream, "#-------------------------------------------\n");
    fprintf(stream, "# M        = %.4f\n", s5_disk_M);
    fprintf(stream, "# a        = %.4f\n", s5_disk_a);
    fprintf(stream, "# rmin     = %.4f\n", (float)disk_nt_r_min_f());
    fprintf(stream, "# rmax     = %.4f\n", disk_rmax);
    fprintf(stream, "# alpha    = %.4f\n", s5_disk_alpha);
    fprintf(stream, "# options  = %d\n", s5_disk_options);
    fprintf(stream, "# L        = %e\n", disk_nt_lumi());
    fprintf(stream, "# mdot     = %e\n", disk_nt_mdot());
    fprintf(stream, "#-------------------------------------------\n");
    fprintf(stream, "# r   flux   sigma   ell   vr   H   dH/dr\n");
    fprintf(stream, "#-------------------------------------------\n");
    for (size_t i = 0; i < n; i++)
        fprintf(stream, "%e  %e  %e  %e  %e  %e  %e\n", rr[i], fl[i], sg[i], el[i], 0.0, 0.0, 0.0);
    fflush(stream);
    if (filename) fclose(stream);
}

sim5complex polarization_constant(double k[4], double f[4], sim5metric *metric)
{
    typedef int (*fn)(size_t, const double *, const double *, const sim5metric *, double *);
    S5_FN(fn, fp, "sim5gpu_polarization_constant");
    double wp[2] = { NAN, NAN };
    s5_check(fp(1, k, f, metric, wp), "polarization_constant");
    return wp[0] + _Complex_I * wp[1];
}

void polarization_vector(double k[4], sim5complex wp, sim5metric *metric, double f[4])
{
    typedef int (*fn)(size_t, const double *, const double *, const sim5metric *, double *);
    S5_FN(fn, fp, "sim5gpu_polarization_vector");
    double w[2] = { creal(wp), cimag(wp) };
    s5_check(fp(1, k, w, metric, f), "polarization_vector");
}

sim5complex polarization_constant_infinity(double a, double alpha, double beta, double incl)
{
    typedef int (*fn)(size_t, const double *, const double *, const double *, const double *, double *);
    S5_FN(fn, fp, "sim5gpu_polarization_constant_infinity");
    double wp[2] = { NAN, NAN };
    s5_check(fp(1, &a, &alpha, &beta, &incl, wp), "polarization_constant_infinity");
    return wp[0] + _Complex_I * wp[1];
}

double polarization_angle_rotation(double a, double inc, double alpha, double beta, sim5complex kappa)
{
    typedef int (*fn)(size_t, const double *, const double *, const double *, const double *, const double *, double *);
    S5_FN(fn, fp, "sim5gpu_polarization_angle_rotation");
    double w[2] = { creal(kappa), cimag(kappa) }, out = NAN;
    s5_check(fp(1, &a, &inc, &alpha, &beta, w, &out), "polarization_angle_rotation");
    return out;
}

double blackbody_Iv(double T, double hardf, double cos_mu, double E)
{
    typedef int (*fn)(size_t, const double *, const double *, const double *, const double *, double *);
    S5_FN(fn, fp, "sim5gpu_blackbody_Iv");
    double out = NAN;
    s5_check(fp(1, &T, &hardf, &cos_mu, &E, &out), "blackbody_Iv");
    return out;
}

/* elliptic functions: selector numbering of sim5gpu_elliptic (include/sim5gpu.h) */
static double s5_ell(int which, const double *x, const double *y, const double *z, const double *w)
{
    typedef int (*fn)(int, size_t, const double *, const double *, const double *, const double *, double *);
    S5_FN(fn, fp, "sim5gpu_elliptic");
    double out = NAN;
    s5_check(fp(which, 1, x, y, z, w, &out), "elliptic");
    return out;
}
double rf(double x, double y, double z) { return s5_ell(0, &x, &y, &z, 0); }
double elliptic_k(double m) { return s5_ell(1, &m, 0, 0, 0); }
double jacobi_isn(double z, double m) { return s5_ell(2, &z, &m, 0, 0); }
double jacobi_icn(double z, double m) { return s5_ell(3, &z, &m, 0, 0); }
double jacobi_itn(double z, double m) { return s5_ell(4, &z, &m, 0, 0); }
double jacobi_sn(double u, double m) { return s5_ell(5, &u, &m, 0, 0); }
double jacobi_cn(double u, double m) { return s5_ell(6, &u, &m, 0, 0); }
double jacobi_dn(double u, double m) { return s5_ell(7, &u, &m, 0, 0); }
double rd(double x, double y, double z) { return s5_ell(8, &x, &y, &z, 0); }
double rc(double x, double y) { return s5_ell(9, &x, &y, 0, 0); }
double rj(double x, double y, double z, double p) { return s5_ell(10, &x, &y, &z, &p); }
double elliptic_f_sin(double sin_phi, double m) { return s5_ell(11, &sin_phi, &m, 0, 0); }

/* the integrals under geodesic_position_azm / geodesic_timedelay: sim5gpu_integral(which, 1, nargs, args, out) */
typedef int (*fn_integral)(int, size_t, int, const double *, double *);
static double s5_int(int which, int nargs, const double *args)
{
    S5_FN(fn_integral, f, "sim5gpu_integral");
    double out = NAN;
    s5_check(f(which, 1, nargs, args, &out), "integral");
    return out;
}
#define S5_ARGS(...) (const double[]){ __VA_ARGS__ }
double elliptic_f_cos(double c, double m) { return s5_int(0, 2, S5_ARGS(c, m)); }
double elliptic_e_cos(double c, double m) { return s5_int(1, 2, S5_ARGS(c, m)); }
double elliptic_pi_complete(double n, double m) { return s5_int(2, 2, S5_ARGS(n, m)); }
double elliptic_pi_cos(double c, double n, double m) { return s5_int(3, 3, S5_ARGS(c, n, m)); }
double integral_R_r0_re(double a, double b, double c, double d, double X) { return s5_int(12, 5, S5_ARGS(a, b, c, d, X)); }
double integral_R_r0_re_inf(double a, double b, double c, double d) { return s5_int(13, 4, S5_ARGS(a, b, c, d)); }
double integral_R_r1_re(double a, double b, double c, double d, double X) { return s5_int(14, 5, S5_ARGS(a, b, c, d, X)); }
double integral_R_r2_re(double a, double b, double c, double d, double X) { return s5_int(15, 5, S5_ARGS(a, b, c, d, X)); }
double integral_R_rp_re(double a, double b, double c, double d, double p, double X) { return s5_int(16, 6, S5_ARGS(a, b, c, d, p, X)); }
double integral_R_rp_re_inf(double a, double b, double c, double d, double p) { return s5_int(17, 5, S5_ARGS(a, b, c, d, p)); }
double integral_R_r0_cc(double a, double b, sim5complex c, double X) { return s5_int(18, 5, S5_ARGS(a, b, creal(c), cimag(c), X)); }
double integral_R_r0_cc_inf(double a, double b, sim5complex c) { return s5_int(19, 4, S5_ARGS(a, b, creal(c), cimag(c))); }
double integral_R_r1_cc(double a, double b, sim5complex c, double X1, double X2) { return s5_int(20, 6, S5_ARGS(a, b, creal(c), cimag(c), X1, X2)); }
double integral_R_r2_cc(double a, double b, sim5complex c, double X1, double X2) { return s5_int(21, 6, S5_ARGS(a, b, creal(c), cimag(c), X1, X2)); }
double integral_R_rp_cc2(double a, double b, sim5complex c, double p, double X1, double X2) { return s5_int(22, 7, S5_ARGS(a, b, creal(c), cimag(c), p, X1, X2)); }
double integral_R_rp_cc2_inf(double a, double b, sim5complex c, double p, double X1) { return s5_int(23, 6, S5_ARGS(a, b, creal(c), cimag(c), p, X1)); }
double integral_T_m0(double a2, double b2, double X) { return s5_int(24, 3, S5_ARGS(a2, b2, X)); }
double integral_T_m2(double a2, double b2, double X) { return s5_int(25, 3, S5_ARGS(a2, b2, X)); }
double integral_T_mp(double a2, double b2, double p, double X) { return s5_int(26, 4, S5_ARGS(a2, b2, p, X)); }
#undef S5_ARGS

/* ref: src/sim5kerr.c:553-573 */
typedef int (*fn_vnorm)(size_t, double *, const double *, const sim5metric *);
void vector_norm_to(double V[4], double norm, sim5metric *m)
{
    S5_FN(fn_vnorm, f, "sim5gpu_vector_norm_to");
    s5_check(f(1, V, &norm, m), "vector_norm_to");
}

void jacobi_sncndn(double u, double m, double *sn, double *cn, double *dn)
{
    *sn = jacobi_sn(u, m); *cn = jacobi_cn(u, m); *dn = jacobi_dn(u, m);
}

/* ---------------------------------------------------------------------------------------------------------
 * The rest of the public prototypes of the reference headers this path cites (ref src/sim5kerr.h:36-175,
 * src/sim5kerr-geod.h:74,77, src/sim5elliptic.h:25-33, src/sim5radiation.h:33-35, src/sim5math.h:69-91,
 * src/sim5polyroots.h:26).  Everything that computes goes to the library (group (1b) of include/sim5gpu.h, n = 1);
 * what only moves, compares or reorders values (vector_set/copy/multiply, the complex accessors, ensure_range,
 * sort_roots, the angle reductions) is done here, as it involves no ray arithmetic.
 * --------------------------------------------------------------------------------------------------------- */
typedef int (*fn_rm_metric)(size_t, const double *, const double *, sim5metric *);
typedef int (*fn_arm_metric)(size_t, const double *, const double *, const double *, sim5metric *);

void flat_metric(double r, double m, sim5metric *metric)
{
    S5_FN(fn_rm_metric, f, "sim5gpu_flat_metric");
    s5_check(f(1, &r, &m, metric), "flat_metric");
}

void flat_metric_contravariant(double r, double m, sim5metric *metric)
{
    S5_FN(fn_rm_metric, f, "sim5gpu_flat_metric_contravariant");
    s5_check(f(1, &r, &m, metric), "flat_metric_contravariant");
}

void kerr_metric_contravariant(double a, double r, double m, sim5metric *metric)
{
    S5_FN(fn_arm_metric, f, "sim5gpu_kerr_metric_contravariant");
    s5_check(f(1, &a, &r, &m, metric), "kerr_metric_contravariant");
}

void kerr_newman_metric(double a, double Q, double r, double m, sim5metric *metric)
{
    typedef int (*fn)(size_t, const double *, const double *, const double *, const double *, sim5metric *);
    S5_FN(fn, f, "sim5gpu_kerr_newman_metric");
    s5_check(f(1, &a, &Q, &r, &m, metric), "kerr_newman_metric");
}

void kerr_newman_metric_contravariant(double a, double Q, double r, double m, sim5metric *metric)
{
    typedef int (*fn)(size_t, const double *, const double *, const double *, const double *, sim5metric *);
    S5_FN(fn, f, "sim5gpu_kerr_newman_metric_contravariant");
    s5_check(f(1, &a, &Q, &r, &m, metric), "kerr_newman_metric_contravariant");
}

void kerr_newman_connection(double a, double Q, double r, double m, double G[4][4][4])
{
    typedef int (*fn)(size_t, const double *, const double *, const double *, const double *, double *);
    S5_FN(fn, f, "sim5gpu_kerr_newman_connection");
    s5_check(f(1, &a, &Q, &r, &m, &G[0][0][0]), "kerr_newman_connection");
}

void flat_connection(double r, double m, double G[4][4][4])
{
    typedef int (*fn)(size_t, const double *, const double *, double *);
    S5_FN(fn, f, "sim5gpu_flat_connection");
    s5_check(f(1, &r, &m, &G[0][0][0]), "flat_connection");
}

void Gamma(double G[4][4][4], double U[4], double V[4], double result[4])
{
    typedef int (*fn)(size_t, const double *, const double *, const double *, double *);
    S5_FN(fn, f, "sim5gpu_Gamma");
    s5_check(f(1, &G[0][0][0], U, V, result), "Gamma");
}

void vector_set(double x[4], double x0, double x1, double x2, double x3) { x[0] = x0; x[1] = x1; x[2] = x2; x[3] = x3; }
void vector_copy(double src[4], double dst[4]) { dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3]; }

void vector_multiply(double V[4], double factor)
{
    /* one rounding per component, the same on any IEEE machine (ref src/sim5kerr.c:536-549) */
    V[0] *= factor; V[1] *= factor; V[2] *= factor; V[3] *= factor;
}

void vector_covariant(double V1[4], double V2[4], sim5metric *m)
{
    typedef int (*fn)(size_t, const double *, double *, const sim5metric *);
    S5_FN(fn, f, "sim5gpu_vector_covariant");
    s5_check(f(1, V1, V2, m), "vector_covariant");
}

double vector_norm(double V[4], sim5metric *m)
{
    typedef int (*fn)(size_t, const double *, const sim5metric *, double *);
    S5_FN(fn, f, "sim5gpu_vector_norm");
    double out = NAN;
    s5_check(f(1, V, m, &out), "vector_norm");
    return out;
}

double vector_3norm(double V[4])
{
    typedef int (*fn)(size_t, const double *, double *);
    S5_FN(fn, f, "sim5gpu_vector_3norm");
    double out = NAN;
    s5_check(f(1, V, &out), "vector_3norm");
    return out;
}

void vector_norm_to_null(double V[4], double V0, sim5metric *m)
{
    typedef int (*fn)(size_t, double *, const double *, const sim5metric *);
    S5_FN(fn, f, "sim5gpu_vector_norm_to_null");
    s5_check(f(1, V, &V0, m), "vector_norm_to_null");
}

void tetrad_general(sim5metric *m, double U[], sim5tetrad *t)
{
    typedef int (*fn)(size_t, const sim5metric *, const double *, sim5tetrad *);
    S5_FN(fn, f, "sim5gpu_tetrad_general");
    s5_check(f(1, m, U, t), "tetrad_general");
}

void tetrad_radial(sim5metric *m, double v_r, sim5tetrad *t)
{
    typedef int (*fn)(size_t, const sim5metric *, const double *, sim5tetrad *);
    S5_FN(fn, f, "sim5gpu_tetrad_radial");
    s5_check(f(1, m, &v_r, t), "tetrad_radial");
}

double omega_r(double r, double a)
{
    S5_FN(fn_d2, f, "sim5gpu_omega_r");
    double out = NAN;
    s5_check(f(1, &r, &a, &out), "omega_r");
    return out;
}

double omega_z(double r, double a)
{
    S5_FN(fn_d2, f, "sim5gpu_omega_z");
    double out = NAN;
    s5_check(f(1, &r, &a, &out), "omega_z");
    return out;
}

double ell_from_Omega(double Omega, sim5metric *m)
{
    typedef int (*fn)(size_t, const double *, const sim5metric *, double *);
    S5_FN(fn, f, "sim5gpu_ell_from_Omega");
    double out = NAN;
    s5_check(f(1, &Omega, m, &out), "ell_from_Omega");
    return out;
}

void fourvelocity_zamo(sim5metric *m, double U[4])
{
    typedef int (*fn)(size_t, const sim5metric *, double *);
    S5_FN(fn, f, "sim5gpu_fourvelocity_zamo");
    s5_check(f(1, m, U), "fourvelocity_zamo");
}

typedef int (*fn_fourvel1)(size_t, const double *, const sim5metric *, double *);
void fourvelocity_azimuthal(double Omega, sim5metric *m, double U[4])
{
    S5_FN(fn_fourvel1, f, "sim5gpu_fourvelocity_azimuthal");
    s5_check(f(1, &Omega, m, U), "fourvelocity_azimuthal");
}

void fourvelocity_radial(double vr, sim5metric *m, double U[4])
{
    S5_FN(fn_fourvel1, f, "sim5gpu_fourvelocity_radial");
    s5_check(f(1, &vr, m, U), "fourvelocity_radial");
}

typedef int (*fn_fourvel3)(size_t, const double *, const double *, const double *, const sim5metric *, double *);
double fourvelocity_norm(double U1, double U2, double U3, sim5metric *m)
{
    S5_FN(fn_fourvel3, f, "sim5gpu_fourvelocity_norm");
    double out = NAN;
    s5_check(f(1, &U1, &U2, &U3, m, &out), "fourvelocity_norm");
    return out;
}

void fourvelocity(double U1, double U2, double U3, sim5metric *m, double U[])
{
    S5_FN(fn_fourvel3, f, "sim5gpu_fourvelocity");
    s5_check(f(1, &U1, &U2, &U3, m, U), "fourvelocity");
}

/* an empty stub in the reference (ref src/sim5kerr-geod.c:266-283): x is left untouched */
void geodesic_position(geodesic *g, double P, double x[4]) { (void)g; (void)P; (void)x; }

double geodesic_position_pol_sign_k_theta(geodesic *g, double P)
{
    S5_FN(fn_geod_P, f, "sim5gpu_geodesic_position_pol_sign_k_theta");
    double s = NAN;
    s5_check(f(1, g, &P, &s), "geodesic_position_pol_sign_k_theta");
    return s;
}

/* Legendre integrals by angle / by sine: selector numbering of sim5gpu_legendre (include/sim5gpu.h) */
typedef int (*fn_legendre)(int, size_t, const double *, const double *, const double *, double *);
double elliptic_f(double phi, double m)
{
    S5_FN(fn_legendre, f, "sim5gpu_legendre");
    double out = NAN;
    s5_check(f(0, 1, &phi, NULL, &m, &out), "elliptic_f");
    return out;
}

double elliptic_e_sin(double sin_phi, double m)
{
    S5_FN(fn_legendre, f, "sim5gpu_legendre");
    double out = NAN;
    s5_check(f(1, 1, &sin_phi, NULL, &m, &out), "elliptic_e_sin");
    return out;
}

double elliptic_pi_sin(double sin_phi, double n, double m)
{
    S5_FN(fn_legendre, f, "sim5gpu_legendre");
    double out = NAN;
    s5_check(f(2, 1, &sin_phi, &n, &m, &out), "elliptic_pi_sin");
    return out;
}

sim5complex elliptic_pi(double phi, double n, double m)
{
    S5_FN(fn_legendre, f, "sim5gpu_legendre");
    double out[2] = { NAN, NAN };
    s5_check(f(3, 1, &phi, &n, &m, out), "elliptic_pi");
    return out[0] + out[1] * _Complex_I;
}

void blackbody(double T, double hardf, double cos_mu, double E[], double Iv[], int en_bins)
{
    typedef int (*fn)(double, double, double, size_t, const double *, double *);
    S5_FN(fn, f, "sim5gpu_blackbody");
    if (en_bins <= 0) return;
    s5_check(f(T, hardf, cos_mu, (size_t)en_bins, E, Iv), "blackbody");
}

double blackbody_photons(double T, double hardf, double cos_mu, double E)
{
    typedef int (*fn)(size_t, const double *, const double *, const double *, const double *, double *);
    S5_FN(fn, f, "sim5gpu_blackbody_photons");
    double out = NAN;
    s5_check(f(1, &T, &hardf, &cos_mu, &E, &out), "blackbody_photons");
    return out;
}

double blackbody_photons_total(double T, double hardf)
{
    S5_FN(fn_d2, f, "sim5gpu_blackbody_photons_total");
    double out = NAN;
    s5_check(f(1, &T, &hardf, &out), "blackbody_photons_total");
    return out;
}

/* ---- values moved, compared or reordered (ref src/sim5math.c:16-58, 188-223, src/sim5polyroots.c:278-325) ---- */
long sim5round(double num) { return (long)(num + 0.5); }

long int factorial(long int n) { return (n <= 1) ? 1 : n * factorial(n - 1); }

double reduce_angle_pi(double phi)
{
    while (phi < 0.0) phi += 2. * PI;
    while (phi > PI) phi -= PI;
    return phi;
}

double reduce_angle_2pi(double phi)
{
    while (phi >= +PI2) phi -= PI2;
    while (phi < 0.0) phi += PI2;
    return phi;
}

int ensure_range(double *val, double min, double max, double acc)
{
    if (*val < min - acc) return 0;
    if (*val > max + acc) return 0;
    if (*val < min) *val = min;
    if (*val > max) *val = max;
    return 1;
}

sim5complex makeComplex(double r, double i) { return r + _Complex_I * i; }
sim5complex nullComplex(void) { return 0.0; }
double sim5creal(sim5complex a) { return creal(a); }
double sim5cimag(sim5complex a) { return cimag(a); }

void sort_roots(int *s, sim5complex *z1, sim5complex *z2, sim5complex *z3, sim5complex *z4)
{
    sim5complex in[4] = { *z1, *z2, *z3, *z4 }, out[4];
    int nreal = 0, k;
    for (int i = 0; i < 4; i++) if (cimag(in[i]) == 0.) out[nreal++] = in[i];          /* real roots first */
    k = nreal;
    for (int i = 0; i < 4; i++) if (cimag(in[i]) != 0.) out[k++] = in[i];              /* complex ones after, in their order */
    for (int i = 0; i < nreal; i++)                                                     /* the real ones descending */
        for (int j = 0; j < nreal - i; j++)
            if (creal(out[i + j]) > creal(out[i])) { sim5complex t = out[i + j]; out[i + j] = out[i]; out[i] = t; }
    *s = nreal;
    *z1 = out[0]; *z2 = out[1]; *z3 = out[2]; *z4 = out[3];
}
