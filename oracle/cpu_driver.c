/*
 * cpu_driver.c -- runs the thin-disk pixel loop on the host cores through either CPU
 * checker library and hands back full-precision per-pixel records.
 *
 * TEST INFRASTRUCTURE ONLY (golden-vector generation, parity tests, and the
 * cpu_baseline leg of bench.py).  The loop body is the caller loop of the reference
 * example (ref: examples/04-disk-image-eqplane/disk-image.c:53-105) written against
 * function pointers, so the same driver times
 *   kind 0: oracle/_ref/libsim5ref.so  (the unmodified reference: SIM5 symbol names)
 *   kind 1: oracle/liboracle.so        (our restatement: orc_disk_pixel)
 * Rows are handed to pthreads through an atomic counter; the reference's disk_nt_*
 * state is process-global but read-only after disk_nt_setup, so concurrent reads are safe
 * (ref: src/sim5disk-nt.c:17-32).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* mirror of struct geodesic (240 B; ref: src/sim5kerr-geod.h:42-68) */
typedef struct {
    double a, alpha, beta, incl, cos_i, l, q;
    double roots[8];
    int nrr, type;
    double m2p, m2m, mm, mK, rp, dmdp_inf, Rpc, Tpp, Tip, k[4], p;
} geod_t;

typedef struct { float mass, spin, mdot, rms, alpha; int options; } disk_t;
typedef struct {
    int cls, gtype, err;
    double r, g, flux;
    float image_f, image_g;
} pixel_t;

typedef struct {
    int kind;
    /* reference entry points */
    int (*init_inf)(double, double, double, double, geod_t *, int *);
    double (*midplane)(geod_t *, int);
    double (*pos_rad)(geod_t *, double);
    double (*gfac)(double, double, double);
    double (*flux)(double);
    /* oracle entry point */
    void (*pixel)(const disk_t *, double, double, double, double, double, pixel_t *);
    disk_t disk;
    /* job */
    int nx, ny, y0, y1, ystride, xstride;
    double a, inc, rms, rmax;
    float *img_f, *img_g;
    unsigned char *cls;
    signed char *gtype;
    double *r, *g, *fl;
    atomic_int next_row;
} job_t;

static void one_pixel(job_t *J, int ix, int iy)
{
    /* impact parameters: ref disk-image.c:57-58 */
    double alpha = (((double)(ix) + .5) / (double)(J->nx) - 0.5) * 2.0 * J->rmax;
    double beta = (((double)(iy) + .5) / (double)(J->ny) - 0.5) * 2.0 * J->rmax * ((double)J->ny / (double)J->nx);
    pixel_t px;
    if (J->kind == 1) {
        J->pixel(&J->disk, J->inc, J->a, J->rms, alpha, beta, &px);
    } else {
        geod_t gd;
        int err = 0;
        px.cls = 0; px.gtype = -1; px.r = NAN; px.g = 0; px.flux = 0; px.image_f = 0; px.image_g = 0;
        J->init_inf(J->inc, J->a, alpha, beta, &gd, &err);
        px.err = err;
        if (!err) {
            px.gtype = gd.type;
            px.cls = 5;
            for (int order = 0; order < 2; order++) {
                double P = J->midplane(&gd, order);
                if (isnan(P)) { px.cls = order ? 3 : 1; break; }
                double r = J->pos_rad(&gd, P);
                if (r >= J->rms) {
                    double g = J->gfac(r, J->a, gd.l);
                    double f = J->flux(r);
                    px.cls = order ? 4 : 2;
                    px.r = r; px.g = g; px.flux = f;
                    px.image_f = f * pow(g, 4.);
                    px.image_g = g;
                    break;
                }
            }
        }
    }
    /* packed output index: sampled rows/cols only */
    size_t ox = (size_t)(ix / J->xstride), oy = (size_t)((iy - J->y0) / J->ystride);
    size_t onx = (size_t)((J->nx + J->xstride - 1) / J->xstride);
    size_t o = oy * onx + ox;
    if (J->img_f) J->img_f[o] = px.image_f;
    if (J->img_g) J->img_g[o] = px.image_g;
    if (J->cls) J->cls[o] = (unsigned char)px.cls;
    if (J->gtype) J->gtype[o] = (signed char)px.gtype;
    if (J->r) J->r[o] = px.r;
    if (J->g) J->g[o] = px.g;
    if (J->fl) J->fl[o] = px.flux;
}

static void *worker(void *arg)
{
    job_t *J = (job_t *)arg;
    int nrows = (J->y1 - J->y0 + J->ystride - 1) / J->ystride;
    for (;;) {
        int k = atomic_fetch_add(&J->next_row, 1);
        if (k >= nrows) break;
        int iy = J->y0 + k * J->ystride;
        for (int ix = 0; ix < J->nx; ix += J->xstride) one_pixel(J, ix, iy);
    }
    return 0;
}

/*
 * Trace rows y0 <= iy < y1 (step ystride), columns 0 <= ix < nx (step xstride) of the
 * nx*ny image; outputs are packed arrays over the sampled pixels (any may be NULL).
 * Returns 0 on success; *seconds = wall time of the pixel loop (CLOCK_MONOTONIC).
 */
int cpu_disk_image(const char *libpath, int kind, int nx, int ny, double a, double inc_rad,
                   double M, double mdot, double alpha_visc,
                   int y0, int y1, int ystride, int xstride, int nthreads,
                   float *img_f, float *img_g, unsigned char *cls, signed char *gtype,
                   double *r, double *g, double *flux, double *seconds)
{
    void *h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "cpu_driver: %s\n", dlerror()); return -1; }
    job_t J;
    memset(&J, 0, sizeof J);
    J.kind = kind;
    double (*r_ms)(double) = (double (*)(double))dlsym(h, kind ? "orc_r_ms" : "r_ms");
    if (!r_ms) return -2;
    if (kind == 1) {
        void (*setup)(disk_t *, double, double, double, double) = dlsym(h, "orc_disk_nt_setup");
        J.pixel = dlsym(h, "orc_disk_pixel");
        if (!setup || !J.pixel) return -2;
        setup(&J.disk, M, a, mdot, alpha_visc);
    } else {
        int (*setup)(double, double, double, double, int) = dlsym(h, "disk_nt_setup");
        J.init_inf = dlsym(h, "geodesic_init_inf");
        J.midplane = dlsym(h, "geodesic_find_midplane_crossing");
        J.pos_rad = dlsym(h, "geodesic_position_rad");
        J.gfac = dlsym(h, "gfactorK");
        J.flux = dlsym(h, "disk_nt_flux");
        if (!setup || !J.init_inf || !J.midplane || !J.pos_rad || !J.gfac || !J.flux) return -2;
        setup(M, a, mdot, alpha_visc, 0);
    }
    J.nx = nx; J.ny = ny; J.y0 = y0; J.y1 = y1;
    J.ystride = ystride < 1 ? 1 : ystride;
    J.xstride = xstride < 1 ? 1 : xstride;
    J.a = a; J.inc = inc_rad;
    J.rms = r_ms(a);                      /* ref disk-image.c:41-42 */
    J.rmax = J.rms + 8.0;
    J.img_f = img_f; J.img_g = img_g; J.cls = cls; J.gtype = gtype; J.r = r; J.g = g; J.fl = flux;
    atomic_init(&J.next_row, 0);

    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (nthreads == 1) {
        worker(&J);
    } else {
        for (int i = 0; i < nthreads; i++) pthread_create(&th[i], 0, worker, &J);
        for (int i = 0; i < nthreads; i++) pthread_join(th[i], 0);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (seconds) *seconds = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    return 0;
}
