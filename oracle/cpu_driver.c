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

/*
 * disk_nt_flux of either checker library at n given radii (ref: src/sim5disk-nt.c:110-146), after its own
 * disk_nt_setup(M, a, mdot, alpha, 0).  The parity tests use it to hold the GPU's flux to the reference's AT THE GPU's OWN
 * RADII -- the same input bits -- where the closed form cancels to its rounding pattern (inner edge of the disk).
 */
int cpu_disk_flux(const char *libpath, int kind, double M, double a, double mdot, double alpha_visc,
                  long n, const double *r, double *flux)
{
    void *h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "cpu_driver: %s\n", dlerror()); return -1; }
    if (kind == 1) {
        disk_t disk;
        void (*setup)(disk_t *, double, double, double, double) = dlsym(h, "orc_disk_nt_setup");
        double (*fl)(const disk_t *, double) = dlsym(h, "orc_disk_nt_flux");
        if (!setup || !fl) return -2;
        memset(&disk, 0, sizeof disk);
        setup(&disk, M, a, mdot, alpha_visc);
        for (long i = 0; i < n; i++) flux[i] = fl(&disk, r[i]);
    } else {
        int (*setup)(double, double, double, double, int) = dlsym(h, "disk_nt_setup");
        double (*fl)(double) = dlsym(h, "disk_nt_flux");
        if (!setup || !fl) return -2;
        setup(M, a, mdot, alpha_visc, 0);
        for (long i = 0; i < n; i++) flux[i] = fl(r[i]);
    }
    return 0;
}

/* ====================================================================================== */
/*  Recipes assembled from public SIM5 routines, run through either checker library        */
/*  (prefix "" = reference symbols, "orc_" = our restatement; same signatures).            */
/* ====================================================================================== */
typedef struct { double a, r, m, g00, g11, g22, g33, g03; } metric_t;
typedef struct { double e[4][4]; metric_t metric; } tetrad_t;
typedef struct { double re, im; } cplx_t;
typedef struct {
    int opt_gr, opt_pol; double step_epsilon, bh_spin, E, Q; cplx_t WP;
    int pass, refines; double dk[4], df[4], kt; float error;
} rtd_t;

typedef struct {
    int (*init_inf)(double, double, double, double, geod_t *, int *);
    double (*midplane)(geod_t *, int);
    double (*pos_rad)(geod_t *, double);
    double (*pos_pol)(geod_t *, double);
    double (*P_int)(geod_t *, double, int);
    void (*momentum)(geod_t *, double, double, double, double *);
    void (*kerr_metric)(double, double, double, metric_t *);
    void (*tetrad_azimuthal)(metric_t *, double, tetrad_t *);
    void (*bl2on)(double *, double *, tetrad_t *);
    void (*on2bl)(double *, double *, tetrad_t *);
    void (*norm_to)(double *, double, metric_t *);
    double (*OmegaK)(double, double);
    double (*gfac)(double, double, double);
    double _Complex (*pol_const)(double *, double *, metric_t *);
    double (*pol_rot)(double, double, double, double, double _Complex);
    void (*rt_prepare)(double, double *, double *, double, int, rtd_t *);
    void (*rt_step)(double *, double *, double *, rtd_t *);
    double (*rt_error)(double *, double *, rtd_t *);
    double (*r_ms)(double);
    void (*flat_metric)(double, double, metric_t *);
    double (*Omega_from_ell)(double, metric_t *);
} api_t;

static void *sym(void *h, const char *prefix, const char *name)
{
    char buf[128];
    snprintf(buf, sizeof buf, "%s%s", prefix, name);
    return dlsym(h, buf);
}

static int load_api(const char *libpath, const char *prefix, api_t *A, void **hout)
{
    void *h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "cpu_driver: %s\n", dlerror()); return -1; }
    A->init_inf = sym(h, prefix, "geodesic_init_inf");
    A->midplane = sym(h, prefix, "geodesic_find_midplane_crossing");
    A->pos_rad = sym(h, prefix, "geodesic_position_rad");
    A->pos_pol = sym(h, prefix, "geodesic_position_pol");
    A->P_int = sym(h, prefix, "geodesic_P_int");
    A->momentum = sym(h, prefix, "geodesic_momentum");
    A->kerr_metric = sym(h, prefix, "kerr_metric");
    A->tetrad_azimuthal = sym(h, prefix, "tetrad_azimuthal");
    A->bl2on = sym(h, prefix, "bl2on");
    A->on2bl = sym(h, prefix, "on2bl");
    A->norm_to = sym(h, prefix, "vector_norm_to");
    A->OmegaK = sym(h, prefix, "OmegaK");
    A->gfac = sym(h, prefix, "gfactorK");
    A->pol_const = sym(h, prefix, "polarization_constant");
    A->pol_rot = sym(h, prefix, "polarization_angle_rotation");
    A->rt_prepare = sym(h, prefix, "raytrace_prepare");
    A->rt_step = sym(h, prefix, "raytrace");
    A->rt_error = sym(h, prefix, "raytrace_error");
    A->r_ms = sym(h, prefix, "r_ms");
    A->flat_metric = sym(h, prefix, "flat_metric");
    A->Omega_from_ell = sym(h, prefix, "Omega_from_ell");
    void **p = (void **)A;
    for (size_t i = 0; i < sizeof(api_t) / sizeof(void *); i++)
        if (!p[i]) { fprintf(stderr, "cpu_driver: missing symbol #%zu in %s\n", i, libpath); return -2; }
    *hout = h;
    return 0;
}

/*
 * Polarization angle at infinity for thin-disk pixels (SURVEY.md 3.4 recipe; the routines are
 * ref src/sim5kerr-geod.c:787, src/sim5kerr.c:75,766,926,948,553,1037, src/sim5polarization.c:145,272).
 * For each of n rays (alpha[i], beta[i]): chi[i] (NaN if the ray does not hit the disk), r, g,
 * and the Walker-Penrose constant (2 doubles).
 */
int cpu_polarized_rays(const char *libpath, const char *prefix, double a, double inc_rad, double rms,
                       int n, const double *alpha, const double *beta,
                       double *chi, double *r_out, double *g_out, double *wp_out)
{
    api_t A; void *h;
    int rc = load_api(libpath, prefix, &A, &h);
    if (rc) return rc;
    if (rms <= 0) rms = A.r_ms(a);
    for (int i = 0; i < n; i++) {
        geod_t gd; int err = 0;
        chi[i] = NAN; r_out[i] = NAN; g_out[i] = 0.0; wp_out[2 * i] = wp_out[2 * i + 1] = NAN;
        A.init_inf(inc_rad, a, alpha[i], beta[i], &gd, &err);
        if (err) continue;
        for (int order = 0; order < 2; order++) {
            double P = A.midplane(&gd, order);
            if (isnan(P)) break;
            double r = A.pos_rad(&gd, P);
            if (r >= rms) {
                double k[4], nloc[4], floc[4], f[4];
                metric_t mt; tetrad_t t;
                A.momentum(&gd, P, r, 0.0, k);
                A.kerr_metric(a, r, 0.0, &mt);
                A.tetrad_azimuthal(&mt, A.OmegaK(r, a), &t);
                A.bl2on(k, nloc, &t);
                floc[0] = 0.0; floc[1] = nloc[3]; floc[2] = 0.0; floc[3] = -nloc[1];
                A.on2bl(floc, f, &t);
                A.norm_to(f, 1.0, &mt);
                double _Complex wp = A.pol_const(k, f, &mt);
                chi[i] = A.pol_rot(a, inc_rad, alpha[i], beta[i], wp);
                r_out[i] = r;
                g_out[i] = A.gfac(r, a, gd.l);
                wp_out[2 * i] = __real__ wp; wp_out[2 * i + 1] = __imag__ wp;
                break;
            }
        }
    }
    return 0;
}

/*
 * Step-wise integration of one ray from radius r0 on the incoming branch (SURVEY.md 8(d) C4
 * start-up: geodesic_init_inf -> geodesic_P_int -> geodesic_position_pol -> geodesic_momentum ->
 * raytrace_prepare), recording after every raytrace() call: x[4], k[4], dl, rtd.error, rtd.kt.
 * trace is nmax x 11 doubles; returns the number of steps made (or <0 on error) and the final
 * raytrace_error() in *carter.  Stops when r <= r_in, r >= r_out, error > max_error or nmax steps.
 */
int cpu_verlet_trace(const char *libpath, const char *prefix, double a, double inc_rad,
                     double alpha, double beta, double r0, double precision, int options,
                     double dl_max, double r_in, double r_out, double max_error, int nmax,
                     double *trace, double *x_start, double *k_start, double *carter)
{
    api_t A; void *h;
    int rc = load_api(libpath, prefix, &A, &h);
    if (rc) return rc;
    geod_t gd; int err = 0;
    A.init_inf(inc_rad, a, alpha, beta, &gd, &err);
    if (err) return -10 - err;
    if (!(r0 > gd.rp)) return -3;
    double P0 = A.P_int(&gd, r0, 0);
    double x[4] = { 0.0, r0, 0.0, 0.0 }, k[4];
    x[2] = A.pos_pol(&gd, P0);
    A.momentum(&gd, P0, r0, x[2], k);
    if (isnan(k[0])) return -4;
    for (int c = 0; c < 4; c++) { x_start[c] = x[c]; k_start[c] = k[c]; }
    rtd_t rtd;
    memset(&rtd, 0, sizeof rtd);
    A.rt_prepare(a, x, k, precision, options, &rtd);
    int n = 0;
    while (n < nmax) {
        double dl = dl_max;
        A.rt_step(x, k, &dl, &rtd);
        double *row = trace + (size_t)n * 11;
        for (int c = 0; c < 4; c++) { row[c] = x[c]; row[4 + c] = k[c]; }
        row[8] = dl; row[9] = rtd.error; row[10] = rtd.kt;
        n++;
        if (!(x[1] > r_in) || !(x[1] < r_out) || (rtd.error > max_error)) break;
    }
    if (carter) *carter = A.rt_error(x, k, &rtd);
    return n;
}


/*
 * The C4 job on the host: n rays (alpha[i], beta[i]) started at r0 as in cpu_verlet_trace, advanced by the
 * library's raytrace() until they leave (r_in, r_out), rtd.error > max_error or max_steps calls were made, with
 * the radiative transfer through the torus accumulated after every call.  The integrator is the checker
 * library's (the unmodified reference with prefix "", our restatement with "orc_"); the transfer model is this
 * project's own (the reference has none) and is restated here in plain C from its
 * definition in DESIGN.md section 7: fluid on circular orbits of constant specific angular momentum ell
 * (Omega = Omega_from_ell, ref src/sim5kerr.c:1101), density rho = exp(-((R-R_t)^2+z^2)/(2 w^2)) cut at
 * 36 * 2w^2 (shape 0) or 1 inside r <= w (shape 1), g = E_inf / (-k.U), and per step of affine length dl
 *      dtau = absorb0 rho dl/g ,   dI = g^4 emis0 rho exp(-tau) dl/g .
 * Outputs per ray (any may be NULL): steps (0 = the ray could not be started), x_end[4], k_end[4], I, tau,
 * raytrace_error() at the end, largest rtd.error seen.
 */
static double shift_ulps(double v, int ulps)
{
    for (; ulps > 0; ulps--) v = nextafter(v, INFINITY);
    for (; ulps < 0; ulps++) v = nextafter(v, -INFINITY);
    return v;
}

/* `ulps` (NULL or 8 ints): the start state (x[0..3], k[0..3]) is moved by that many units in the last place before
 * raytrace_prepare() -- the conditioning probe of the parity tests: how far does the checker's OWN end state move when
 * its start state changes in the last bit? */
static int torus_rays_impl(const char *libpath, const char *prefix, double a, double inc_rad, int n,
                   const double *alpha, const double *beta, double r0, double precision, int options,
                   double dl_max, double r_in, double r_out, double max_error, int max_steps,
                   int shape, double torus_r, double torus_w, double torus_l, double emis0, double absorb0,
                   int *steps, double *x_end, double *k_end, double *I_out, double *tau_out,
                   double *carter, float *max_step_error, const int *ulps)
{
    api_t A; void *h;
    int rc = load_api(libpath, prefix, &A, &h);
    if (rc) return rc;
    for (int i = 0; i < n; i++) {
        geod_t gd; int err = 0;
        double x[4] = { 0.0, r0, 0.0, 0.0 }, k[4] = { 0.0, 0.0, 0.0, 0.0 };
        double I = 0.0, tau = 0.0, car = NAN;
        float worst = 0.0f;
        int made = 0;
        A.init_inf(inc_rad, a, alpha[i], beta[i], &gd, &err);
        if (!err && r0 > gd.rp) {
            double P0 = A.P_int(&gd, r0, 0);
            x[2] = A.pos_pol(&gd, P0);
            A.momentum(&gd, P0, r0, x[2], k);
            if (!isnan(k[0]) && !isnan(x[2])) {
                rtd_t rtd;
                memset(&rtd, 0, sizeof rtd);
                if (ulps) for (int c = 0; c < 4; c++) { x[c] = shift_ulps(x[c], ulps[c]); k[c] = shift_ulps(k[c], ulps[4 + c]); }
                A.rt_prepare(a, x, k, precision, options, &rtd);
                while (made < max_steps) {
                    double dl = dl_max;
                    A.rt_step(x, k, &dl, &rtd);
                    made++;
                    if (rtd.error > worst) worst = rtd.error;
                    /* transfer over the step just taken, evaluated at its end point */
                    double rho;
                    if (shape == 1) rho = (x[1] <= torus_w) ? 1.0 : 0.0;
                    else {
                        double R = x[1] * sqrt(1. - x[2] * x[2]), z = x[1] * x[2];
                        double d2 = (R - torus_r) * (R - torus_r) + z * z, w2 = 2. * torus_w * torus_w;
                        rho = (d2 < 36. * w2) ? exp(-d2 / w2) : 0.0;
                    }
                    if (rho > 0.0) {
                        metric_t g;
                        if (rtd.opt_gr) A.kerr_metric(a, x[1], x[2], &g); else A.flat_metric(x[1], x[2], &g);
                        double Om = A.Omega_from_ell(torus_l, &g);
                        double nrm = -(g.g00 + 2. * Om * g.g03 + Om * Om * g.g33);
                        if (nrm > 0.0) {
                            double ut = 1. / sqrt(nrm);
                            double k_t = k[0] * g.g00 + k[3] * g.g03, k_f = k[3] * g.g33 + k[0] * g.g03;
                            double gfac = rtd.E / (ut * (k_t + Om * k_f));
                            double ds = dl / gfac, g2 = gfac * gfac;
                            double att = (absorb0 == 0.0) ? 1.0 : exp(-tau);
                            I += (g2 * g2) * emis0 * rho * att * ds;
                            tau += absorb0 * rho * ds;
                        }
                    }
                    if (!(x[1] > r_in) || !(x[1] < r_out) || (rtd.error > max_error)) break;
                }
                car = A.rt_error(x, k, &rtd);
            }
        }
        if (steps) steps[i] = made;
        if (x_end) for (int c = 0; c < 4; c++) x_end[4 * i + c] = x[c];
        if (k_end) for (int c = 0; c < 4; c++) k_end[4 * i + c] = k[c];
        if (I_out) I_out[i] = I;
        if (tau_out) tau_out[i] = tau;
        if (carter) carter[i] = car;
        if (max_step_error) max_step_error[i] = worst;
    }
    return 0;
}

int cpu_torus_rays(const char *libpath, const char *prefix, double a, double inc_rad, int n,
                   const double *alpha, const double *beta, double r0, double precision, int options,
                   double dl_max, double r_in, double r_out, double max_error, int max_steps,
                   int shape, double torus_r, double torus_w, double torus_l, double emis0, double absorb0,
                   int *steps, double *x_end, double *k_end, double *I_out, double *tau_out,
                   double *carter, float *max_step_error)
{
    return torus_rays_impl(libpath, prefix, a, inc_rad, n, alpha, beta, r0, precision, options, dl_max, r_in, r_out,
                           max_error, max_steps, shape, torus_r, torus_w, torus_l, emis0, absorb0, steps, x_end, k_end,
                           I_out, tau_out, carter, max_step_error, 0);
}

int cpu_torus_rays_perturbed(const char *libpath, const char *prefix, double a, double inc_rad, int n,
                   const double *alpha, const double *beta, double r0, double precision, int options,
                   double dl_max, double r_in, double r_out, double max_error, int max_steps,
                   int shape, double torus_r, double torus_w, double torus_l, double emis0, double absorb0,
                   int *steps, double *x_end, double *k_end, double *I_out, double *tau_out,
                   double *carter, float *max_step_error, const int *ulps)
{
    return torus_rays_impl(libpath, prefix, a, inc_rad, n, alpha, beta, r0, precision, options, dl_max, r_in, r_out,
                           max_error, max_steps, shape, torus_r, torus_w, torus_l, emis0, absorb0, steps, x_end, k_end,
                           I_out, tau_out, carter, max_step_error, ulps);
}
