/* stub_sim5gpu.c -- TEST DOUBLE for libsim5gpu.so, CPU only (tests/test_host_shim.py).  NOT a CPU implementation of the
 * library and not an oracle: the "physics" below is made-up arithmetic (deterministic functions of the arguments), good for
 * one thing -- running the HOST side of the SIM5 scalar API (sim5_amd/host/sim5lib.c: per-thread records, look-ahead
 * book-keeping, lazy symbol resolution, the disk-model generation stamp) under ThreadSanitizer and in a container without a
 * GPU.  It exports exactly the entry points that path resolves, with the signatures of include/sim5gpu.h. */
#include <math.h>
#include <stdatomic.h>
#include <stddef.h>
#include <string.h>
#include "../../include/sim5gpu.h"

static _Atomic unsigned long generation;
static _Atomic double disk_scale;                 /* "mdot / M" of the model set last */
static _Atomic long calls;                        /* batch calls served (the tests read it: look-ahead makes few) */

const char *sim5gpu_last_error(void) { return "stub"; }
long stub_batch_calls(void) { return atomic_load(&calls); }

int sim5gpu_disk_nt_setup(double M, double a, double mdot_or_L, double alpha, int options)
{
    (void)a; (void)alpha; (void)options;
    atomic_store(&disk_scale, mdot_or_L / M);
    atomic_fetch_add(&generation, 1ul);
    return 0;
}
unsigned long sim5gpu_disk_nt_generation(void) { return atomic_load(&generation); }

static double fake_flux(double r) { return atomic_load(&disk_scale) * 1e20 / (r * r * r); }
static double fake_P(const sim5gpu_geodesic *g, int order) { return (g->q > 3.0 + order) ? 0.1 * (order + 1) * g->Rpc : NAN; }
static double fake_r(const sim5gpu_geodesic *g, double P) { return 2.0 + fabs(g->alpha) + 10.0 * P / g->Rpc; }
static double fake_gK(double r, double a, double l) { return sqrt(1.0 - 2.0 / (r + 4.0)) / (1.0 + 0.01 * a * l); }

static int fake_geodesic(double i, double a, double alpha, double beta, sim5gpu_geodesic *g, int *err)
{
    /* writes what geodesic_init_inf writes; leaves dmdp_inf, k, p alone (like the device routine) */
    g->a = a < 1e-4 ? 1e-4 : a; g->alpha = alpha; g->beta = beta == 0.0 ? 1e-6 : beta; g->incl = i; g->cos_i = cos(i);
    g->l = -alpha * sin(i); g->q = beta * beta + cos(i) * cos(i) * (alpha * alpha - a * a);
    g->nrr = 2; g->type = 2; g->m2p = 0.5; g->m2m = 0.25; g->mm = 0.1; g->mK = 1.0; g->rp = 1.0 + fabs(alpha);
    g->Rpc = 3.0 + 0.01 * fabs(beta); g->Tpp = 2.0; g->Tip = 0.5 + 0.001 * alpha;
    *err = (g->q <= 0.0) ? 1 : 0;
    return *err == 0;
}

int sim5gpu_geodesic_init_inf(size_t n, const double *incl, const double *a, const double *alpha, const double *beta,
                              sim5gpu_geodesic *g, int *error, int *ok)
{
    atomic_fetch_add(&calls, 1);
    for (size_t k = 0; k < n; k++) ok[k] = fake_geodesic(incl[k], a[k], alpha[k], beta[k], &g[k], &error[k]);
    return 0;
}

int sim5gpu_geodesic_init_inf_chain(size_t n, const double *incl, const double *a, const double *alpha, const double *beta,
                                    sim5gpu_geodesic *g, int *error, int *ok, sim5gpu_geodesic_chain *c)
{
    atomic_fetch_add(&calls, 1);
    for (size_t k = 0; k < n; k++) {
        ok[k] = fake_geodesic(incl[k], a[k], alpha[k], beta[k], &g[k], &error[k]);
        memset(&c[k], 0, sizeof c[k]);
        c[k].valid = ok[k]; c[k].a = a[k]; c[k].l = g[k].l; c[k].flux_valid = atomic_load(&generation) > 0;
        for (int o = 0; o < 2 && ok[k]; o++) {
            c[k].P[o] = fake_P(&g[k], o);
            c[k].r[o] = NAN;
            if (!isnan(c[k].P[o])) {
                c[k].r[o] = fake_r(&g[k], c[k].P[o]); c[k].have_r[o] = 1;
                c[k].g[o] = fake_gK(c[k].r[o], a[k], g[k].l); c[k].flux[o] = fake_flux(c[k].r[o]);
            }
        }
    }
    return 0;
}

int sim5gpu_geodesic_find_midplane_crossing(size_t n, const sim5gpu_geodesic *g, const int *order, double *P)
{
    atomic_fetch_add(&calls, 1);
    for (size_t k = 0; k < n; k++) P[k] = fake_P(&g[k], order[k]);
    return 0;
}
int sim5gpu_geodesic_position_rad(size_t n, const sim5gpu_geodesic *g, const double *P, double *r)
{
    atomic_fetch_add(&calls, 1);
    for (size_t k = 0; k < n; k++) r[k] = fake_r(&g[k], P[k]);
    return 0;
}
int sim5gpu_gfactorK(size_t n, const double *r, const double *a, const double *l, double *out)
{
    atomic_fetch_add(&calls, 1);
    for (size_t k = 0; k < n; k++) out[k] = fake_gK(r[k], a[k], l[k]);
    return 0;
}
int sim5gpu_disk_nt_flux(size_t n, const double *r, double *out)
{
    atomic_fetch_add(&calls, 1);
    for (size_t k = 0; k < n; k++) out[k] = fake_flux(r[k]);
    return 0;
}
int sim5gpu_r_ms(size_t n, const double *a, double *out) { for (size_t k = 0; k < n; k++) out[k] = 6.0 - 4.7 * a[k]; return 0; }
