/* shim_probe.c -- a small program written against the SIM5 scalar API (sim5_amd/host/sim5lib.h).
 * It traces an N x N thin-disk image ray by ray, the way SIM5 callers do, and prints one record per
 * pixel; tests/test_gpu_host_shim.py compares the records with the CPU oracle.
 *   usage: shim_probe <spin> <incl_deg> <N>
 */
#include "sim5lib.h"

#include <time.h>

/* `quiet`: (twice) the loop of ref examples/04-disk-image-eqplane/disk-image.c:53-105 as that program runs it -- results into two
 * float images, nothing printed per pixel -- timed by itself; prints the image's sums and the loop's wall-clock seconds */
static int quiet_image(double a, double inc, int N, int pass)
{
    const double rms = r_ms(a);
    const double rmax = rms + 8.0;
    disk_nt_setup(10.0, a, 0.1, 0.1, 0);
    float *image_f = (float *)calloc((size_t)N * N, sizeof(float)), *image_g = (float *)calloc((size_t)N * N, sizeof(float));
    long errors = 0, hits = 0;
    struct timespec t0, t1;
    if (pass > 0) { memset(image_f, 0, (size_t)N * N * sizeof(float)); memset(image_g, 0, (size_t)N * N * sizeof(float)); }
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int iy = 0; iy < N; iy++) for (int ix = 0; ix < N; ix++) {
        const double alpha = (((double)ix + .5) / (double)N - 0.5) * 2.0 * rmax;
        const double beta = (((double)iy + .5) / (double)N - 0.5) * 2.0 * rmax;
        geodesic gd;
        int error;
        geodesic_init_inf(inc, a, alpha, beta, &gd, &error);
        if (error) { errors++; continue; }
        for (int order = 0; order < 2; order++) {
            const double P = geodesic_find_midplane_crossing(&gd, order);
            if (isnan(P)) break;
            const double r = geodesic_position_rad(&gd, P);
            if (r >= rms) {
                const double g = gfactorK(r, a, gd.l), f = disk_nt_flux(r);
                image_f[ix + N * iy] = f * pow(g, 4.);
                image_g[ix + N * iy] = g;
                hits++;
                break;
            }
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double sg = 0.0, sf = 0.0;
    for (long i = 0; i < (long)N * N; i++) { sg += image_g[i]; sf += image_f[i]; }
    printf("# quiet pass %d rays %ld hits %ld errors %ld sum_g %.17g sum_f %.17g loop_seconds %.6f\n", pass, (long)N * N, hits, errors, sg, sf,
           (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
    free(image_f); free(image_g);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc == 5 && strcmp(argv[4], "quiet") == 0) {
        /* twice in one process: the second pass has the library's one-off set-up (staging memory, code load) behind it */
        for (int pass = 0; pass < 2; pass++) quiet_image(atof(argv[1]), deg2rad(atof(argv[2])), atoi(argv[3]), pass);
        return 0;
    }
    if (argc != 4) { fprintf(stderr, "usage: %s spin incl N [quiet]\n", argv[0]); return 2; }
    const double a = atof(argv[1]);
    const double inc = deg2rad(atof(argv[2]));
    const int N = atoi(argv[3]);
    const double rms = r_ms(a);
    const double rmax = rms + 8.0;
    disk_nt_setup(10.0, a, 0.1, 0.1, 0);
    printf("# rms %.17g rmin %.17g rbh %.17g\n", rms, disk_nt_r_min(), r_bh(a));
    for (int iy = 0; iy < N; iy++) {
        for (int ix = 0; ix < N; ix++) {
            const double alpha = (((double)ix + .5) / (double)N - 0.5) * 2.0 * rmax;
            const double beta = (((double)iy + .5) / (double)N - 0.5) * 2.0 * rmax;
            geodesic gd;
            int err = 0, hit = 0;
            double r = NAN, g = 0.0, f = 0.0;
            if (geodesic_init_inf(inc, a, alpha, beta, &gd, &err)) {
                for (int order = 0; order < 2 && !hit; order++) {
                    const double P = geodesic_find_midplane_crossing(&gd, order);
                    if (isnan(P)) break;
                    r = geodesic_position_rad(&gd, P);
                    if (r >= rms) {
                        g = gfactorK(r, a, gd.l);
                        f = disk_nt_flux(r);
                        hit = 1 + order;
                    }
                }
            }
            printf("%d %d %d %d %.17g %.17g %.17g\n", iy, ix, err, hit, hit ? r : 0.0, g, f);
        }
    }
    /* one step-wise ray through the same API */
    {
        geodesic gd; int err;
        geodesic_init_inf(inc, a, 3.0, 4.0, &gd, &err);
        double P0 = geodesic_P_int(&gd, 50.0, 0);
        double x[4] = { 0.0, 50.0, geodesic_position_pol(&gd, P0), 0.0 }, k[4];
        geodesic_momentum(&gd, P0, x[1], x[2], k);
        raytrace_data rtd;
        raytrace_prepare(a, x, k, 1.0, RTOPT_NONE, &rtd);
        int n = 0;
        while (x[1] > 1.05 * r_bh(a) && x[1] < 60.0 && rtd.error < 1e-2 && n < 5000) {
            double dl = 1e9;
            raytrace(x, k, &dl, &rtd);
            n++;
        }
        printf("# verlet steps %d r_end %.12g carter_err %.3e\n", n, x[1], raytrace_error(x, k, &rtd));
    }
    /* azimuth and light-travel time along one ray (SIM5 scalar API) */
    {
        geodesic gd; int err;
        geodesic_init_inf(inc, a, 6.0, 5.0, &gd, &err);
        const double P1 = 0.6 * gd.Rpc, P2 = 1.4 * gd.Rpc;
        const double r1 = geodesic_position_rad(&gd, P1), m1 = geodesic_position_pol(&gd, P1);
        printf("# azm %.17g delay %.17g\n", geodesic_position_azm(&gd, r1, m1, P1),
               geodesic_timedelay(&gd, P1, 0.0, 0.0, P2, 0.0, 0.0));
    }
    /* the integrals of sim5elliptic.h and vector_norm_to through the scalar API */
    {
        sim5metric mt; kerr_metric(a, 7.0, 0.3, &mt);
        double V[4] = { 0.0, 0.3, 0.1, 0.02 };
        vector_norm_to(V, 1.0, &mt);
        printf("# ints %.17g %.17g %.17g %.17g %.17g %.17g\n", elliptic_f_sin(0.7, 0.4), elliptic_pi_cos(0.35, -1.7, 0.62),
               integral_R_rp_cc2(4.0, 1.5, 0.8 + 1.1 * I, 1.2, 4.5, 30.0), integral_T_mp(3.0, 0.7, 1.0, -0.4),
               dotprod(V, V, &mt), V[1]);
    }
    return 0;
}
