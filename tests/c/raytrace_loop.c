/* raytrace_loop.c -- the other scalar loop BASELINE.json's north_star names: raytrace_prepare / raytrace one call at a time
 * (ref README.md:184-193; the timing loop of ref src/sim5unittests.c:116-127: precision 0.01, dl = 1e9 at every call, stop when
 * the ray leaves (r_min, r_max) or rtd.error > 1e-3).  A handful of rays started at r0 = 50 on the way in; the loop is timed
 * inside the program (CLOCK_MONOTONIC) and the end state of every ray is printed, so two builds of this file -- over the host
 * shim (the GPU) and over the reference library (one CPU core) -- can be compared value by value and call by call.
 *   usage: raytrace_loop <spin> <incl_deg> <rays> [quiet] */
#include <time.h>
#include "sim5lib.h"

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s spin incl rays [quiet]\n", argv[0]); return 2; }
    const double a = atof(argv[1]), inc = deg2rad(atof(argv[2]));
    const int rays = atoi(argv[3]);
    const double r0 = 50.0, r_min = 1.05 * r_bh(a), r_max = 60.0;
    long calls = 0;
    double secs = 0.0;
    for (int j = 0; j < rays; j++) {
        const double alpha = -7.0 + 14.0 * (j + 0.5) / rays, beta = 2.0 + 0.37 * j;
        geodesic gd; int err = 0;
        if (!geodesic_init_inf(inc, a, alpha, beta, &gd, &err)) { printf("# ray %d rejected by geodesic_init_inf: %d\n", j, err); continue; }
        const double P0 = geodesic_P_int(&gd, r0, 0);
        double x[4] = { 0.0, r0, geodesic_position_pol(&gd, P0), 0.0 }, k[4];
        geodesic_momentum(&gd, P0, x[1], x[2], k);
        raytrace_data rtd;
        memset(&rtd, 0, sizeof rtd);
        raytrace_prepare(a, x, k, 0.01, RTOPT_NONE, &rtd);
        struct timespec t0, t1;
        int n = 0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        while (1) {
            double dl = 1e9;                                  /* use maximal step */
            raytrace(x, k, &dl, &rtd);
            n++;
            if ((x[1] < r_min) || (x[1] > r_max)) break;
            if (rtd.error > 1e-3) break;
            if (n >= 200000) break;
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        secs += (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
        calls += n;
        if (argc < 5) printf("%d %d %.17g %.17g %.17g %.17g %.17g %.17g %.9g\n", j, n, x[0], x[1], x[2], x[3], k[1], k[2], raytrace_error(x, k, &rtd));
        else printf("%d %d %.12g\n", j, n, x[1]);
    }
    printf("# raytrace loop: rays %d calls %ld loop_seconds %.6f calls_per_s %.6g\n", rays, calls, secs, calls / secs);
    return 0;
}
