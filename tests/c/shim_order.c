/* shim_order.c -- the SIM5 scalar API (sim5_amd/host/sim5lib.h) asked for the SAME pixels in different orders: raster (what
 * the shim's look-ahead expects), column-major, shuffled, raster with every third pixel skipped, and two images interleaved
 * row by row.  Whatever the order, a pixel's record must be the same text: the look-ahead answers a call only from a record
 * made for exactly its arguments (tests/test_gpu_host_shim.py compares the outputs of the orders, and of the look-ahead
 * switched off).
 *   usage: shim_order <spin> <incl_deg> <NX> <NY> <order: 0 raster, 1 column-major, 2 shuffled, 3 skipping, 4 interleaved, 5 second image of 4 alone> */
#include "sim5lib.h"

static double rms_, rmax_, a_, inc_;
static int NX, NY;

static void pixel(int ix, int iy, double a, char *line)
{
    const double alpha = (((double)ix + .5) / (double)NX - 0.5) * 2.0 * rmax_;
    const double beta = (((double)iy + .5) / (double)NY - 0.5) * 2.0 * rmax_ * ((double)NY / (double)NX);
    geodesic gd;
    int err = 0, hit = 0;
    double r = NAN, g = 0.0, f = 0.0;
    if (geodesic_init_inf(inc_, a, alpha, beta, &gd, &err)) {
        for (int order = 0; order < 2 && !hit; order++) {
            const double P = geodesic_find_midplane_crossing(&gd, order);
            if (isnan(P)) break;
            r = geodesic_position_rad(&gd, P);
            if (r >= rms_) { g = gfactorK(r, a, gd.l); f = disk_nt_flux(r); hit = 1 + order; }
        }
    }
    sprintf(line, "%d %d %d %d %.17g %.17g %.17g %.17g %.17g", iy, ix, err, hit, hit ? r : 0.0, g, f, gd.l, gd.q);
}

int main(int argc, char **argv)
{
    if (argc != 6) { fprintf(stderr, "usage: %s spin incl NX NY order\n", argv[0]); return 2; }
    a_ = atof(argv[1]); inc_ = deg2rad(atof(argv[2])); NX = atoi(argv[3]); NY = atoi(argv[4]);
    const int order = atoi(argv[5]);
    rms_ = r_ms(a_); rmax_ = rms_ + 8.0;
    disk_nt_setup(10.0, a_, 0.1, 0.1, 0);
    const int n = NX * NY;
    char (*lines)[200] = calloc((size_t)n, 200);
    char (*lines2)[200] = calloc((size_t)n, 200);
    int *seq = malloc(sizeof(int) * (size_t)n);
    for (int k = 0; k < n; k++) seq[k] = k;
    if (order == 1) for (int k = 0; k < n; k++) seq[k] = (k % NY) * NX + k / NY;
    if (order == 2) { unsigned s = 12345u; for (int k = n - 1; k > 0; k--) { s = s * 1664525u + 1013904223u; int j = (int)((s >> 8) % (unsigned)(k + 1)); int t = seq[k]; seq[k] = seq[j]; seq[j] = t; } }
    if (order == 4) {
        /* two images (two spins) asked for row by row in turn: the book of rows sees (i, a) change at every row */
        for (int iy = 0; iy < NY; iy++) {
            for (int ix = 0; ix < NX; ix++) pixel(ix, iy, a_, lines[iy * NX + ix]);
            for (int ix = 0; ix < NX; ix++) pixel(ix, iy, a_ * 0.5, lines2[iy * NX + ix]);
        }
    } else if (order == 5) {
        /* the second image of order 4 on its own (same field of view and disk, half the spin) */
        for (int k = 0; k < n; k++) pixel(k % NX, k / NX, a_ * 0.5, lines[k]);
    } else {
        for (int k = 0; k < n; k++) {
            if (order == 3 && (seq[k] % 3) == 1) { sprintf(lines[seq[k]], "skipped"); continue; }
            pixel(seq[k] % NX, seq[k] / NX, a_, lines[seq[k]]);
        }
        if (order == 3) for (int k = 0; k < n; k++) if ((k % 3) == 1) pixel(k % NX, k / NX, a_, lines[k]);
    }
    for (int k = 0; k < n; k++) puts(lines[k]);
    if (order == 4) for (int k = 0; k < n; k++) printf("second %s\n", lines2[k]);
    return 0;
}
