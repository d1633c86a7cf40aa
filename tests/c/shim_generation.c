/* shim_generation.c -- ADVICE r5 (medium): a record of the scalar API must not answer disk_nt_flux after the disk model has been
 * changed BESIDE the shim (sim5gpu_disk_nt_setup called directly: ctypes, DiskModel_ThinDisk, another library of the process).
 * Runs against the real library on the GPU box and against the test double tests/c/stub_sim5gpu.c on the CPU.
 *   usage: shim_generation   -> prints "ok" and returns 0, or the two fluxes that should differ */
#include "sim5lib.h"
#include "../../include/sim5gpu.h"

int main(void)
{
    const double a = 0.9, inc = deg2rad(60.0);
    disk_nt_setup(10.0, a, 0.1, 0.1, 0);
    const double rmax = r_ms(a) + 8.0;
    int bad = 0, tried = 0;
    for (int iy = 0; iy < 8; iy++) for (int ix = 0; ix < 24; ix++) {                 /* raster order: the look-ahead makes records ahead */
        const double alpha = ((ix + .5) / 24.0 - .5) * 2.0 * rmax, beta = ((iy + .5) / 8.0 - .5) * 2.0 * rmax * (8.0 / 24.0);
        geodesic gd; int err = 0;
        if (!geodesic_init_inf(inc, a, alpha, beta, &gd, &err)) continue;
        const double P = geodesic_find_midplane_crossing(&gd, 0);
        if (isnan(P)) continue;
        const double r = geodesic_position_rad(&gd, P);
        if (!(r > r_ms(a) + 0.5)) continue;
        const double f_old = disk_nt_flux(r);                                            /* answered from the record */
        if ((ix + iy) % 5 == 0) {
            /* the model changes behind the shim's back: ten times the accretion rate */
            sim5gpu_disk_nt_setup(10.0, a, (tried % 2) ? 0.1 : 1.0, 0.1, 0);
            const double f_new = disk_nt_flux(r);                                        /* same r, same record -- must NOT be the old value */
            double f_direct = NAN;
            sim5gpu_disk_nt_flux(1, &r, &f_direct);
            tried++;
            if (memcmp(&f_new, &f_direct, sizeof f_new) != 0 || f_new == f_old) {
                bad++;
                printf("stale flux at r = %.17g: record %.17g, model now set %.17g, before %.17g\n", r, f_new, f_direct, f_old);
            }
        }
    }
    printf("%s: %d model changes beside the shim, %d stale answers\n", bad ? "FAILED" : "ok", tried, bad);
    return (bad || tried < 5) ? 1 : 0;
}
