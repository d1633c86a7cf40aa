/* A caller written against the SIM5 scalar API (sim5lib.h): the radii loop of the reference's example 01
 * (r_bh, r_ph, r_mb, r_ms per spin) and the disk-model report (disk_nt_setup by accretion rate and by luminosity,
 * disk_nt_mdot / disk_nt_lumi / disk_nt_sigma, disk_nt_dump).  Used by tests/test_gpu_host_shim.py. */
#include <stdio.h>
#include <stdlib.h>
#include "sim5lib.h"

int main(int argc, char *argv[])
{
    double a;
    for (a = 0.0; a < 1.0; a += 0.25) printf("radii %.4f  %.17g  %.17g  %.17g  %.17g\n", a, r_bh(a), r_ph(a), r_mb(a), r_ms(a));
    disk_nt_setup(10.0, 0.9, 1.0, 0.1, 0);
    printf("model0 %.17g %.17g %.17g %.17g\n", disk_nt_mdot(), disk_nt_lumi(), disk_nt_r_min(), disk_nt_sigma(10.0));
    disk_nt_setup(10.0, 0.5, 0.3, 0.1, DISK_NT_OPTION_LUMINOSITY);
    printf("model1 %.17g %.17g %.17g %.17g\n", disk_nt_mdot(), disk_nt_lumi(), disk_nt_r_min(), disk_nt_sigma(10.0));
    printf("zeros %g %g %g\n", disk_nt_vr(10.0), disk_nt_h(10.0), disk_nt_dhdr(10.0));
    disk_nt_dump(NULL);
    return 0;
}
