// init_probe.hip -- geodesic_init_inf of ONE ray on the device, compiled like a chosen translation unit (flags on the command
// line), with the intermediate values of the polar roots printed.  Diagnostic tool:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DS5_FAST=1 -ffp-contract=fast -Isim5_amd/csrc tests/tools/init_probe.hip -o gpurun_out/init_probe
#include <stdio.h>
#include <math.h>
#include "s5_geod.hpp"
using namespace S5NS;
__global__ void k(double incl, double si, double ci, double a, double alpha, double beta, double* out)
{
    Geod g; GeodCache c; int err = -1;
    const bool ok = init_inf(incl, si, ci, a, alpha, beta, g, err, c);
    out[0] = ok; out[1] = err; out[2] = g.l; out[3] = g.q; out[4] = g.m2p; out[5] = g.m2m;
    const double a2 = g.a * g.a, l2 = g.l * g.l;
    const double qla = g.q + l2 - a2;
    const double X = msqrt(sq(qla) + 4. * g.q * a2) + qla;
    out[6] = mdiv(g.q + g.q, X);
    double mm, mp; const bool e = s5x87::polar_roots_x87(qla, 4. * g.q * a2, a2 + a2, g.q + g.q, mm, mp);
    out[7] = e; out[8] = mp; out[9] = mm;
    out[10] = polar_tests_marginal(out[6], msqrt(out[6]), ci);
}
int main(int argc, char** argv)
{
    const double a = atof(argv[1]), inc = atof(argv[2]) / 180.0 * M_PI, alpha = atof(argv[3]), beta = atof(argv[4]);
    double* d; hipMalloc(&d, 16 * 8);
    k<<<1, 1>>>(inc, sin(inc), cos(inc), a, alpha, beta, d);
    double h[16]; hipMemcpy(h, d, 16 * 8, hipMemcpyDeviceToHost);
    printf("ok %g err %g l %.17g q %.17g m2p %.17g m2m %.17g | double-only m2p %.17g | x87 ran %g m2p %.17g m2m %.17g | marginal %g\n",
           h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10]);
    return 0;
}
