// conn_probe.hip -- metric and connection of the step-wise integrator at ONE point (a, r, m) on the device, compiled like a
// chosen translation unit; all entries printed with 17 digits so that two builds can be diffed.  Diagnostic tool.
#include <stdio.h>
#include <math.h>
#include "s5_kerr.hpp"
using namespace S5NS;
__global__ void k(double a, double r, double m, double* out)
{
    Metric g; Conn G;
#if S5_FAST
    kerr_metric_connection(a, r, m, g, G);
#else
    kerr_metric(a, r, m, g); kerr_connection(a, r, m, G);
#endif
    const double* c = (const double*)&G;
    for (int i = 0; i < 20; ++i) out[i] = c[i];
    out[20] = g.g00; out[21] = g.g11; out[22] = g.g22; out[23] = g.g33; out[24] = g.g03;
}
int main(int argc, char** argv)
{
    const double a = atof(argv[1]), r = atof(argv[2]), m = 1.0 - atof(argv[3]);
    double* d; (void)hipMalloc(&d, 32 * 8);
    k<<<1, 1>>>(a, r, m, d);
    double h[32]; (void)hipMemcpy(h, d, 32 * 8, hipMemcpyDeviceToHost);
    const char* nm[25] = {"t01","t02","t13","t23","r00","r03","r11","r12","r22","r33","h00","h03","h11","h12","h22","h33","p01","p02","p13","p23","g00","g11","g22","g33","g03"};
    for (int i = 0; i < 25; ++i) printf("%s %.17g\n", nm[i], h[i]);
    return 0;
}
