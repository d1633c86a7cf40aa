// accuracy of the v_rcp_f64 / v_rsq_f64 seeds and of the Newton sequences built on them (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* o, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double b = x[i];
    double r = __builtin_amdgcn_rcp(b);
    o[i] = r;
    double y = __builtin_amdgcn_rsq(b);
    o[n + i] = y;
    // one-iteration reciprocal
    double e = __builtin_fma(-b, r, 1.0);
    double r1 = __builtin_fma(r, e, r);
    o[2 * n + i] = r1;
    e = __builtin_fma(-b, r1, 1.0);
    o[3 * n + i] = __builtin_fma(r1, e, r1);
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n), o(4 * n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = 1.0 + (double)(s >> 11) / 9007199254740992.0 * 3.0; }
    double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, 4 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(o.data(), dout, 4 * n * 8, hipMemcpyDeviceToHost);
    long double m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)x[i];
        long double q = 1.0L / sqrtl((long double)x[i]);
        m0 = fmaxl(m0, fabsl((o[i] - t) / t));
        m1 = fmaxl(m1, fabsl((o[n + i] - q) / q));
        m2 = fmaxl(m2, fabsl((o[2 * n + i] - t) / t));
        m3 = fmaxl(m3, fabsl((o[3 * n + i] - t) / t));
    }
    printf("rcp seed max rel err %.3Le (2^%.1f)\nrsq seed %.3Le (2^%.1f)\nrcp 1 iter %.3Le\nrcp 2 iter %.3Le\n",
           m0, (double)log2l(m0), m1, (double)log2l(m1), m2, m3);
    return 0;
}
