// instruction-throughput microbenchmark (gfx950): cycles per wave64 instruction for the ops the ray kernel issues
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 4096
#define CHAINS 8
template <int OP> __device__ __forceinline__ double step(double x, double c)
{
    if (OP == 0) return __builtin_fma(x, c, c);
    if (OP == 1) return x * c;
    if (OP == 2) return x + c;
    if (OP == 3) return __builtin_amdgcn_rcp(x);
    if (OP == 4) return __builtin_amdgcn_rsq(x);
    if (OP == 5) return __builtin_amdgcn_sqrt(x);
    if (OP == 6) return (double)__builtin_amdgcn_rsqf((float)x);        // cvt + rsq_f32 + cvt
    if (OP == 7) return (double)(float)x;                                // cvt + cvt
    if (OP == 8) return __builtin_amdgcn_ldexp(x, 1);
    if (OP == 9) return (x > c) ? x : c + 1.0;                           // cmp + 2 cndmask + add
    if (OP == 10) return fmax(x, c);
    if (OP == 11) return __builtin_amdgcn_frexp_mant(x);
    if (OP == 12) return __builtin_amdgcn_fract(x);
    if (OP == 13) return __builtin_amdgcn_trig_preop(x, 1);
    return x;
}
template <int OP> __global__ __launch_bounds__(256) void k(double* out, double c, long long* cyc)
{
    double v[CHAINS];
    for (int j = 0; j < CHAINS; ++j) v[j] = 1.0 + 0.001 * (threadIdx.x + j);
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
        for (int j = 0; j < CHAINS; ++j) { v[j] = step<OP>(v[j], c); asm volatile("" : "+v"(v[j])); }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int j = 0; j < CHAINS; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> __global__ __launch_bounds__(256) void kf(float* out, float c)
{
    float v[CHAINS];
    for (int j = 0; j < CHAINS; ++j) v[j] = 1.0f + 0.001f * (threadIdx.x + j);
#pragma unroll 1
    for (int i = 0; i < N_ITER; ++i) {
#pragma unroll
        for (int j = 0; j < CHAINS; ++j) {
            if (OP == 0) v[j] = __builtin_fmaf(v[j], c, c);
            if (OP == 1) v[j] = __builtin_amdgcn_rsqf(v[j]);
            if (OP == 2) v[j] = __builtin_amdgcn_rcpf(v[j]);
            asm volatile("" : "+v"(v[j]));
        }
    }
    float s = 0; for (int j = 0; j < CHAINS; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;       // 256 threads = 4 waves = 1 per SIMD
    double* d; long long* cyc; hipMalloc(&d, sizeof(double) * blocks * 256); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 1.0000001, cyc);
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, 1.0000001, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: waves_per_simd waves, each N_ITER*CHAINS ops
    const double ops = (double)waves_per_simd * N_ITER * CHAINS;
    const double cyc_per = ms * 1e-3 * 2.4e9 / ops;
    printf("%-14s waves/SIMD %d : %.3f ms  -> %.2f cycles / wave-op (at 2.4 GHz)\n", name, waves_per_simd, ms, cyc_per);
    hipFree(d); hipFree(cyc);
}
template <int OP> void runf(const char* name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;
    float* d; hipMalloc(&d, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kf<OP><<<blocks, 256>>>(d, 1.0000001f);
    hipEventRecord(e0); kf<OP><<<blocks, 256>>>(d, 1.0000001f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)waves_per_simd * N_ITER * CHAINS;
    printf("%-14s waves/SIMD %d : %.3f ms  -> %.2f cycles / wave-op\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / ops);
    hipFree(d);
}
int main()
{
    for (int w : {3, 4, 8}) {
        run<0>("fma_f64", w); run<1>("mul_f64", w); run<2>("add_f64", w); run<3>("rcp_f64", w); run<4>("rsq_f64", w);
        run<5>("sqrt_f64", w); run<6>("cvt+rsqf+cvt", w); run<7>("cvt+cvt", w); run<8>("ldexp_f64", w);
        run<9>("cmp+2cnd+add", w); run<10>("max_f64", w); run<11>("frexp_mant", w); run<12>("fract_f64", w);
        run<13>("trig_preop", w);
        runf<0>("fma_f32", w); runf<1>("rsq_f32", w); runf<2>("rcp_f32", w);
    }
    return 0;
}
