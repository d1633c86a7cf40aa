// k_assemble.hip -- put the row shares of ONE image back into row order on the device (gfx950).
//
// A multi-GPU split deals an image's rows to the ranks in mirrored stripe pairs; a rank's output holds its rows packed
// (include/sim5gpu.h: SIM5GPU_IMG_MIRROR, stripe_rows / stripe_step).  After the gather the root holds one packed block
// per rank; this kernel copies every row of every block to its image row -- pure data movement, HBM-bound: 16-byte
// loads and stores, a workgroup per (row, plane), all shares in one launch.  The row of a packed row follows from the
// share's job description (kernels.hpp: RowMap / image_row), the very rule the tracing kernels use, so no table travels.
#include "kernels.hpp"

namespace s5asm {

using namespace s5abi;

constexpr int MAX_SHARES = 16;
struct PlaceArgs {
    RowMap map[MAX_SHARES];
    int row_start[MAX_SHARES + 1];      // prefix sums of the shares' row counts: blockIdx.y -> (share, local row)
    int n_shares, nx;
    size_t share_rows;                  // rows per plane of a share block (>= the share's own row count)
    const float* shares;
    float* image_f;
    float* image_g;
};

__global__ __launch_bounds__(256)
void place_shares_kernel(PlaceArgs a)
{
    const int row = (int)blockIdx.y;                       // packed row over all shares
    int s = 0;
    while (s + 1 < a.n_shares && row >= a.row_start[s + 1]) ++s;      // wave-uniform, <= 16 steps
    const int lr = row - a.row_start[s];
    const int iy = image_row(a.map[s], lr);
    const int plane = (int)blockIdx.z;
    const float* __restrict__ src = a.shares + ((size_t)s * 2 + (size_t)plane) * a.share_rows * (size_t)a.nx + (size_t)lr * (size_t)a.nx;
    float* __restrict__ dst = (plane ? a.image_g : a.image_f) + (size_t)iy * (size_t)a.nx;
    const int nx = a.nx;
    if (((nx & 3) == 0) && ((((size_t)src | (size_t)dst) & 15) == 0)) {
        const float4* __restrict__ s4 = (const float4*)src;
        float4* __restrict__ d4 = (float4*)dst;
        for (int i = (int)(blockIdx.x * 256 + threadIdx.x); i < nx / 4; i += (int)(gridDim.x * 256)) d4[i] = s4[i];
    } else {
        for (int i = (int)(blockIdx.x * 256 + threadIdx.x); i < nx; i += (int)(gridDim.x * 256)) dst[i] = src[i];
    }
}

// number of 32-bit words in which two device buffers differ (bit comparison: NaN patterns count like any other word): the
// self-check of an assembled image against a single-launch one without bringing 134 MB to the host.  HBM-bound, 16-byte loads.
__global__ __launch_bounds__(256)
void words_differ_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, size_t n, unsigned long long* count)
{
    unsigned mine = 0;
    const size_t n4 = ((((size_t)a | (size_t)b) & 15) == 0) ? n / 4 : 0;
    const uint4* __restrict__ a4 = (const uint4*)a;
    const uint4* __restrict__ b4 = (const uint4*)b;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const uint4 x = a4[i], y = b4[i];
        mine += (x.x != y.x) + (x.y != y.y) + (x.z != y.z) + (x.w != y.w);
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) mine += (a[i] != b[i]);
    // one atomic per wave that found something (the usual answer is zero: no atomics at all)
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, (unsigned long long)mine);
}

} // namespace s5asm

int s5_launch_words_differ(const void* a, const void* b, size_t n_words, unsigned long long* d_count, hipStream_t stream)
{
    using namespace s5asm;
    if (n_words == 0) return 0;
    size_t blocks = (n_words / 4 + 1023) / 1024;
    if (blocks < 1) blocks = 1;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(words_differ_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint32_t*)a, (const uint32_t*)b, n_words, d_count);
    return (int)hipGetLastError();
}

int s5_launch_place_shares(int n_shares, const s5abi::RowMap* maps, const float* shares, size_t share_rows, int nx,
                           float* image_f, float* image_g, hipStream_t stream)
{
    using namespace s5asm;
    if (n_shares <= 0) return 0;
    if (n_shares > MAX_SHARES) return (int)hipErrorInvalidValue;
    PlaceArgs a;
    a.n_shares = n_shares; a.nx = nx; a.share_rows = share_rows; a.shares = shares; a.image_f = image_f; a.image_g = image_g;
    int total = 0;
    for (int i = 0; i < MAX_SHARES; ++i) {
        a.row_start[i] = total;
        if (i < n_shares) { a.map[i] = maps[i]; total += maps[i].nrows; }
        else a.map[i] = RowMap{ 0, 0, 0, 0, 0, 0, 0 };
    }
    a.row_start[MAX_SHARES] = total;
    if (total == 0) return 0;
    const unsigned bx = (unsigned)((nx / 4 + 1023) / 1024 > 0 ? (nx / 4 + 1023) / 1024 : 1);      // <= 4 float4 per thread
    hipLaunchKernelGGL(place_shares_kernel, dim3(bx, (unsigned)total, 2), dim3(256), 0, stream, a);
    return (int)hipGetLastError();
}
