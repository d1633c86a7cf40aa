// k_disk_image.hip -- thin-disk image kernels (elliptic-integral path) for gfx950.
//
// One lane traces one image-plane ray through
//   geodesic_init_inf -> geodesic_find_midplane_crossing -> geodesic_position_rad ->
//   gfactorK / disk_nt_flux
// i.e. the body of the caller loop of ref examples/04-disk-image-eqplane/disk-image.c:53-105,
// for up to max_order crossings, and writes (float)(F g^4) and (float)g.  Where the row set is symmetric about the
// middle of the image (whole images, centred bands, SIM5GPU_IMG_MIRROR jobs) the fast variant gives a lane the ray AND its
// mirror image in beta: they share the geodesic (disk_image_mirror_kernel below, s5_thindisk.hpp).
//
// Launch geometry: a 256-thread workgroup covers a 16 x 16 pixel tile, each wave64 a 16 x 4
// patch of it.  Rays of a wave are image-plane neighbours, so they share the geodesic class and
// the Carlson trip counts almost always (measured lane utilisation 97 % in both kernels); each wave row stores 16
// consecutive f32 = half a 128-B line per plane, the workgroup whole lines.  No input is read in grid mode
// (alpha, beta follow from the pixel index, ref disk-image.c:57-58); in list mode alpha[]/beta[] are read
// coalesced, 8 B per lane.
#include <string.h>
#include "s5_thindisk.hpp"
#include "kernels.hpp"

namespace S5NS {

using namespace s5abi;

struct RayResult {
    int    cls;      // SIM5GPU_PX_*
    int    gtype;    // geodesic type or -1
    double r, g, flux;
    float  image_f, image_g;
};

S5_DEV RayResult ray_result(const ThinRay& t)
{
    RayResult out;
    out.cls = t.cls; out.gtype = t.gtype;
    out.r = t.r; out.g = t.g; out.flux = t.flux;
    const double g2 = t.g * t.g;
    const bool hit = (t.cls == PX_HIT0) || (t.cls == PX_HIT1);
    out.image_f = hit ? (float)(t.flux * (g2 * g2)) : 0.0f;
    out.image_g = hit ? (float)t.g : 0.0f;
    return out;
}

S5_DEV RayResult trace_disk_ray(const ImageParams& p, double alpha, double beta, const int iy = -1)
{
    ThinRay t;
    trace_thin_disk<false>(p, alpha, beta, t, iy);
    return ray_result(t);
}

// AUX = false: the instantiation for jobs without full-precision planes (every production job: bench, sharded images) --
// five pointers fewer held in SGPRs through the kernel and no tests around the stores
template <bool AUX>
S5_DEV void store_ray(const ImageParams& p, size_t o, const RayResult& res)
{
    p.img_f[o] = res.image_f;
    p.img_g[o] = res.image_g;
    if (AUX) {
        if (p.cls) p.cls[o] = (uint8_t)res.cls;
        if (p.gtype) p.gtype[o] = (int8_t)res.gtype;
        if (p.r) p.r[o] = res.r;
        if (p.g) p.g[o] = res.g;
        if (p.flux) p.flux[o] = res.flux;
    }
}

#define S5_TILE_W 16                     // pixels per wave row: a wave covers 16 x 4 pixels (64-B store segments).
                                         // Compact patches keep more waves class-uniform (s5_thindisk.hpp); measured
                                         // on MI355X, 4096^2, same box: 8x8 0.903 ms, 16x4 0.890, 32x2 0.901-0.906,
                                         // 64x1 0.909
#define S5_LB_WAVES 2                    // a floor only: the kernel needs ~110 VGPRs and 34 KB of LDS per workgroup
                                         // (ladder rungs), so 4 waves/SIMD are resident.  Occupancy is not a lever:
                                         // 4 -> 6 waves/SIMD (shorter ladder) 0.922 -> 0.915 ms; forcing 8 spills.
constexpr int TILE_W = S5_TILE_W;        // workgroup tile: TILE_W x (256 / TILE_W) pixels
constexpr int TILE_H = 256 / TILE_W;

template <bool AUX>
__global__ __launch_bounds__(256, S5_LB_WAVES)
void disk_image_grid_kernel(ImageParams p)
{
    const int lane_x = threadIdx.x % TILE_W;
    const int lane_y = threadIdx.x / TILE_W;
    const int ix = blockIdx.x * TILE_W + lane_x;
    const int lr = blockIdx.y * TILE_H + lane_y;                 // packed (local) row
    if (ix >= p.nx || lr >= p.nrows) return;
    const int iy = image_row(p, lr);
    const RayResult res = trace_disk_ray(p, pixel_alpha(p, ix), pixel_beta(p, iy), iy);
    store_ray<AUX>(p, (size_t)(p.inplace ? iy : lr) * (size_t)p.nx + (size_t)ix, res);     // packed rows, or in place in the whole image
}

#if S5_FAST
// A row set that is symmetric about the middle of the image -- a plain range with y0 + y1 == ny (the whole image, a centred
// band) or a SIM5GPU_IMG_MIRROR job (rows of the upper half, striped or not, plus their mirror images) -- has its packed
// rows in increasing image-row order, so local rows lr and nrows - 1 - lr are mirror images of each other: a lane
// takes the pixel (ix, iy) of the upper half AND its mirror image (ix, ny - 1 - iy).  The two rays differ in the sign of
// beta only and share the geodesic (trace_thin_disk_impl<.., PAIR>) -- about two thirds of a ray's arithmetic.  The image
// is the plain kernel's bit for bit (pixel_beta above); an odd middle row is its own mirror and is written once.
#define S5_LB_WAVES_MIRROR 4             // four waves per SIMD (the kernel needs 112 VGPRs, no scratch).  Measured, 4096^2, same
                                         // call: 0.527 ms at three -> 0.478 ms
template <bool AUX>
__global__ __launch_bounds__(256, S5_LB_WAVES_MIRROR)
void disk_image_mirror_kernel(ImageParams p)
{
    const int lane_x = threadIdx.x % TILE_W;
    const int lane_y = threadIdx.x / TILE_W;
    const int ix = blockIdx.x * TILE_W + lane_x;
    // Row tiles are handed out from the MIDDLE of the image outwards (workgroups are dispatched in blockIdx order): the rays
    // around the shadow -- second crossings, RC geodesics, deeper Landen ladders -- cost several times the rays of the
    // outer rows, which mostly miss; started first they are done when the cheap rows fill the end of the launch, instead of
    // forming its tail.  Matters for small images (1024^2 is two rounds of resident waves); the pixel a lane traces is the same.
    const int lr = (int)(gridDim.y - 1u - blockIdx.y) * TILE_H + lane_y;      // local row in the upper half
    const int half = (p.nrows + 1) / 2;
    if (ix >= p.nx || lr >= half) return;
    const int lr2 = p.nrows - 1 - lr;                            // its mirror row (== lr for an odd middle row)
    ThinRay t, t2;
    const int iy = image_row_top(p, lr);
    trace_thin_disk_impl<false, true, false, true>(p, pixel_alpha(p, ix), pixel_beta(p, iy), t, t2, iy);   // (p is this kernel's first parameter)
    store_ray<AUX>(p, (size_t)(p.inplace ? iy : lr) * (size_t)p.nx + (size_t)ix, ray_result(t));
    if (lr2 != lr) store_ray<AUX>(p, (size_t)(p.inplace ? p.ny - 1 - iy : lr2) * (size_t)p.nx + (size_t)ix, ray_result(t2));
}

// JOB LIST.  The same pairing kernel for up to JOBS_MAX jobs in ONE launch: the grid is the concatenation of the jobs' tiles
// (blockIdx.x -> job by the prefix sums in the list, then the tile of that job, middle rows first), so the hardware
// dispatcher streams the jobs through the chip back to back -- no launch gap, no ramp and no ragged last round between
// them; a small image (1024^2 is two rounds of resident waves) or a rank's share of a split image then costs what its rays
// cost.  The jobs' parameters are read from the kernel-argument segment through the constant address space (kernels.hpp:
// JobList): scalar loads at the point of use, the late ones (flux table, output pointers) behind param_reload().  A lane's
// arithmetic is that of disk_image_mirror_kernel<false>, value for value: images are the same bits.
// SINGLE: the list holds one job (sim5gpu_disk_image of a symmetric row set): no search, the job at a constant offset
template <bool SINGLE>
__global__ __launch_bounds__(256, S5_LB_WAVES_MIRROR)
void disk_image_jobs_kernel(JobList list_arg)
{
    const S5_AS4 JobList* L = (const S5_AS4 JobList*)__builtin_amdgcn_kernarg_segment_ptr();
    const int t = (int)blockIdx.x;
    int j = 0, t0 = 0;
    if (!SINGLE) {
        // the job of this tile: all prefix sums in one scalar load, compared in registers (a search with a load per step is a
        // chain of dependent memory round trips at the head of every workgroup).  Entries past the last job hold the total.
#pragma unroll
        for (int k = 0; k < JOBS_MAX - 1; ++k) {
            const int te = L->tile_end[k];
            if (t >= te) { j = k + 1; t0 = te; }
        }
    }
    const S5_AS4 FastJob& p = L->job[SINGLE ? 0 : j];
    const int nx = p.nx, nrows = p.nrows;
    const int half = (nrows + 1) / 2;
    const int tiles_x = (nx + TILE_W - 1) / TILE_W;
    const int tiles_y = (half + TILE_H - 1) / TILE_H;
    const int tj = t - t0;                                               // tile within the job, row-major, dispatched in order
    const int by = tj / tiles_x, bx = tj - by * tiles_x;
    const int lane_x = threadIdx.x % TILE_W;
    const int lane_y = threadIdx.x / TILE_W;
    const int ix = bx * TILE_W + lane_x;
    const int lr = (tiles_y - 1 - by) * TILE_H + lane_y;                 // middle rows first (as disk_image_mirror_kernel)
    if (ix >= nx || lr >= half) return;
    const int lr2 = nrows - 1 - lr;
    ThinRay t1, t2;
    // image row of packed row lr (kernels.hpp image_row_top).  A striped job (a rank's share of a split image) divides by
    // the stripe height: per lane that is ~25 vector instructions of integer division; a tile's 16 rows lie in ONE stripe
    // when the stripe height is a multiple of 16 (the dealt stripes are 64 rows), so the tile's first row is divided once,
    // on a wave-uniform value, and the lanes add their row
    int iy;
    const int sr = p.stripe_rows;
    if (sr > 0 && (sr % TILE_H) == 0) {
        const int lr0 = (tiles_y - 1 - by) * TILE_H;
        const int q = __builtin_amdgcn_readfirstlane(lr0 / sr);
        iy = p.y0 + q * p.stripe_step + (lr0 - q * sr) + lane_y;
    } else iy = image_row_top(p, lr);
    trace_thin_disk_impl<false, true, false, false>(p, pixel_alpha(p, ix), pixel_beta(p, iy), t1, t2, iy);
    const RayResult r1 = ray_result(t1), r2 = ray_result(t2);
    const S5_AS4 FastJob& po = param_reload(p);                          // the output side: loaded here, not carried through the trace
    const int inplace = po.inplace, ny = po.ny;
    float* __restrict__ img_f = po.img_f;
    float* __restrict__ img_g = po.img_g;
    const size_t o1 = (size_t)(inplace ? iy : lr) * (size_t)nx + (size_t)ix;
    img_f[o1] = r1.image_f; img_g[o1] = r1.image_g;
    if (lr2 != lr) {
        const size_t o2 = (size_t)(inplace ? ny - 1 - iy : lr2) * (size_t)nx + (size_t)ix;
        img_f[o2] = r2.image_f; img_g[o2] = r2.image_g;
    }
}
#endif

__global__ __launch_bounds__(256, 2)
void disk_image_list_kernel(ImageParams p)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const RayResult res = trace_disk_ray(p, p.alpha[i], p.beta[i]);
    store_ray<true>(p, i, res);
}

} // namespace S5NS

#if S5_FAST
// a job the job-list kernel serves: the fast pairing kernel's row sets, two f32 planes only
bool s5_jobs_eligible(const s5abi::ImageParams& p)
{
    const bool aux = p.cls || p.gtype || p.r || p.g || p.flux;
    return !aux && !p.alpha && p.img_f && p.img_g && p.nrows >= 2 && (p.mirror || (p.stripe_rows == 0 && p.y0 + p.y1 == p.ny));
}

int s5_launch_disk_image_jobs_fast(const s5abi::ImageParams* jobs, int n, hipStream_t stream)
{
    using namespace S5NS;
    if (n <= 0) return 0;
    if (n > JOBS_MAX) return (int)hipErrorInvalidValue;
    JobList L;
    memset(&L, 0, sizeof L);
    L.njobs = n;
    long long total = 0;
    for (int j = 0; j < n; ++j) {
        const ImageParams& p = jobs[j];
        if (!s5_jobs_eligible(p)) return (int)hipErrorInvalidValue;
        FastJob& f = L.job[j];
        f.nx = p.nx; f.ny = p.ny; f.y0 = p.y0; f.y1 = p.y1; f.nrows = p.nrows; f.stripe_rows = p.stripe_rows; f.stripe_step = p.stripe_step;
        f.mirror = p.mirror; f.nrows_top = p.nrows_top; f.max_order = p.max_order; f.inplace = p.inplace; f.direct = p.direct;
        f.a = p.a; f.incl = p.incl; f.sin_i = p.sin_i; f.cos_i = p.cos_i; f.rmax = p.rmax; f.rms = p.rms;
        f.inv_nx = p.inv_nx; f.inv_ny = p.inv_ny; f.ny_over_nx = p.ny_over_nx; f.inv_2a2 = p.inv_2a2; f.ktab = p.ktab; f.sctab = p.sctab;
        f.disk.rms = p.disk.rms; f.disk.x0 = p.disk.x0; f.disk.scale = p.disk.scale; f.disk.ft_wmin = p.disk.ft_wmin;
        f.disk.ft_inv_dw = p.disk.ft_inv_dw; f.disk.ftab = p.disk.ftab; f.disk.cold = p.disk.cold;
        f.img_f = p.img_f; f.img_g = p.img_g;
        total += (long long)((p.nx + TILE_W - 1) / TILE_W) * (long long)(((p.nrows + 1) / 2 + TILE_H - 1) / TILE_H);
        if (total > 0x7fffffffLL) return (int)hipErrorInvalidValue;
        L.tile_end[j] = (int)total;
    }
    for (int j = n; j < JOBS_MAX; ++j) L.tile_end[j] = (int)total;
    if (n == 1) hipLaunchKernelGGL(disk_image_jobs_kernel<true>, dim3((unsigned)total), dim3(256), 0, stream, L);
    else hipLaunchKernelGGL(disk_image_jobs_kernel<false>, dim3((unsigned)total), dim3(256), 0, stream, L);
    return (int)hipGetLastError();
}

int s5_launch_disk_image_fast(const s5abi::ImageParams& p, hipStream_t stream)
#else
int s5_launch_disk_image_strict(const s5abi::ImageParams& p, hipStream_t stream)
#endif
{
    using namespace S5NS;
    const bool aux = p.cls || p.gtype || p.r || p.g || p.flux;
    if (p.alpha) {
        const unsigned blocks = (unsigned)((p.n + 255) / 256);
        hipLaunchKernelGGL(disk_image_list_kernel, dim3(blocks), dim3(256), 0, stream, p);
    } else {
#if S5_FAST
        if ((p.mirror || (p.stripe_rows == 0 && p.y0 + p.y1 == p.ny)) && p.nrows >= 2) {
            // production jobs (two f32 planes): the job-list kernel with a list of one -- same time as the by-value kernel
            // (measured: 0.3123 against 0.3125 ms at 4096^2, 0.0257 / 0.0259 at 1024^2), and its hot path holds no spilled
            // scalar register (parameters are read where they are used); the by-value kernel serves the full-precision planes
            if (!aux) return s5_launch_disk_image_jobs_fast(&p, 1, stream);
            const dim3 grid((p.nx + TILE_W - 1) / TILE_W, ((p.nrows + 1) / 2 + TILE_H - 1) / TILE_H);
            if (aux) hipLaunchKernelGGL(disk_image_mirror_kernel<true>, grid, dim3(256), 0, stream, p);
            else hipLaunchKernelGGL(disk_image_mirror_kernel<false>, grid, dim3(256), 0, stream, p);
            return (int)hipGetLastError();
        }
#endif
        const dim3 grid((p.nx + TILE_W - 1) / TILE_W, (p.nrows + TILE_H - 1) / TILE_H);
        if (aux) hipLaunchKernelGGL(disk_image_grid_kernel<true>, grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(disk_image_grid_kernel<false>, grid, dim3(256), 0, stream, p);
    }
    return (int)hipGetLastError();
}
