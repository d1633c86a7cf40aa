// k_disk_image.hip -- thin-disk image kernels (elliptic-integral path) for gfx950.
//
// One lane traces one image-plane ray through
//   geodesic_init_inf -> geodesic_find_midplane_crossing -> geodesic_position_rad ->
//   gfactorK / disk_nt_flux
// i.e. the body of the caller loop of ref examples/04-disk-image-eqplane/disk-image.c:53-105,
// for up to max_order crossings, and writes (float)(F g^4) and (float)g.
//
// Launch geometry: a 256-thread workgroup covers a 16 x 16 pixel tile, each wave64 a 16 x 4
// patch of it.  Rays of a wave are image-plane neighbours, so they share the geodesic class and
// the Carlson trip counts almost always (measured lane utilisation 98 %); each wave row stores 16
// consecutive f32 = half a 128-B line per plane, the workgroup whole lines.  No input is read in grid mode
// (alpha, beta follow from the pixel index, ref disk-image.c:57-58); in list mode alpha[]/beta[] are read
// coalesced, 8 B per lane.
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

S5_DEV RayResult trace_disk_ray(const ImageParams& p, double alpha, double beta)
{
    ThinRay t;
    trace_thin_disk<false>(p, alpha, beta, t);
    RayResult out;
    out.cls = t.cls; out.gtype = t.gtype;
    out.r = t.r; out.g = t.g; out.flux = t.flux;
    const double g2 = t.g * t.g;
    const bool hit = (t.cls == PX_HIT0) || (t.cls == PX_HIT1);
    out.image_f = hit ? (float)(t.flux * (g2 * g2)) : 0.0f;
    out.image_g = hit ? (float)t.g : 0.0f;
    return out;
}

S5_DEV void store_ray(const ImageParams& p, size_t o, const RayResult& res)
{
    p.img_f[o] = res.image_f;
    p.img_g[o] = res.image_g;
    if (p.cls) p.cls[o] = (uint8_t)res.cls;
    if (p.gtype) p.gtype[o] = (int8_t)res.gtype;
    if (p.r) p.r[o] = res.r;
    if (p.g) p.g[o] = res.g;
    if (p.flux) p.flux[o] = res.flux;
}

#ifndef S5_TILE_W
#define S5_TILE_W 16                     // pixels per wave row: a wave covers 16 x 4 pixels (64-B store segments).
                                         // Compact patches keep more waves class-uniform (s5_thindisk.hpp); measured
                                         // on MI355X, 4096^2, same box: 8x8 0.903 ms, 16x4 0.890, 32x2 0.901-0.906,
                                         // 64x1 0.909
#endif
#ifndef S5_LB_WAVES
#define S5_LB_WAVES 2                    // a floor only: the kernel needs 111 VGPRs and 40 KB of LDS per workgroup
                                         // (ladder rungs), so 4 waves/SIMD are resident.  Occupancy is not a lever:
                                         // 4 -> 6 waves/SIMD (shorter ladder) 0.922 -> 0.915 ms; forcing 8 spills.
#endif
constexpr int TILE_W = S5_TILE_W;        // workgroup tile: TILE_W x (256 / TILE_W) pixels
constexpr int TILE_H = 256 / TILE_W;

__global__ __launch_bounds__(256, S5_LB_WAVES)
void disk_image_grid_kernel(ImageParams p)
{
    const int lane_x = threadIdx.x % TILE_W;
    const int lane_y = threadIdx.x / TILE_W;
    const int ix = blockIdx.x * TILE_W + lane_x;
    const int lr = blockIdx.y * TILE_H + lane_y;                 // packed (local) row
    if (ix >= p.nx || lr >= p.nrows) return;
    const int iy = p.stripe_rows > 0 ? p.y0 + (lr / p.stripe_rows) * p.stripe_step + lr % p.stripe_rows
                                     : p.y0 + lr;

    // ref disk-image.c:57-58 (operation order kept; the fast variant multiplies by the reciprocals of the
    // image size instead of dividing: alpha, beta move by at most 1 ulp)
#if S5_FAST
    const double alpha = (((double)(ix) + .5) * p.inv_nx - 0.5) * 2.0 * p.rmax;
    const double beta = (((double)(iy) + .5) * p.inv_ny - 0.5) * 2.0 * p.rmax * p.ny_over_nx;
#else
    const double alpha = (((double)(ix) + .5) / (double)(p.nx) - 0.5) * 2.0 * p.rmax;
    const double beta = (((double)(iy) + .5) / (double)(p.ny) - 0.5) * 2.0 * p.rmax *
                        ((double)p.ny / (double)p.nx);
#endif

    const RayResult res = trace_disk_ray(p, alpha, beta);
    store_ray(p, (size_t)lr * (size_t)p.nx + (size_t)ix, res);
}

__global__ __launch_bounds__(256, 2)
void disk_image_list_kernel(ImageParams p)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const RayResult res = trace_disk_ray(p, p.alpha[i], p.beta[i]);
    store_ray(p, i, res);
}

} // namespace S5NS

#if S5_FAST
int s5_launch_disk_image_fast(const s5abi::ImageParams& p, hipStream_t stream)
#else
int s5_launch_disk_image_strict(const s5abi::ImageParams& p, hipStream_t stream)
#endif
{
    using namespace S5NS;
    if (p.alpha) {
        const unsigned blocks = (unsigned)((p.n + 255) / 256);
        hipLaunchKernelGGL(disk_image_list_kernel, dim3(blocks), dim3(256), 0, stream, p);
    } else {
        const dim3 grid((p.nx + TILE_W - 1) / TILE_W, (p.nrows + TILE_H - 1) / TILE_H);
        hipLaunchKernelGGL(disk_image_grid_kernel, grid, dim3(256), 0, stream, p);
    }
    return (int)hipGetLastError();
}
