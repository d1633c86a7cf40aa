// kernels.hpp -- kernel argument blocks (PODs, namespace s5abi) and the launch entry points of the
// kernel translation units, shared with the C-ABI.  Each image kernel exists in the two build
// variants of s5_config.hpp; the launchers carry the variant in their name.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace s5abi {

// per-pixel outcome codes (= SIM5GPU_PX_* of include/sim5gpu.h)
enum : int { PX_ERROR = 0, PX_NAN0 = 1, PX_HIT0 = 2, PX_NAN1 = 3, PX_HIT1 = 4, PX_MISS = 5 };

// Novikov-Thorne disk folded to constants on the host (see s5_disk.hpp); wave-uniform (SGPRs)
constexpr int COLD_N = 16;            // doubles at the head of a disk model's device block (DiskConsts::cold)
constexpr int FT_N = 128, FT_DEG = 7;
constexpr int KT_N = 128, KT_DEG = 7;          // K(m) table: [0, KT_MMAX] in KT_N intervals, degree KT_DEG
constexpr double KT_MMAX = 0.9;

struct DiskConsts {
    double a;            // (double)(float)spin                                  ref :27-28,51
    double rms;          // (double)(float)(r_ms_pow(a) + 1e-3): inner edge     ref :58,91-105
    double x0;           // sqrt(rms)                                            ref :124
    double x1, x2, x3;   // roots of x^3 - 3x + 2a                               ref :125-127
    double p1, p2, p3;   // 3 (x_i - a)^2 / (x_i (x_i - x_j)(x_i - x_k))          ref :131-133
    double d1, d2, d3;   // x0 - x_i                                              ref :131-133
    double mdot, mass;   // (double)(float) values                               ref :145
    // reciprocals and the overall scale, folded on the host for the fast variant
    double inv_x0, inv_d1, inv_d2, inv_d3, scale;
    double alpha;        // (double)(float) viscosity parameter (disk_nt_sigma only)  ref :31,59
    double a2f;          // (double)((float)spin * (float)spin): sqr() of the float static in disk_nt_lumi  ref :172-173
    // Radial profile table of the fast variant (DEVICE memory, made once per spin by the host, s5_disk.hpp): F(r) /
    // (scale (x - x0)) as piecewise polynomials of degree FT_DEG on FT_N equal intervals of w = x0 / x, x = sqrt(r),
    // covering x0 <= x <= 16; NULL = evaluate the closed form
    const double* ftab;
    double ft_wmin, ft_inv_dw;
    // the constants of the closed form once more, in DEVICE memory (a, x0, x1, x2, x3, p1, p2, p3, d1, d2, d3, mdot, mass):
    // the image kernels' fast variant reads them from here in the rare lanes that need the closed form,
    // instead of holding them in SGPRs through the whole kernel.  Set by attach_flux_table (capi_core.hip)
    const double* cold;
    int    ready;
};

// kernel argument block of the thin-disk image kernels (wave-uniform: lives in SGPRs)
struct ImageParams {
    int nx, ny, y0, y1;
    int nrows;                         // rows traced by this launch (packed output rows)
    int stripe_rows, stripe_step;      // 0: rows y0..y1-1; else stripes of stripe_rows rows every stripe_step
    int mirror;                        // SIM5GPU_IMG_MIRROR: packed rows nrows_top .. nrows-1 are the mirror images of the first ones
    int nrows_top;                     // rows named by y0, y1 and the striping (== nrows without mirror)
    int max_order;
    int inplace;                       // SIM5GPU_IMG_INPLACE: the outputs are whole-image planes, a traced row is written at its image row
    int direct;                        // SIM5GPU_IMG_DIRECT: every ray through the direct routine (fast variant)
    double a, incl, sin_i, cos_i;      // sin/cos from the host libm
    double rmax, rms;
    double inv_nx, inv_ny, ny_over_nx; // 1/nx, 1/ny, ny/nx (host doubles; used by the fast variant)
    double inv_2a2;                    // 1 / (2 max(a, 1e-4)^2)              (fast variant)
    double pol_degree;
    const double* ktab;                // K(m) table of the fast variant (DEVICE memory, capi_core.hip) or NULL
    const double* sctab;               // sin / cos table of the fast variant (s5_trig.hpp: msincos_tab; DEVICE memory) or NULL
    DiskConsts disk;
    // outputs (tile-local, row-major)
    float*   img_f;
    float*   img_g;
    uint8_t* cls;
    int8_t*  gtype;
    double*  r;
    double*  g;
    double*  flux;
    double*  stokes;                   // polarized image: planes I | Q | U
    double*  chi;
    // optional explicit ray list
    const double* alpha;
    const double* beta;
    size_t n;
};

// ---- job list of the fast pairing kernel (k_disk_image.hip: disk_image_jobs_kernel) ----------------------------------------
// What that kernel reads of a job, 216 bytes, same member names as ImageParams so that the ray routines take either.  Up to
// JOBS_MAX of them travel BY VALUE in the kernel-argument segment (3.5 KB of its 4 KB): no device allocation, no copy, and the
// kernel reads them through the constant address space -- scalar loads where a value is used, instead of ~80 scalar registers
// of argument block held (and spilled) from the first instruction.
struct FastDisk {
    double rms, x0, scale, ft_wmin, ft_inv_dw;
    const double* ftab;
    const double* cold;
};
struct FastJob {
    int nx, ny, y0, y1;
    int nrows, stripe_rows, stripe_step, mirror, nrows_top, max_order, inplace, direct;
    double a, incl, sin_i, cos_i;
    double rmax, rms;
    double inv_nx, inv_ny, ny_over_nx, inv_2a2;
    const double* ktab;
    const double* sctab;
    FastDisk disk;
    float* img_f;
    float* img_g;
};
constexpr int JOBS_MAX = 16;
struct JobList {
    int njobs, pad;
    int tile_end[JOBS_MAX];           // prefix sums of the jobs' workgroup tiles: job j owns blocks [tile_end[j-1], tile_end[j])
    FastJob job[JOBS_MAX];
};
static_assert(sizeof(FastJob) == 216 && sizeof(JobList) <= 4096, "the job list must fit the kernel-argument segment");

// image row of packed (local) output row lr: the rows named by y0, y1 and the striping first, then -- with mirror -- their
// mirror images ny - 1 - y, so that the packed rows are in increasing image-row order
template <class PRM>
__host__ __device__ inline int image_row_top(const PRM& p, int t)
{
    return p.stripe_rows > 0 ? p.y0 + (t / p.stripe_rows) * p.stripe_step + t % p.stripe_rows : p.y0 + t;
}

__host__ __device__ inline int image_row(const ImageParams& p, int lr)
{
    if (p.mirror && lr >= p.nrows_top) return p.ny - 1 - image_row_top(p, p.nrows - 1 - lr);
    return image_row_top(p, lr);
}

// the rows a job description traces, without the rest of the job: what placing a share into the whole image needs
// (k_assemble.hip); same rule as image_row()
struct RowMap { int ny, y0, nrows, nrows_top, stripe_rows, stripe_step, mirror; };

__host__ __device__ inline int image_row(const RowMap& p, int lr)
{
    const int t = (p.mirror && lr >= p.nrows_top) ? p.nrows - 1 - lr : lr;
    const int y = p.stripe_rows > 0 ? p.y0 + (t / p.stripe_rows) * p.stripe_step + t % p.stripe_rows : p.y0 + t;
    return (p.mirror && lr >= p.nrows_top) ? p.ny - 1 - y : y;
}

// spectrum job (k_spectrum.hip)
struct SpectrumParams {
    int n_energies;
    int bins_per_pass;     // power of two <= 256: energy bins handled concurrently by a workgroup
    int limb_darkening;
    double hardening;
};

// surface search job (k_surface.hip)
struct SurfaceParams {
    size_t n;
    int n_table;
    double a, incl, sin_i, cos_i;
    // optional local-frame outputs at the surface point (sim5gpu_disk_surface_frame): NULL = not wanted
    DiskConsts disk;            // Novikov-Thorne profiles for flux and angular momentum
    const double* tab_vr;       // radial velocity on the nodes of the surface table, or NULL (= 0)
    double* out_g;              // E_inf / E_local
    double* out_mue;            // cosine of the emission angle
    double* out_flux;           // local flux
};

} // namespace s5abi

// fast = tuned FP64 sequences (default); strict = reference parameters, IEEE sqrt/div, no contraction
int s5_launch_disk_image_fast(const s5abi::ImageParams& p, hipStream_t stream);
// n <= JOBS_MAX jobs of the fast variant without full-precision planes, each a symmetric row set (s5_jobs_eligible), ONE launch
int s5_launch_disk_image_jobs_fast(const s5abi::ImageParams* jobs, int n, hipStream_t stream);
bool s5_jobs_eligible(const s5abi::ImageParams& p);
int s5_launch_disk_image_strict(const s5abi::ImageParams& p, hipStream_t stream);
// k_assemble.hip: rows of n shares ([2][share_rows][nx] floats each, share i at shares + i * 2 * share_rows * nx) to their image rows
int s5_launch_place_shares(int n_shares, const s5abi::RowMap* maps, const float* shares, size_t share_rows, int nx,
                           float* image_f, float* image_g, hipStream_t stream);
// k_assemble.hip: *d_count += number of 32-bit words in which the two device buffers differ
int s5_launch_geodesic_chain_fast(size_t n, const double* incl, const double* sin_i, const double* cos_i, const double* a, const double* alpha, const double* beta,
                                  void* geod, int* err, int* ok, struct sim5gpu_geodesic_chain* chain,
                                  const void* disk, size_t disk_bytes, int have_disk, int* done, hipStream_t stream);
int s5_launch_words_differ(const void* a, const void* b, size_t n_words, unsigned long long* d_count, hipStream_t stream);
int s5_launch_disk_image_polarized_fast(const s5abi::ImageParams& p, hipStream_t stream);
int s5_launch_disk_image_polarized_strict(const s5abi::ImageParams& p, hipStream_t stream);
int s5_launch_disk_spectrum_fast(const s5abi::ImageParams& p, const s5abi::SpectrumParams& sp,
                                 const double* energies, double* partial, double* spectrum, hipStream_t stream);
int s5_launch_disk_spectrum_strict(const s5abi::ImageParams& p, const s5abi::SpectrumParams& sp,
                                   const double* energies, double* partial, double* spectrum, hipStream_t stream);
int s5_launch_disk_surface_strict(const s5abi::SurfaceParams& p, const double* tabR, const double* tabH,
                                  const double* alpha, const double* beta, double* P, double* r, double* m,
                                  double* k, int* status, hipStream_t stream);
int s5_launch_disk_surface_fast(const s5abi::SurfaceParams& p, const double* tabR, const double* tabH,
                                const double* alpha, const double* beta, double* P, double* r, double* m,
                                double* k, int* status, hipStream_t stream);
// grow-only per-device workspaces of the surface job given back (bytes freed); k_torus.hpp has the march kernel's
size_t s5_release_surface_workspace_fast();
size_t s5_release_surface_workspace_strict();
