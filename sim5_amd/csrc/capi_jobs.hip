// capi_jobs.hip -- C-ABI of the polarized thin-disk image and of the step-wise torus ray tracer.
#include <vector>
#include "capi_util.hpp"
#include "k_torus.hpp"
#include <math.h>
#include <string.h>

using namespace s5;

extern "C" {

int sim5gpu_disk_image_polarized(const sim5gpu_image_desc* desc, double* d_stokes, double* d_chi,
                                 const sim5gpu_image_aux* d_aux, void* stream)
{
    if (!d_stokes) { snprintf(g_err, sizeof g_err, "disk_image_polarized: stokes pointer is NULL"); return SIM5GPU_E_ARG; }
    ImageParams p;
    int rc = fill_image_params(desc, p);
    if (rc) return rc;
    if (desc->flags & SIM5GPU_IMG_INPLACE) { snprintf(g_err, sizeof g_err, "disk_image_polarized: SIM5GPU_IMG_INPLACE is not supported"); return SIM5GPU_E_ARG; }
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    FluxPinScope pins;                                       // the table block stays until the launches below have been enqueued
    if (!(desc->flags & SIM5GPU_IMG_STRICT) && ((rc = attach_flux_table(p.disk)) != 0 || (rc = attach_K_table(p)) != 0)) return rc;
    p.stokes = d_stokes;
    p.chi = d_chi;
    if (d_aux) { p.cls = d_aux->cls; p.gtype = d_aux->gtype; p.r = d_aux->r; p.g = d_aux->g; p.flux = d_aux->flux; }
    hipError_t e = (hipError_t)((desc->flags & SIM5GPU_IMG_STRICT) ? s5_launch_disk_image_polarized_strict(p, (hipStream_t)stream)
                                                                 : s5_launch_disk_image_polarized_fast(p, (hipStream_t)stream));
    if (e != hipSuccess) { set_error("disk_image_polarized launch", e); return SIM5GPU_E_HIP; }
    return SIM5GPU_OK;
}

// The surface table must have strictly ascending radii: equal neighbours would divide by zero in the interpolation
// and a descending pair breaks the bisection.  The table (<= 32 KB) is read back on the job's stream and checked
// before the launch, so a bad table is an argument error of this call and not NaNs in its output.  The read-back blocks
// the host until the stream has drained; a caller that has checked its table (it usually holds the host copy) passes
// SIM5GPU_SURFACE_TABLE_CHECKED and the job is enqueued without touching the host, overlapping whatever is in flight.
static int check_surface_table(const char* fn, int n_table, const double* d_R, hipStream_t stream)
{
    std::vector<double> h((size_t)n_table);
    hipError_t e = hipMemcpyAsync(h.data(), d_R, sizeof(double) * (size_t)n_table, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { set_error(fn, e); return SIM5GPU_E_HIP; }
    for (int i = 1; i < n_table; i++)
        if (!(h[i] > h[i - 1])) {
            snprintf(g_err, sizeof g_err, "%s: the radii of the surface table must be strictly ascending (R[%d] = %g, R[%d] = %g)",
                     fn, i - 1, h[i - 1], i, h[i]);
            return SIM5GPU_E_ARG;
        }
    if (!(h[0] == h[0])) { snprintf(g_err, sizeof g_err, "%s: NaN in the surface table", fn); return SIM5GPU_E_ARG; }
    return SIM5GPU_OK;
}

int sim5gpu_disk_surface_rays(double a, double incl, int n_table, const double* d_R, const double* d_H,
                              size_t n, const double* d_alpha, const double* d_beta,
                              double* d_P, double* d_r, double* d_m, double* d_k, int* d_status,
                              int strict, void* stream)
{
    if (!d_R || !d_H || !d_alpha || !d_beta || !d_P || !d_r || !d_m || !d_status) {
        snprintf(g_err, sizeof g_err, "disk_surface_rays: NULL pointer argument");
        return SIM5GPU_E_ARG;
    }
    if (n_table < 2 || n_table > 4096) {
        snprintf(g_err, sizeof g_err, "disk_surface_rays: n_table must be in [2, 4096]");
        return SIM5GPU_E_ARG;
    }
    if (n == 0) return SIM5GPU_OK;
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    if (!(strict & SIM5GPU_SURFACE_TABLE_CHECKED)) { int rc = check_surface_table("disk_surface_rays", n_table, d_R, (hipStream_t)stream); if (rc) return rc; }
    strict &= 1;
    SurfaceParams p;
    p.n = n; p.n_table = n_table; p.a = a; p.incl = incl; reference_sincos(incl, p.sin_i, p.cos_i);
    p.tab_vr = nullptr; p.out_g = nullptr; p.out_mue = nullptr; p.out_flux = nullptr;
    memset(&p.disk, 0, sizeof p.disk);
    hipError_t e = (hipError_t)(strict
        ? s5_launch_disk_surface_strict(p, d_R, d_H, d_alpha, d_beta, d_P, d_r, d_m, d_k, d_status, (hipStream_t)stream)
        : s5_launch_disk_surface_fast(p, d_R, d_H, d_alpha, d_beta, d_P, d_r, d_m, d_k, d_status, (hipStream_t)stream));
    if (e != hipSuccess) { set_error("disk_surface_rays launch", e); return SIM5GPU_E_HIP; }
    return SIM5GPU_OK;
}

int sim5gpu_disk_surface_frame(double a, double incl, double bh_mass, double mdot, double disk_spin,
                               int n_table, const double* d_R, const double* d_H, const double* d_vr,
                               size_t n, const double* d_alpha, const double* d_beta,
                               double* d_P, double* d_r, double* d_m, double* d_k, int* d_status,
                               double* d_g, double* d_mue, double* d_flux, int strict, void* stream)
{
    if (!d_R || !d_H || !d_alpha || !d_beta || !d_P || !d_r || !d_m || !d_status || !d_g || !d_mue || !d_flux) {
        snprintf(g_err, sizeof g_err, "disk_surface_frame: NULL pointer argument");
        return SIM5GPU_E_ARG;
    }
    if (n_table < 2 || n_table > 4096) {
        snprintf(g_err, sizeof g_err, "disk_surface_frame: n_table must be in [2, 4096]");
        return SIM5GPU_E_ARG;
    }
    if (!(bh_mass > 0.0) || !(mdot > 0.0)) {
        snprintf(g_err, sizeof g_err, "disk_surface_frame: need bh_mass > 0 and mdot > 0");
        return SIM5GPU_E_ARG;
    }
    if (n == 0) return SIM5GPU_OK;
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    if (!(strict & SIM5GPU_SURFACE_TABLE_CHECKED)) { int rc = check_surface_table("disk_surface_frame", n_table, d_R, (hipStream_t)stream); if (rc) return rc; }
    strict &= 1;
    SurfaceParams p;
    p.n = n; p.n_table = n_table; p.a = a; p.incl = incl; reference_sincos(incl, p.sin_i, p.cos_i);
    p.disk = make_disk_consts(bh_mass, disk_spin >= 0.0 ? disk_spin : a, mdot);
    FluxPinScope pins;                                       // the table block stays until the launches below have been enqueued
    if (!strict) { int rc = attach_flux_table(p.disk); if (rc) return rc; }
    p.tab_vr = d_vr; p.out_g = d_g; p.out_mue = d_mue; p.out_flux = d_flux;
    hipError_t e = (hipError_t)(strict
        ? s5_launch_disk_surface_strict(p, d_R, d_H, d_alpha, d_beta, d_P, d_r, d_m, d_k, d_status, (hipStream_t)stream)
        : s5_launch_disk_surface_fast(p, d_R, d_H, d_alpha, d_beta, d_P, d_r, d_m, d_k, d_status, (hipStream_t)stream));
    if (e != hipSuccess) { set_error("disk_surface_frame launch", e); return SIM5GPU_E_HIP; }
    return SIM5GPU_OK;
}

static void spectrum_grid(const sim5gpu_image_desc* desc, size_t& nblocks)
{
    // an upper bound for both variants' tilings (strict: 32 x 8 pixels per workgroup; fast: 16 x 16, and 16 x (16 + 16) for a
    // symmetric row set): the partial spectra and the tree's scratch are sized from it
    const int rows = sim5gpu_image_rows(desc);
    nblocks = (size_t)((desc->nx + 15) / 16) * (size_t)((rows + 7) / 8);
}

size_t sim5gpu_disk_spectrum_workspace(const sim5gpu_image_desc* desc, int n_energies)
{
    if (!desc || n_energies <= 0 || desc->nx <= 0) return 0;
    size_t nblocks;
    spectrum_grid(desc, nblocks);
    return (nblocks + (nblocks + 63) / 64) * (size_t)n_energies * sizeof(double);      // partial rows + one tree level of scratch
}

int sim5gpu_disk_spectrum(const sim5gpu_image_desc* desc, int n_energies, const double* d_energies,
                          double hardening, int limb_darkening, double* d_spectrum, void* d_workspace, void* stream)
{
    if (!d_energies || !d_spectrum || !d_workspace || n_energies <= 0 || !(hardening > 0.0)) {
        snprintf(g_err, sizeof g_err, "disk_spectrum: need energies, spectrum, workspace, n_energies > 0, hardening > 0");
        return SIM5GPU_E_ARG;
    }
    if (desc && (desc->stripe_rows != 0 || (desc->flags & (SIM5GPU_IMG_MIRROR | SIM5GPU_IMG_INPLACE)))) { snprintf(g_err, sizeof g_err, "disk_spectrum: striping / mirrored / in-place rows are not supported"); return SIM5GPU_E_ARG; }
    ImageParams p;
    int rc = fill_image_params(desc, p);
    if (rc) return rc;
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    FluxPinScope pins;                                       // the table block stays until the launches below have been enqueued
    if (!(desc->flags & SIM5GPU_IMG_STRICT) && ((rc = attach_flux_table(p.disk)) != 0 || (rc = attach_K_table(p)) != 0)) return rc;
    p.max_order = 1;                  // the Python ray tracer uses the first crossing only
    p.rms = 0.0;                      // and lets the disk model decide (zero flux inside its inner edge)
    SpectrumParams sp;
    sp.n_energies = n_energies;
    int eb = 1;
    while (eb < n_energies && eb < 256) eb *= 2;
    sp.bins_per_pass = eb;
    sp.limb_darkening = limb_darkening;
    sp.hardening = hardening;
    hipError_t e = (hipError_t)((desc->flags & SIM5GPU_IMG_STRICT)
        ? s5_launch_disk_spectrum_strict(p, sp, d_energies, (double*)d_workspace, d_spectrum, (hipStream_t)stream)
        : s5_launch_disk_spectrum_fast(p, sp, d_energies, (double*)d_workspace, d_spectrum, (hipStream_t)stream));
    if (e != hipSuccess) { set_error("disk_spectrum launch", e); return SIM5GPU_E_HIP; }
    return SIM5GPU_OK;
}

int sim5gpu_torus_image(const sim5gpu_torus_desc* desc, sim5gpu_stokes* d_stokes,
                        const sim5gpu_torus_aux* d_aux, void* stream)
{
    if (!desc || !d_stokes) { snprintf(g_err, sizeof g_err, "torus_image: NULL pointer argument"); return SIM5GPU_E_ARG; }
    ImageParams ip;
    int rc = fill_image_params(&desc->img, ip, false);       // image geometry only: the disk fields are unused
    if (rc) return rc;
    if (!(desc->r0 > 0.0) || !(desc->precision > 0.0) || desc->max_steps < 1) {
        snprintf(g_err, sizeof g_err, "torus_image: need r0 > 0, precision > 0, max_steps >= 1");
        return SIM5GPU_E_ARG;
    }
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    TorusParams p;
    memset(&p, 0, sizeof p);
    if (desc->img.stripe_rows != 0 || (desc->img.flags & SIM5GPU_IMG_MIRROR)) { snprintf(g_err, sizeof g_err, "torus_image: striping / mirrored rows are not supported"); return SIM5GPU_E_ARG; }
    p.nx = ip.nx; p.ny = ip.ny; p.y0 = ip.y0; p.y1 = ip.y1;
    p.nrays = (size_t)(ip.y1 - ip.y0) * (size_t)ip.nx;
    p.a = ip.a; p.incl = ip.incl; p.sin_i = ip.sin_i; p.cos_i = ip.cos_i; p.rmax = ip.rmax;
    p.r0 = desc->r0;
    p.precision = desc->precision;
    p.dl_max = desc->dl_max > 0.0 ? desc->dl_max : 1e9;
    p.options = desc->options;
    p.max_steps = desc->max_steps;
    p.shape = desc->shape;
    p.max_error = desc->max_error > 0.0 ? desc->max_error : 1e-2;
    p.r_stop_in = desc->r_stop_in > 0.0 ? desc->r_stop_in : 1.05;
    p.r_stop_out = desc->r_stop_out > 0.0 ? desc->r_stop_out : 1.01;
    p.torus_r = desc->torus_r; p.torus_w = desc->torus_w; p.torus_l = desc->torus_l;
    p.inv_2w2 = 1.0 / (2. * desc->torus_w * desc->torus_w); p.cut_d2 = 36. * (2. * desc->torus_w * desc->torus_w);
    p.emis0 = desc->emis0; p.absorb0 = desc->absorb0;
    TorusAux aux = { nullptr, nullptr, nullptr, nullptr, nullptr };
    if (d_aux) {
        aux.steps = d_aux->steps; aux.max_step_error = d_aux->max_step_error;
        aux.carter_error = d_aux->carter_error; aux.x_end = d_aux->x_end; aux.k_end = d_aux->k_end;
    }
    hipError_t e = (hipError_t)((desc->img.flags & SIM5GPU_IMG_STRICT) ? s5::launch_torus_strict(p, d_stokes, aux, (hipStream_t)stream)
                                                                     : s5f::launch_torus_fast(p, d_stokes, aux, (hipStream_t)stream));
    if (e != hipSuccess) { set_error("torus_image launch", e); return SIM5GPU_E_HIP; }
    return SIM5GPU_OK;
}

/* Workspaces of the surface-search and torus jobs grow to the largest job seen on a device and are kept (0.5 GB per
 * million rays for the surface search, 120 B per ray for the torus job): this gives them back.  Waits for the devices
 * that own them.  *bytes (may be NULL) receives what was freed. */
int sim5gpu_release_workspaces(size_t* bytes)
{
    if (!have_device()) { if (bytes) *bytes = 0; return SIM5GPU_OK; }
    const size_t freed = s5_release_surface_workspace_fast() + s5_release_surface_workspace_strict() +
                         s5f::release_torus_workspace_fast() + s5::release_torus_workspace_strict() + release_flux_tables();
    if (bytes) *bytes = freed;
    return SIM5GPU_OK;
}

} // extern "C"
