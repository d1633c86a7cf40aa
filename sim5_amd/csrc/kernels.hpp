// kernels.hpp -- launch-side declarations shared by the kernel translation units and the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "s5_disk.hpp"

namespace s5 {

// per-pixel outcome codes (= SIM5GPU_PX_* of include/sim5gpu.h)
enum : int { PX_ERROR = 0, PX_NAN0 = 1, PX_HIT0 = 2, PX_NAN1 = 3, PX_HIT1 = 4, PX_MISS = 5 };

// kernel argument block of the thin-disk image kernels (wave-uniform: lives in SGPRs)
struct ImageParams {
    int nx, ny, y0, y1;
    int max_order;
    double a, incl, sin_i, cos_i;      // sin/cos from the host libm
    double rmax, rms;
    double pol_degree;
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

int launch_disk_image(const ImageParams& p, hipStream_t stream);
int launch_disk_image_polarized(const ImageParams& p, hipStream_t stream);

} // namespace s5
