// k_torus.hpp -- argument blocks of the step-wise ray tracer kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/sim5gpu.h"

namespace s5abi {

struct TorusParams {
    int nx, ny, y0, y1;
    size_t nrays;
    double a, incl, sin_i, cos_i, rmax;
    double r0, precision, dl_max;
    int options, max_steps, shape;
    double max_error, r_stop_in, r_stop_out;
    double torus_r, torus_w, torus_l, emis0, absorb0;
    double inv_2w2, cut_d2;       // 1 / (2 w^2) and 36 * 2 w^2, folded on the host (fast variant)
};

struct TorusAux {
    int* steps;
    float* max_step_error;
    double* carter_error;
    double* x_end;
    double* k_end;
};


} // namespace s5abi

namespace s5 { int launch_torus_strict(const s5abi::TorusParams& p, sim5gpu_stokes* out, const s5abi::TorusAux& aux, hipStream_t stream);
               size_t release_torus_workspace_strict();
               // the strict variant's start kernel over the rays the fast job marked (k_torus.hip START_AGAIN)
               hipError_t launch_torus_start_again_strict(const s5abi::TorusParams& p, double* cols, size_t cap, int* ok, hipStream_t stream); }
namespace s5f { int launch_torus_fast(const s5abi::TorusParams& p, sim5gpu_stokes* out, const s5abi::TorusAux& aux, hipStream_t stream);
                size_t release_torus_workspace_fast(); }
