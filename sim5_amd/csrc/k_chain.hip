// k_chain.hip -- one ray of the SIM5 scalar API's example-04 loop in ONE launch, in the arithmetic of the fast variant.
//
// What sim5gpu_geodesic_init_inf_chain (capi_batch.hip) does with the strict routines -- geodesic_init_inf, then per crossing
// order the equatorial crossing, the radius there, gfactorK and disk_nt_flux (ref examples/04-disk-image-eqplane/
// disk-image.c:62-100) -- with the routines of namespace s5f: Newton-refined v_rsq / v_rcp instead of IEEE square roots and
// divisions, Carlson's R_F in three passes of a 9th-order series instead of ~6.5 of the 5th-order one, an 8-rung Landen ladder
// without divisions (s5_config.hpp).  A caller of the scalar API waits for the LATENCY of one ray's dependent
// FP64 chain in a lane or two, and that chain is three to four times shorter here; the values agree with the strict ones to
// ~1e-12 relative (tests/test_gpu_host_shim.py).  Compiled for the fast variant only.
#include <string.h>
#include "s5_chain.hpp"
#include "kernels.hpp"

#if S5_FAST
namespace S5NS {

// one workgroup; lane pair (2 i, 2 i + 1) = ray i, crossing orders 0 and 1; announces its end in `done` (page-locked host
// memory, capi_batch.hip run_map) when that is given
__global__ __launch_bounds__(256)
void geodesic_chain_kernel(size_t n, const double* __restrict__ pi, const double* __restrict__ psi, const double* __restrict__ pci,
                           const double* __restrict__ pa, const double* __restrict__ pal,
                           const double* __restrict__ pbe, Geod* pg, int* pe, int* po, sim5gpu_geodesic_chain* pc,
                           DiskConsts d, int have_disk, int* done)
{
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j < 2 * n) geodesic_chain_lane(j, pi, psi, pci, pa, pal, pbe, pg, pe, po, pc, d, have_disk != 0);      // s5_chain.hpp
    if (done) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

} // namespace S5NS

// `disk`: the process's disk constants (s5::DiskConsts of capi_core.hip; the two namespaces' structs are the same bytes).
// `done` != nullptr: n <= 128 rays, one workgroup, the kernel raises *done at its end.
int s5_launch_geodesic_chain_fast(size_t n, const double* incl, const double* sin_i, const double* cos_i, const double* a, const double* alpha, const double* beta,
                                  void* geod, int* err, int* ok, sim5gpu_geodesic_chain* chain,
                                  const void* disk, size_t disk_bytes, int have_disk, int* done, hipStream_t stream)
{
    using namespace S5NS;
    static_assert(sizeof(Geod) == sizeof(sim5gpu_geodesic), "geodesic record layout");
    DiskConsts d;
    if (disk_bytes != sizeof d) return (int)hipErrorInvalidValue;
    memcpy(&d, disk, sizeof d);
    if (done && 2 * n > 256) return (int)hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((2 * n + 255) / 256);
    hipLaunchKernelGGL(geodesic_chain_kernel, dim3(blocks), dim3(256), 0, stream, n, incl, sin_i, cos_i, a, alpha, beta, (Geod*)geod, err, ok, chain,
                       d, have_disk, done);
    return (int)hipGetLastError();
}
#endif
