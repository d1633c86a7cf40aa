// s5_chain.hpp -- one lane of the per-ray record of the SIM5 scalar API (ref examples/04-disk-image-eqplane/disk-image.c:
// 62-100): geodesic_init_inf, then for ONE crossing order the equatorial crossing, the radius there, gfactorK and -- when the
// disk model has been set up -- disk_nt_flux, each by the routine the single entry point calls with the same arguments.
// Lane pair (2 i, 2 i + 1) = ray i, orders 0 and 1; both lanes set the geodesic up (a single ray is a chain of dependent FP64
// operations, and its latency -- not the launch -- is what a caller of the scalar API waits for).  Compiled in both
// arithmetic variants: strict in capi_batch.hip (sim5gpu_geodesic_init_inf_chain), fast in k_chain.hip (..._chain_fast).
#pragma once
#include "s5_disk.hpp"
#include "../../include/sim5gpu.h"

namespace S5NS {

S5_DEV void geodesic_chain_lane(size_t j, const double* __restrict__ pi, const double* __restrict__ psi, const double* __restrict__ pci,
                                const double* __restrict__ pa, const double* __restrict__ pal, const double* __restrict__ pbe,
                                Geod* pg, int* pe, int* po, sim5gpu_geodesic_chain* pc, const DiskConsts& d, bool have_disk)
{
    const size_t i = j >> 1;
    const int k = (int)(j & 1);
    Geod gd = pg[i];
    GeodCache cache;
    int err = 0;
    // (psi, pci: sin and cos of the inclination from the host's libm, capi_batch.hip host_sincos)
    const bool ok_ = init_inf(pi[i], psi[i], pci[i], pa[i], pal[i], pbe[i], gd, err, cache);
    // the record's values in registers, stored ONCE at the end: the record may be page-locked host memory the kernel writes
    // over the bus (capi_util.hpp DevBuf), and nothing is read back through that pointer
    double P = NAN, r = NAN, gf = NAN, flux = NAN;
    int have_r = 0;
    if (ok_) {
        // K(mm) and the inverse cn of the observer's position come from init_inf (GeodCache): the very values the
        // crossing search would form again from the same expressions (the image kernels rely on the same identity)
        P = midplane_crossing(gd, k, cache);
        if (!isnan(P)) {
            r = position_rad(gd, P);
            have_r = 1;
            if (!isnan(r)) {
                gf = gfactor_kepler(r, pa[i], gd.l);
                if (have_disk) flux = disk_flux(d, r);
            }
        }
    }
    sim5gpu_geodesic_chain* c = &pc[i];
    c->P[k] = P; c->r[k] = r; c->g[k] = gf; c->flux[k] = flux; c->have_r[k] = have_r;
    // both lanes of a ray (j even / odd: one wave) have read pg[i] above; lane 0 writes the geodesic back after the wave has
    // passed this point together
    __builtin_amdgcn_wave_barrier();
    if (k == 0) {
        c->flux_valid = have_disk ? 1 : 0; c->valid = ok_ ? 1 : 0;
        c->a = pa[i]; c->l = gd.l;
        pe[i] = err;
        po[i] = ok_ ? 1 : 0;
        pg[i] = gd;
    }
}

} // namespace S5NS
