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

S5_DEV void geodesic_chain_lane(size_t j, const double* __restrict__ pi, const double* __restrict__ pa,
                                const double* __restrict__ pal, const double* __restrict__ pbe, Geod* pg, int* pe, int* po,
                                sim5gpu_geodesic_chain* pc, const DiskConsts& d, bool have_disk)
{
    const size_t i = j >> 1;
    const int k = (int)(j & 1);
    Geod gd = pg[i];
    GeodCache cache;
    int err = 0;
    const double inc = pi[i];
    const bool ok_ = init_inf(inc, sin(inc), cos(inc), pa[i], pal[i], pbe[i], gd, err, cache);
    sim5gpu_geodesic_chain* c = &pc[i];
    c->P[k] = NAN; c->r[k] = NAN; c->g[k] = NAN; c->flux[k] = NAN; c->have_r[k] = 0;
    if (ok_) {
        // K(mm) and the inverse cn of the observer's position come from init_inf (GeodCache): the very values the
        // crossing search would form again from the same expressions (the image kernels rely on the same identity)
        c->P[k] = midplane_crossing(gd, k, cache);
        if (!isnan(c->P[k])) {
            c->r[k] = position_rad(gd, c->P[k]);
            c->have_r[k] = 1;
            if (!isnan(c->r[k])) {
                c->g[k] = gfactor_kepler(c->r[k], pa[i], gd.l);
                if (have_disk) c->flux[k] = disk_flux(d, c->r[k]);
            }
        }
    }
    if (k == 0) {
        c->flux_valid = have_disk ? 1 : 0; c->valid = ok_ ? 1 : 0;
        c->a = pa[i]; c->l = gd.l;
        pe[i] = err;
        po[i] = ok_ ? 1 : 0;
        // both lanes read pg[i] above; the geodesic is written back by lane 0 after its partner has read it too: the two
        // lanes of a ray sit in one wave (j even / odd), which executes the read before the write in program order
        pg[i] = gd;
    }
}

} // namespace S5NS
