// s5_disk.hpp -- Novikov-Thorne thin disk (Page & Thorne 1974 flux) for the gfx950 kernels.
//
// Restated from the reference (ref: /root/reference/src/sim5disk-nt.c:37-146, 260-266).  The
// reference keeps the model in file-static *float* variables; everything that depends only on
// them is folded on the host (sim5gpu_disk_nt_setup / the image launchers, in double, with the
// host libm, in the reference's operation order) into DiskConsts, which reaches the kernel as
// kernel arguments, i.e. in SGPRs: the profile is a closed form (4 logs per hit), so there is no
// radial table to stage in LDS -- the constants are wave-uniform scalars.
#pragma once
#include "s5_geod.hpp"
#include "kernels.hpp"

namespace S5NS {

using s5abi::DiskConsts;

S5_DEV double disk_flux(const DiskConsts& d, double r)                 // ref :110-146
{
    if (r <= d.rms) return 0.0;
    const double a = d.a;
    const double x = sqrt_pos(r);                     // r > rms > 0
#if S5_FAST
    // same expression with the constant divisors replaced by their host-computed reciprocals and the two
    // prefactor divisions merged into one
    const double f0 = x - d.x0 - 1.5 * a * mlog(x * d.inv_x0);
    const double f1 = d.p1 * mlog((x - d.x1) * d.inv_d1);
    const double f2 = d.p2 * mlog((x - d.x2) * d.inv_d2);
    const double f3 = d.p3 * mlog((x - d.x3) * d.inv_d3);
    const double F = mdiv(1.5, (4. * M_PI * r) * (x * x * (x * x * x - 3. * x + 2. * a))) * (f0 - f1 - f2 - f3);
    return d.scale * F;
#else
    const double f0 = x - d.x0 - 1.5 * a * mlog(mdiv(x, d.x0));
    const double f1 = d.p1 * mlog(mdiv(x - d.x1, d.d1));
    const double f2 = d.p2 * mlog(mdiv(x - d.x2, d.d2));
    const double f3 = d.p3 * mlog(mdiv(x - d.x3, d.d3));
    const double F = mdiv(mdiv(1., 4. * M_PI * r) * 1.5, x * x * (x * x * x - 3. * x + 2. * a)) * (f0 - f1 - f2 - f3);
    return mdiv(9.1721376255e+28 * F * d.mdot, d.mass);
#endif
}

S5_DEV double disk_ell(const DiskConsts& d, double r)                  // ref :260-266
{
    const double a = d.a;
    r = fmax(d.rms, r);
    return (r * r - 2. * a * sqrt(r) + a * a) / (sqrt(r) * r - 2. * sqrt(r) + a);
}

} // namespace S5NS
