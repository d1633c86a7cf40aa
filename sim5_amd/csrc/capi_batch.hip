// capi_batch.hip -- batch forms of the SIM5 per-ray functions (group (1) of include/sim5gpu.h).
//
// Every entry point: validate -> upload the caller's host arrays -> one kernel, one lane per ray,
// calling the same device routines the image kernels inline -> download.  These exist so that a
// host program written against the SIM5 scalar API (and the parity tests) can reach each routine
// through the C-ABI; throughput work goes through the whole-job kernels instead.
#include <vector>
#include <math.h>
#include "capi_util.hpp"
#include "s5_disk.hpp"
#include "s5_chain.hpp"
#include "s5_raytrace.hpp"
#include "s5_polar.hpp"
#include "s5_azimuth.hpp"

namespace s5 {

// (the batch runner -- one lane per element, small batches announcing their own end -- is run_batch of capi_util.hpp)
template <typename F>
static int run_map(size_t n, F body, const char* what) { return run_batch(n, body, what); }

static int arg_error(const char* fn)
{
    snprintf(g_err, sizeof g_err, "%s: NULL pointer argument", fn);
    return SIM5GPU_E_ARG;
}

#define S5_NEED(fn, cond) do { if (!(cond)) return arg_error(fn); } while (0)
#define S5_DEVICE_OR_FAIL() do { if (!have_device()) return SIM5GPU_E_NO_DEVICE; } while (0)
#define S5_BUFS_OK(fn, cond) do { if (!(cond)) { snprintf(g_err, sizeof g_err, "%s: device allocation/copy failed", fn); return SIM5GPU_E_HIP; } } while (0)
#define S5_RUN(n, what, ...) do { int rc_ = run_map(n, __VA_ARGS__, what); if (rc_) return rc_; } while (0)

// sin and cos of the inclinations from the HOST's libm -- the numbers the reference itself forms (ref src/sim5kerr-geod.c:
// 73-77: cos(i), sin(i) by glibc): the last bit of cos(i) enters q and, for a ray with l = 0, decides its class (s5_geod.hpp).
// The whole-job kernels have always taken them from the host (one inclination per job); the batch entry points now do too.
// A caller's rays nearly always share one inclination: consecutive equal values cost a comparison, not two libm calls.
static void host_sincos(size_t n, const double* incl, std::vector<double>& s, std::vector<double>& c)
{
    s.resize(n); c.resize(n);
    double last = 0.0, ls = 0.0, lc = 1.0;
    bool have = false;
    for (size_t i = 0; i < n; ++i) {
        if (!have || memcmp(&incl[i], &last, sizeof last) != 0) { last = incl[i]; reference_sincos(last, ls, lc); have = true; }
        s[i] = ls; c[i] = lc;
    }
}

// Total disk luminosity in Eddington units (ref src/sim5disk-nt.c:151-188): the reference's Simpson rule on the
// refined trapezoid rule (src/sim5integration.c:26-52 stage rule with its running abscissa x += del, :96-133:
// at most 23 stages, relative accuracy 1e-5, at least 4 stages).  The abscissae of a stage are generated on the
// host exactly as the reference's loop does, the integrand (flux, u_t) is evaluated for the whole stage in one
// launch, the values are summed in the reference's order.
int disk_lumi(const DiskConsts& d, double* lumi)
{
    const float disk_rmax = 1e5;
    const double lo = log(d.rms), hi = log((double)disk_rmax), acc = 1e-5;
    double s = 0.0, st = 0.0, ost = -1.e50, os = -1.e50;
    std::vector<double> x, f;
    for (int n = 1; n <= 23; n++) {
        double del = 0.0;
        x.clear();
        if (n == 1) { x.push_back(hi); x.push_back(lo); }
        else {
            int it = 1;
            for (int j = 1; j < n - 1; j++) it <<= 1;
            del = (hi - lo) / (double)it;
            double xx = lo + 0.5 * del;
            for (int j = 1; j <= it; j++, xx += del) x.push_back(xx);
        }
        const size_t m = x.size();
        f.resize(m);
        {
            DevBuf<double> dx(x.data(), m), df(m);
            S5_BUFS_OK("disk_nt_lumi", dx.ok() && df.ok());
            const double* px = dx.ptr; double* pf = df.ptr; const DiskConsts dd = d;
            S5_RUN(m, "disk_nt_lumi", [=] __device__(size_t i) { pf[i] = disk_lumi_integrand(dd, px[i]); });
            S5_HIP(df.to_host(f.data()));
        }
        if (n == 1) st = 0.5 * (hi - lo) * (f[0] + f[1]);
        else {
            double sum = 0.0;
            for (size_t j = 0; j < m; j++) sum += f[j];
            st = 0.5 * (st + del * sum);
        }
        s = (4. * st - ost) / 3.;
        if (n > 3 && ((fabs(s - os) < acc * fabs(os)) || ((s == 0.) && (os == 0.)))) break;
        os = s;
        ost = st;
    }
    const double grav_radius = 1.476716e+05, L_Edd = 1.257142540e+38;      // ref src/sim5const.h:32,51
    const double L = s * ((d.mass * grav_radius) * (d.mass * grav_radius));
    *lumi = L / (L_Edd * d.mass);
    return SIM5GPU_OK;
}

} // namespace s5

using namespace s5;

extern "C" {

// ------------------------------------------------------------------------------------------
// geodesics
// ------------------------------------------------------------------------------------------
int sim5gpu_geodesic_init_inf(size_t n, const double* incl, const double* a, const double* alpha,
                              const double* beta, sim5gpu_geodesic* g, int* error, int* ok)
{
    S5_NEED("geodesic_init_inf", incl && a && alpha && beta && g);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    std::vector<double> hs, hc;
    host_sincos(n, incl, hs, hc);
    DevBuf<double> di(incl, n), dsi(hs.data(), n), dci(hc.data(), n), da(a, n), dal(alpha, n), dbe(beta, n);
    DevBuf<Geod> dg((const Geod*)g, n);            // keep caller's bytes in fields we never write
    DevBuf<int> derr(n), dok(n);
    S5_BUFS_OK("geodesic_init_inf", di.ok() && dsi.ok() && dci.ok() && da.ok() && dal.ok() && dbe.ok() && dg.ok() && derr.ok() && dok.ok());
    const double *pi = di.ptr, *psi = dsi.ptr, *pci = dci.ptr, *pa = da.ptr, *pal = dal.ptr, *pbe = dbe.ptr;
    Geod* pg = dg.ptr; int *pe = derr.ptr, *po = dok.ptr;
    S5_RUN(n, "geodesic_init_inf", [=] __device__(size_t i) {
        Geod gd = pg[i];
        GeodCache cache;
        int err = 0;                                // on failure *error receives the GD_* code
        const bool ok_ = init_inf(pi[i], psi[i], pci[i], pa[i], pal[i], pbe[i], gd, err, cache);
        pg[i] = gd;
        pe[i] = err;
        po[i] = ok_ ? 1 : 0;
    });
    S5_HIP(dg.to_host((Geod*)g));
    if (error) S5_HIP(derr.to_host(error));
    if (ok) S5_HIP(dok.to_host(ok));
    return SIM5GPU_OK;
}

/* geodesic_init_inf AND what the caller loop of ref examples/04-disk-image-eqplane/disk-image.c:62-100 asks of the geodesic
 * next, in the same launch: the equatorial crossings of orders 0 and 1, the radii there, gfactorK(r, a, g.l) and -- when the
 * disk model has been set up -- disk_nt_flux(r).  Each value is produced by the very device routine the single entry point
 * calls with the same arguments, so a host that answers the follow-up calls from this record (sim5_amd/host/sim5lib.c does,
 * after checking that the arguments are the ones the record was made for, bit for bit) returns the same numbers with one
 * round trip to the GPU per ray instead of five. */
int sim5gpu_geodesic_init_inf_chain(size_t n, const double* incl, const double* a, const double* alpha, const double* beta,
                                    sim5gpu_geodesic* g, int* error, int* ok, sim5gpu_geodesic_chain* chain)
{
    S5_NEED("geodesic_init_inf_chain", incl && a && alpha && beta && g && chain);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    std::vector<double> hs, hc;
    host_sincos(n, incl, hs, hc);
    DevBuf<double> di(incl, n), dsi(hs.data(), n), dci(hc.data(), n), da(a, n), dal(alpha, n), dbe(beta, n);
    DevBuf<Geod> dg((const Geod*)g, n);            // keep caller's bytes in fields we never write
    DevBuf<int> derr(n), dok(n);
    DevBuf<sim5gpu_geodesic_chain> dch(n);
    S5_BUFS_OK("geodesic_init_inf_chain", di.ok() && dsi.ok() && dci.ok() && da.ok() && dal.ok() && dbe.ok() && dg.ok() && derr.ok() && dok.ok() && dch.ok());
    const double *pi = di.ptr, *psi = dsi.ptr, *pci = dci.ptr, *pa = da.ptr, *pal = dal.ptr, *pbe = dbe.ptr;
    Geod* pg = dg.ptr; int *pe = derr.ptr, *po = dok.ptr;
    sim5gpu_geodesic_chain* pc = dch.ptr;
    const bool have_disk = g_disk.ready != 0;
    const DiskConsts d = g_disk;
    S5_RUN(2 * n, "geodesic_init_inf_chain", [=] __device__(size_t j) {
        geodesic_chain_lane(j, pi, psi, pci, pa, pal, pbe, pg, pe, po, pc, d, have_disk);          // s5_chain.hpp
    });
    S5_HIP(dg.to_host((Geod*)g));
    if (error) S5_HIP(derr.to_host(error));
    if (ok) S5_HIP(dok.to_host(ok));
    S5_HIP(dch.to_host(chain));
    return SIM5GPU_OK;
}

/* The same record in the arithmetic of the FAST variant (k_chain.hip): what a caller of the scalar API waits for is the latency
 * of one ray's dependent FP64 chain, and the fast routines make it three to four times shorter.  Values agree with the strict
 * entry point's to ~1e-12 relative; the host shim uses this one unless SIM5_SHIM_STRICT is set. */
int sim5gpu_geodesic_init_inf_chain_fast(size_t n, const double* incl, const double* a, const double* alpha, const double* beta,
                                         sim5gpu_geodesic* g, int* error, int* ok, sim5gpu_geodesic_chain* chain)
{
    S5_NEED("geodesic_init_inf_chain_fast", incl && a && alpha && beta && g && chain);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    std::vector<double> hs, hc;
    host_sincos(n, incl, hs, hc);
    DevBuf<double> di(incl, n), dsi(hs.data(), n), dci(hc.data(), n), da(a, n), dal(alpha, n), dbe(beta, n);
    DevBuf<Geod> dg((const Geod*)g, n);            // keep caller's bytes in fields we never write
    DevBuf<int> derr(n), dok(n);
    DevBuf<sim5gpu_geodesic_chain> dch(n);
    S5_BUFS_OK("geodesic_init_inf_chain_fast", di.ok() && dsi.ok() && dci.ok() && da.ok() && dal.ok() && dbe.ok() && dg.ok() && derr.ok() && dok.ok() && dch.ok());
    int* done = (2 * n <= 256) ? take_done_word() : nullptr;
    hipStream_t stream = thread_stream();
    hipError_t e = (hipError_t)s5_launch_geodesic_chain_fast(n, di.ptr, dsi.ptr, dci.ptr, da.ptr, dal.ptr, dbe.ptr, dg.ptr, derr.ptr, dok.ptr, dch.ptr,
                                                             &g_disk, sizeof g_disk, g_disk.ready != 0, done, stream);
    if (e == hipSuccess) e = done ? wait_done_word(done, stream) : hipStreamSynchronize(stream);
    if (done) arena().give_pinned();
    if (e != hipSuccess) { set_error("geodesic_init_inf_chain_fast", e); return SIM5GPU_E_HIP; }
    S5_HIP(dg.to_host((Geod*)g));
    if (error) S5_HIP(derr.to_host(error));
    if (ok) S5_HIP(dok.to_host(ok));
    S5_HIP(dch.to_host(chain));
    return SIM5GPU_OK;
}

int sim5gpu_geodesic_init_src(size_t n, const double* a, const double* r, const double* m,
                              const double* k, const int* ppc, sim5gpu_geodesic* g, int* error, int* ok)
{
    S5_NEED("geodesic_init_src", a && r && m && k && ppc && g);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> da(a, n), dr(r, n), dm(m, n), dk(k, 4 * n);
    DevBuf<int> dp(ppc, n), derr(n), dok(n);
    DevBuf<Geod> dg((const Geod*)g, n);
    S5_BUFS_OK("geodesic_init_src", da.ok() && dr.ok() && dm.ok() && dk.ok() && dp.ok() && derr.ok() && dok.ok() && dg.ok());
    const double *pa = da.ptr, *pr = dr.ptr, *pm = dm.ptr, *pk = dk.ptr;
    const int* pp = dp.ptr; Geod* pg = dg.ptr; int *pe = derr.ptr, *po = dok.ptr;
    S5_RUN(n, "geodesic_init_src", [=] __device__(size_t i) {
        Geod gd = pg[i];
        int err = 0;
        const double kk[4] = { pk[4 * i], pk[4 * i + 1], pk[4 * i + 2], pk[4 * i + 3] };
        const bool ok_ = init_src(pa[i], pr[i], pm[i], kk, pp[i], gd, err);
        pg[i] = gd; pe[i] = err; po[i] = ok_ ? 1 : 0;
    });
    S5_HIP(dg.to_host((Geod*)g));
    if (error) S5_HIP(derr.to_host(error));
    if (ok) S5_HIP(dok.to_host(ok));
    return SIM5GPU_OK;
}

int sim5gpu_geodesic_find_midplane_crossing(size_t n, const sim5gpu_geodesic* g, const int* order, double* P)
{
    S5_NEED("geodesic_find_midplane_crossing", g && order && P);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<Geod> dg((const Geod*)g, n); DevBuf<int> dord(order, n); DevBuf<double> dP(n);
    S5_BUFS_OK("geodesic_find_midplane_crossing", dg.ok() && dord.ok() && dP.ok());
    const Geod* pg = dg.ptr; const int* po = dord.ptr; double* pP = dP.ptr;
    S5_RUN(n, "geodesic_find_midplane_crossing", [=] __device__(size_t i) {
        GeodCache none; none.valid = false; none.K = none.icn_i = none.u_i = 0.0;
        pP[i] = midplane_crossing(pg[i], po[i], none);
    });
    S5_HIP(dP.to_host(P));
    return SIM5GPU_OK;
}

int sim5gpu_geodesic_P_int(size_t n, const sim5gpu_geodesic* g, const double* r, const int* ppc, double* P)
{
    S5_NEED("geodesic_P_int", g && r && ppc && P);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<Geod> dg((const Geod*)g, n); DevBuf<double> dr(r, n), dP(n); DevBuf<int> dp(ppc, n);
    S5_BUFS_OK("geodesic_P_int", dg.ok() && dr.ok() && dP.ok() && dp.ok());
    const Geod* pg = dg.ptr; const double* pr = dr.ptr; const int* pp = dp.ptr; double* pP = dP.ptr;
    S5_RUN(n, "geodesic_P_int", [=] __device__(size_t i) { pP[i] = P_int(pg[i], pr[i], pp[i]); });
    S5_HIP(dP.to_host(P));
    return SIM5GPU_OK;
}

#define S5_GEOD_P_FN(NAME, DEVFN)                                                              \
int NAME(size_t n, const sim5gpu_geodesic* g, const double* P, double* out)                    \
{                                                                                              \
    S5_NEED(#NAME, g && P && out);                                                             \
    if (n == 0) return SIM5GPU_OK;                                                             \
    S5_DEVICE_OR_FAIL();                                                                       \
    DevBuf<Geod> dg((const Geod*)g, n); DevBuf<double> dP(P, n), dout(n);                      \
    S5_BUFS_OK(#NAME, dg.ok() && dP.ok() && dout.ok());                                        \
    const Geod* pg = dg.ptr; const double* pP = dP.ptr; double* po = dout.ptr;                 \
    S5_RUN(n, #NAME, [=] __device__(size_t i) { po[i] = DEVFN(pg[i], pP[i]); });               \
    S5_HIP(dout.to_host(out));                                                                 \
    return SIM5GPU_OK;                                                                         \
}
S5_GEOD_P_FN(sim5gpu_geodesic_position_rad, position_rad)
S5_GEOD_P_FN(sim5gpu_geodesic_position_pol, position_pol)
S5_GEOD_P_FN(sim5gpu_geodesic_dm_sign, dm_sign)
#undef S5_GEOD_P_FN

int sim5gpu_geodesic_momentum(size_t n, const sim5gpu_geodesic* g, const double* P, const double* r,
                              const double* m, double* k)
{
    S5_NEED("geodesic_momentum", g && P && r && m && k);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<Geod> dg((const Geod*)g, n); DevBuf<double> dP(P, n), dr(r, n), dm(m, n), dk(k, 4 * n);
    S5_BUFS_OK("geodesic_momentum", dg.ok() && dP.ok() && dr.ok() && dm.ok() && dk.ok());
    const Geod* pg = dg.ptr; const double *pP = dP.ptr, *pr = dr.ptr, *pm = dm.ptr; double* pk = dk.ptr;
    S5_RUN(n, "geodesic_momentum", [=] __device__(size_t i) {
        double kk[4] = { pk[4 * i], pk[4 * i + 1], pk[4 * i + 2], pk[4 * i + 3] };
        momentum(pg[i], pP[i], pr[i], pm[i], kk);
        pk[4 * i] = kk[0]; pk[4 * i + 1] = kk[1]; pk[4 * i + 2] = kk[2]; pk[4 * i + 3] = kk[3];
    });
    S5_HIP(dk.to_host(k));
    return SIM5GPU_OK;
}

int sim5gpu_geodesic_follow(size_t n, const sim5gpu_geodesic* g, const double* step, double* P,
                            double* r, double* m, int* status)
{
    S5_NEED("geodesic_follow", g && step && P && r && m);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<Geod> dg((const Geod*)g, n); DevBuf<double> ds(step, n), dP(P, n), dr(r, n), dm(m, n); DevBuf<int> dst(n);
    S5_BUFS_OK("geodesic_follow", dg.ok() && ds.ok() && dP.ok() && dr.ok() && dm.ok() && dst.ok());
    const Geod* pg = dg.ptr; const double* ps = ds.ptr; double *pP = dP.ptr, *pr = dr.ptr, *pm = dm.ptr; int* pst = dst.ptr;
    S5_RUN(n, "geodesic_follow", [=] __device__(size_t i) {
        double PP = pP[i], rr = pr[i], mm = pm[i]; int st = 0;
        follow(pg[i], ps[i], PP, rr, mm, st);
        pP[i] = PP; pr[i] = rr; pm[i] = mm; pst[i] = st;
    });
    S5_HIP(dP.to_host(P)); S5_HIP(dr.to_host(r)); S5_HIP(dm.to_host(m));
    if (status) S5_HIP(dst.to_host(status));
    return SIM5GPU_OK;
}

// ------------------------------------------------------------------------------------------
// photon kinematics, metric, tetrads
// ------------------------------------------------------------------------------------------
int sim5gpu_photon_momentum(size_t n, const double* a, const double* r, const double* m, const double* l,
                            const double* q, const double* r_sign, const double* m_sign, double* k)
{
    S5_NEED("photon_momentum", a && r && m && l && q && r_sign && m_sign && k);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> da(a, n), dr(r, n), dm(m, n), dl(l, n), dq(q, n), drs(r_sign, n), dms(m_sign, n), dk(k, 4 * n);
    S5_BUFS_OK("photon_momentum", da.ok() && dr.ok() && dm.ok() && dl.ok() && dq.ok() && drs.ok() && dms.ok() && dk.ok());
    const double *pa = da.ptr, *pr = dr.ptr, *pm = dm.ptr, *pl = dl.ptr, *pq = dq.ptr, *prs = drs.ptr, *pms = dms.ptr;
    double* pk = dk.ptr;
    S5_RUN(n, "photon_momentum", [=] __device__(size_t i) {
        double kk[4] = { pk[4 * i], pk[4 * i + 1], pk[4 * i + 2], pk[4 * i + 3] };
        photon_momentum(pa[i], pr[i], pm[i], pl[i], pq[i], prs[i], pms[i], kk);
        pk[4 * i] = kk[0]; pk[4 * i + 1] = kk[1]; pk[4 * i + 2] = kk[2]; pk[4 * i + 3] = kk[3];
    });
    S5_HIP(dk.to_host(k));
    return SIM5GPU_OK;
}

int sim5gpu_photon_motion_constants(size_t n, const double* a, const double* r, const double* m,
                                    const double* k, double* L, double* Q)
{
    S5_NEED("photon_motion_constants", a && r && m && k && L && Q);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> da(a, n), dr(r, n), dm(m, n), dk(k, 4 * n), dL(n), dQ(n);
    S5_BUFS_OK("photon_motion_constants", da.ok() && dr.ok() && dm.ok() && dk.ok() && dL.ok() && dQ.ok());
    const double *pa = da.ptr, *pr = dr.ptr, *pm = dm.ptr, *pk = dk.ptr; double *pL = dL.ptr, *pQ = dQ.ptr;
    S5_RUN(n, "photon_motion_constants", [=] __device__(size_t i) {
        const double kk[4] = { pk[4 * i], pk[4 * i + 1], pk[4 * i + 2], pk[4 * i + 3] };
        double L_, Q_;
        photon_motion_constants(pa[i], pr[i], pm[i], kk, L_, Q_);
        pL[i] = L_; pQ[i] = Q_;
    });
    S5_HIP(dL.to_host(L)); S5_HIP(dQ.to_host(Q));
    return SIM5GPU_OK;
}

int sim5gpu_photon_carter_const(size_t n, const double* k, const sim5gpu_metric* metric, double* Q)
{
    S5_NEED("photon_carter_const", k && metric && Q);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dk(k, 4 * n), dQ(n); DevBuf<Metric> dmt((const Metric*)metric, n);
    S5_BUFS_OK("photon_carter_const", dk.ok() && dQ.ok() && dmt.ok());
    const double* pk = dk.ptr; const Metric* pg = dmt.ptr; double* pQ = dQ.ptr;
    S5_RUN(n, "photon_carter_const", [=] __device__(size_t i) {
        const double kk[4] = { pk[4 * i], pk[4 * i + 1], pk[4 * i + 2], pk[4 * i + 3] };
        pQ[i] = carter_constant(kk, pg[i]);
    });
    S5_HIP(dQ.to_host(Q));
    return SIM5GPU_OK;
}

#define S5_UNARY_FN(NAME, EXPR)                                                                \
int NAME(size_t n, const double* x, double* out)                                               \
{                                                                                              \
    S5_NEED(#NAME, x && out);                                                                  \
    if (n == 0) return SIM5GPU_OK;                                                             \
    S5_DEVICE_OR_FAIL();                                                                       \
    DevBuf<double> dx(x, n), dout(n);                                                          \
    S5_BUFS_OK(#NAME, dx.ok() && dout.ok());                                                   \
    const double* px = dx.ptr; double* po = dout.ptr;                                          \
    S5_RUN(n, #NAME, [=] __device__(size_t i) { const double v = px[i]; po[i] = (EXPR); });    \
    S5_HIP(dout.to_host(out));                                                                 \
    return SIM5GPU_OK;                                                                         \
}
S5_UNARY_FN(sim5gpu_r_bh, r_horizon(v))
S5_UNARY_FN(sim5gpu_r_ms, r_isco(v))
S5_UNARY_FN(sim5gpu_r_mb, (2. - v) + 2. * sqrt(1. - v))                    // ref src/sim5kerr.c:1007-1017
S5_UNARY_FN(sim5gpu_r_ph, 2.0 * (1.0 + cos(2. / 3. * acos(-v))))          // ref src/sim5kerr.c:1020-1031
#undef S5_UNARY_FN

#define S5_BINARY_FN(NAME, EXPR)                                                               \
int NAME(size_t n, const double* x, const double* y, double* out)                              \
{                                                                                              \
    S5_NEED(#NAME, x && y && out);                                                             \
    if (n == 0) return SIM5GPU_OK;                                                             \
    S5_DEVICE_OR_FAIL();                                                                       \
    DevBuf<double> dx(x, n), dy(y, n), dout(n);                                                \
    S5_BUFS_OK(#NAME, dx.ok() && dy.ok() && dout.ok());                                        \
    const double *px = dx.ptr, *py = dy.ptr; double* po = dout.ptr;                            \
    S5_RUN(n, #NAME, [=] __device__(size_t i) { const double u = px[i], v = py[i]; po[i] = (EXPR); }); \
    S5_HIP(dout.to_host(out));                                                                 \
    return SIM5GPU_OK;                                                                         \
}
// OmegaK uses pow(r, 1.5) in the reference; r*sqrt(r) is within 1 ulp of it
S5_BINARY_FN(sim5gpu_OmegaK, omega_kepler(u, v))
S5_BINARY_FN(sim5gpu_ellK, ell_kepler(u, v))
#undef S5_BINARY_FN

int sim5gpu_Omega_from_ell(size_t n, const double* ell, const sim5gpu_metric* metric, double* Omega)
{
    S5_NEED("Omega_from_ell", ell && metric && Omega);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dl(ell, n), dout(n); DevBuf<Metric> dmt((const Metric*)metric, n);
    S5_BUFS_OK("Omega_from_ell", dl.ok() && dout.ok() && dmt.ok());
    const double* pl = dl.ptr; double* po = dout.ptr; const Metric* pg = dmt.ptr;
    S5_RUN(n, "Omega_from_ell", [=] __device__(size_t i) { po[i] = omega_from_ell(pl[i], pg[i]); });
    S5_HIP(dout.to_host(Omega));
    return SIM5GPU_OK;
}

int sim5gpu_dotprod(size_t n, const double* v1, const double* v2, const sim5gpu_metric* metric, double* out)
{
    S5_NEED("dotprod", v1 && v2 && out);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> d1(v1, 4 * n), d2(v2, 4 * n), dout(n); DevBuf<Metric> dmt((const Metric*)metric, metric ? n : 0);
    S5_BUFS_OK("dotprod", d1.ok() && d2.ok() && dout.ok() && dmt.ok());
    const double *p1 = d1.ptr, *p2 = d2.ptr; double* po = dout.ptr; const Metric* pg = dmt.ptr;
    S5_RUN(n, "dotprod", [=] __device__(size_t i) {
        const double u[4] = { p1[4 * i], p1[4 * i + 1], p1[4 * i + 2], p1[4 * i + 3] };
        const double w[4] = { p2[4 * i], p2[4 * i + 1], p2[4 * i + 2], p2[4 * i + 3] };
        po[i] = pg ? dot(u, w, pg[i]) : (-u[0] * w[0] + u[1] * w[1] + u[2] * w[2] + u[3] * w[3]);
    });
    S5_HIP(dout.to_host(out));
    return SIM5GPU_OK;
}

int sim5gpu_vector_norm_to(size_t n, double* v, const double* norm, const sim5gpu_metric* metric)
{
    S5_NEED("vector_norm_to", v && norm);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dv(v, 4 * n), dn(norm, n); DevBuf<Metric> dmt((const Metric*)metric, metric ? n : 0);
    S5_BUFS_OK("vector_norm_to", dv.ok() && dn.ok() && dmt.ok());
    double* pv = dv.ptr; const double* pn = dn.ptr; const Metric* pg = dmt.ptr;
    S5_RUN(n, "vector_norm_to", [=] __device__(size_t i) {                  // ref src/sim5kerr.c:553-573
        const double u[4] = { pv[4 * i], pv[4 * i + 1], pv[4 * i + 2], pv[4 * i + 3] };
        const double N = pg ? dot(u, u, pg[i]) : (-u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + u[3] * u[3]);
        for (int c = 0; c < 4; ++c) pv[4 * i + c] = u[c] * sqrt(pn[i] / N);
    });
    S5_HIP(dv.to_host(v));
    return SIM5GPU_OK;
}

int sim5gpu_gfactorK(size_t n, const double* r, const double* a, const double* l, double* g)
{
    S5_NEED("gfactorK", r && a && l && g);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dr(r, n), da(a, n), dl(l, n), dg(n);
    S5_BUFS_OK("gfactorK", dr.ok() && da.ok() && dl.ok() && dg.ok());
    const double *pr = dr.ptr, *pa = da.ptr, *pl = dl.ptr; double* pg = dg.ptr;
    S5_RUN(n, "gfactorK", [=] __device__(size_t i) { pg[i] = gfactor_kepler(pr[i], pa[i], pl[i]); });
    S5_HIP(dg.to_host(g));
    return SIM5GPU_OK;
}

int sim5gpu_kerr_metric(size_t n, const double* a, const double* r, const double* m, sim5gpu_metric* metric)
{
    S5_NEED("kerr_metric", a && r && m && metric);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> da(a, n), dr(r, n), dm(m, n); DevBuf<Metric> dmt(n);
    S5_BUFS_OK("kerr_metric", da.ok() && dr.ok() && dm.ok() && dmt.ok());
    const double *pa = da.ptr, *pr = dr.ptr, *pm = dm.ptr; Metric* pg = dmt.ptr;
    S5_RUN(n, "kerr_metric", [=] __device__(size_t i) { Metric g; kerr_metric(pa[i], pr[i], pm[i], g); pg[i] = g; });
    S5_HIP(dmt.to_host((Metric*)metric));
    return SIM5GPU_OK;
}

int sim5gpu_kerr_connection(size_t n, const double* a, const double* r, const double* m, double* G)
{
    S5_NEED("kerr_connection", a && r && m && G);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> da(a, n), dr(r, n), dm(m, n), dG(64 * n);
    S5_BUFS_OK("kerr_connection", da.ok() && dr.ok() && dm.ok() && dG.ok());
    const double *pa = da.ptr, *pr = dr.ptr, *pm = dm.ptr; double* pG = dG.ptr;
    S5_RUN(n, "kerr_connection", [=] __device__(size_t i) {
        Conn c; kerr_connection(pa[i], pr[i], pm[i], c);
        conn_to_dense(c, pG + 64 * i);
    });
    S5_HIP(dG.to_host(G));
    return SIM5GPU_OK;
}

int sim5gpu_tetrad_zamo(size_t n, const sim5gpu_metric* metric, sim5gpu_tetrad* t)
{
    S5_NEED("tetrad_zamo", metric && t);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<Metric> dmt((const Metric*)metric, n); DevBuf<Tetrad> dt(n);
    S5_BUFS_OK("tetrad_zamo", dmt.ok() && dt.ok());
    const Metric* pg = dmt.ptr; Tetrad* pt = dt.ptr;
    S5_RUN(n, "tetrad_zamo", [=] __device__(size_t i) { Tetrad t_; tetrad_zamo(pg[i], t_); pt[i] = t_; });
    S5_HIP(dt.to_host((Tetrad*)t));
    return SIM5GPU_OK;
}

int sim5gpu_tetrad_azimuthal(size_t n, const sim5gpu_metric* metric, const double* Omega, sim5gpu_tetrad* t)
{
    S5_NEED("tetrad_azimuthal", metric && Omega && t);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<Metric> dmt((const Metric*)metric, n); DevBuf<double> dO(Omega, n); DevBuf<Tetrad> dt(n);
    S5_BUFS_OK("tetrad_azimuthal", dmt.ok() && dO.ok() && dt.ok());
    const Metric* pg = dmt.ptr; const double* pO = dO.ptr; Tetrad* pt = dt.ptr;
    S5_RUN(n, "tetrad_azimuthal", [=] __device__(size_t i) { Tetrad t_; tetrad_azimuthal(pg[i], pO[i], t_); pt[i] = t_; });
    S5_HIP(dt.to_host((Tetrad*)t));
    return SIM5GPU_OK;
}

int sim5gpu_tetrad_surface(size_t n, const sim5gpu_metric* metric, const double* Omega, const double* V,
                           const double* dhdr, sim5gpu_tetrad* t)
{
    S5_NEED("tetrad_surface", metric && Omega && V && dhdr && t);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<Metric> dmt((const Metric*)metric, n); DevBuf<double> dO(Omega, n), dV(V, n), dh(dhdr, n); DevBuf<Tetrad> dt(n);
    S5_BUFS_OK("tetrad_surface", dmt.ok() && dO.ok() && dV.ok() && dh.ok() && dt.ok());
    const Metric* pg = dmt.ptr; const double *pO = dO.ptr, *pV = dV.ptr, *ph = dh.ptr; Tetrad* pt = dt.ptr;
    S5_RUN(n, "tetrad_surface", [=] __device__(size_t i) { Tetrad t_; tetrad_surface(pg[i], pO[i], pV[i], ph[i], t_); pt[i] = t_; });
    S5_HIP(dt.to_host((Tetrad*)t));
    return SIM5GPU_OK;
}

#define S5_FRAME_FN(NAME, DEVFN)                                                               \
int NAME(size_t n, const double* vin, double* vout, const sim5gpu_tetrad* t)                   \
{                                                                                              \
    S5_NEED(#NAME, vin && vout && t);                                                          \
    if (n == 0) return SIM5GPU_OK;                                                             \
    S5_DEVICE_OR_FAIL();                                                                       \
    DevBuf<double> di(vin, 4 * n), dout(4 * n); DevBuf<Tetrad> dt((const Tetrad*)t, n);        \
    S5_BUFS_OK(#NAME, di.ok() && dout.ok() && dt.ok());                                        \
    const double* pi = di.ptr; double* po = dout.ptr; const Tetrad* pt = dt.ptr;               \
    S5_RUN(n, #NAME, [=] __device__(size_t i) {                                                \
        const double v[4] = { pi[4 * i], pi[4 * i + 1], pi[4 * i + 2], pi[4 * i + 3] };        \
        double w[4];                                                                           \
        DEVFN(v, w, pt[i]);                                                                    \
        po[4 * i] = w[0]; po[4 * i + 1] = w[1]; po[4 * i + 2] = w[2]; po[4 * i + 3] = w[3];    \
    });                                                                                        \
    S5_HIP(dout.to_host(vout));                                                                \
    return SIM5GPU_OK;                                                                         \
}
S5_FRAME_FN(sim5gpu_bl2on, bl2on)
S5_FRAME_FN(sim5gpu_on2bl, on2bl)
#undef S5_FRAME_FN

// ------------------------------------------------------------------------------------------
// elliptic functions
// ------------------------------------------------------------------------------------------
int sim5gpu_elliptic(int which, size_t n, const double* x, const double* y, const double* z,
                     const double* w, double* out)
{
    S5_NEED("elliptic", x && out);
    if (which < 0 || which > 11) { snprintf(g_err, sizeof g_err, "elliptic: unknown selector %d", which); return SIM5GPU_E_ARG; }
    const bool need_y = (which != 1), need_z = (which == 0 || which == 8 || which == 10), need_w = (which == 10);
    S5_NEED("elliptic", (!need_y || y) && (!need_z || z) && (!need_w || w));
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dx(x, n), dy(need_y ? y : nullptr, need_y ? n : 0), dz(need_z ? z : nullptr, need_z ? n : 0),
        dw(need_w ? w : nullptr, need_w ? n : 0), dout(n);
    S5_BUFS_OK("elliptic", dx.ok() && dy.ok() && dz.ok() && dw.ok() && dout.ok());
    const double *px = dx.ptr, *py = dy.ptr, *pz = dz.ptr, *pw = dw.ptr; double* po = dout.ptr;
    S5_RUN(n, "elliptic", [=] __device__(size_t i) {
        double v = 0.0;
        switch (which) {                         // wave-uniform selector
        case 0: v = carlson_rf(px[i], py[i], pz[i]); break;
        case 1: v = ell_K(px[i]); break;
        case 2: v = inv_sn(px[i], py[i]); break;
        case 3: v = inv_cn(px[i], py[i]); break;
        case 4: v = inv_tn(px[i], py[i]); break;
        case 5: v = jac_sn(px[i], py[i]); break;
        case 6: v = jac_cn(px[i], py[i]); break;
        case 7: v = jac_dn(px[i], py[i]); break;
        case 8: v = carlson_rd(px[i], py[i], pz[i]); break;
        case 9: v = carlson_rc(px[i], py[i]); break;
        case 10: v = carlson_rj(px[i], py[i], pz[i], pw[i]); break;
        case 11: v = ell_F_sin(px[i], py[i]); break;
        }
        po[i] = v;
    });
    S5_HIP(dout.to_host(out));
    return SIM5GPU_OK;
}

// Legendre / Byrd & Friedman integrals behind position_azm and timedelay: selector in the order of the
// reference's definitions (see include/sim5gpu.h); args is nargs x n, one row per argument
int sim5gpu_integral(int which, size_t n, int nargs, const double* args, double* out)
{
    static const int need[27] = { 2, 2, 2, 3, 2, 2, 4, 4, 3, 3, 3, 3, 5, 4, 5, 5, 6, 5, 5, 4, 6, 6, 7, 6, 3, 3, 4 };
    S5_NEED("integral", args && out);
    if (which < 0 || which > 26) { snprintf(g_err, sizeof g_err, "integral: unknown selector %d", which); return SIM5GPU_E_ARG; }
    if (nargs != need[which]) { snprintf(g_err, sizeof g_err, "integral: selector %d takes %d arguments, got %d", which, need[which], nargs); return SIM5GPU_E_ARG; }
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> da(args, (size_t)nargs * n), dout(n);
    S5_BUFS_OK("integral", da.ok() && dout.ok());
    const double* pa = da.ptr; double* po = dout.ptr;
    S5_RUN(n, "integral", [=] __device__(size_t i) {
        double v[7] = { 0, 0, 0, 0, 0, 0, 0 };
        for (int k = 0; k < nargs; ++k) v[k] = pa[(size_t)k * n + i];
        double r = 0.0;
        switch (which) {                         // wave-uniform selector
        case 0: r = ell_F_cos(v[0], v[1]); break;
        case 1: r = ell_E_cos(v[0], v[1]); break;
        case 2: r = ell_Pi_complete(v[0], v[1]); break;
        case 3: r = ell_Pi_cos(v[0], v[1], v[2]); break;
        case 4: r = int_C2(v[0], v[1]); break;
        case 5: r = int_C2_cos(v[0], v[1]); break;
        case 6: r = int_Z1(v[0], v[1], v[2], v[3]); break;
        case 7: r = int_Z2(v[0], v[1], v[2], v[3]); break;
        case 8: r = int_Rm1(v[0], v[1], v[2]); break;
        case 9: r = int_Rm2(v[0], v[1], v[2]); break;
        case 10: r = int_R1(v[0], v[1], v[2]); break;
        case 11: r = int_R2(v[0], v[1], v[2]); break;
        case 12: r = R_r0_re(v[0], v[1], v[2], v[3], v[4]); break;
        case 13: r = R_r0_re_inf(v[0], v[1], v[2], v[3]); break;
        case 14: r = R_r1_re(v[0], v[1], v[2], v[3], v[4]); break;
        case 15: r = R_r2_re(v[0], v[1], v[2], v[3], v[4]); break;
        case 16: r = R_rp_re(v[0], v[1], v[2], v[3], v[4], v[5], false); break;
        case 17: r = R_rp_re(v[0], v[1], v[2], v[3], v[4], 0.0, true); break;
        case 18: r = R_r0_cc(v[0], v[1], v[2], v[3], v[4]); break;
        case 19: r = R_r0_cc_inf(v[0], v[1], v[2], v[3]); break;
        case 20: r = R_r1_cc(v[0], v[1], v[2], v[3], v[4], v[5]); break;
        case 21: r = R_r2_cc(v[0], v[1], v[2], v[3], v[4], v[5]); break;
        case 22: r = R_rp_cc2(v[0], v[1], v[2], v[3], v[4], v[5], v[6], false); break;
        case 23: r = R_rp_cc2(v[0], v[1], v[2], v[3], v[4], v[5], 0.0, true); break;
        case 24: r = T_m0(v[0], v[1], v[2]); break;
        case 25: r = T_m2(v[0], v[1], v[2]); break;
        case 26: r = T_mp(v[0], v[1], v[2], v[3]); break;
        }
        po[i] = r;
    });
    S5_HIP(dout.to_host(out));
    return SIM5GPU_OK;
}

int sim5gpu_geodesic_position_azm(size_t n, const sim5gpu_geodesic* g, const double* r, const double* m,
                                  const double* P, double* phi)
{
    S5_NEED("geodesic_position_azm", g && r && m && P && phi);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<Geod> dg((const Geod*)g, n); DevBuf<double> dr(r, n), dm(m, n), dP(P, n), dout(n);
    S5_BUFS_OK("geodesic_position_azm", dg.ok() && dr.ok() && dm.ok() && dP.ok() && dout.ok());
    const Geod* pg = dg.ptr; const double *pr = dr.ptr, *pm = dm.ptr, *pP = dP.ptr; double* po = dout.ptr;
    S5_RUN(n, "geodesic_position_azm", [=] __device__(size_t i) { po[i] = position_azm(pg[i], pr[i], pm[i], pP[i]); });
    S5_HIP(dout.to_host(phi));
    return SIM5GPU_OK;
}

int sim5gpu_geodesic_timedelay(size_t n, const sim5gpu_geodesic* g, const double* P1, const double* r1,
                               const double* m1, const double* P2, const double* r2, const double* m2, double* dt)
{
    S5_NEED("geodesic_timedelay", g && P1 && r1 && m1 && P2 && r2 && m2 && dt);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<Geod> dg((const Geod*)g, n); DevBuf<double> a1(P1, n), b1(r1, n), c1(m1, n), a2(P2, n), b2(r2, n), c2(m2, n), dout(n);
    S5_BUFS_OK("geodesic_timedelay", dg.ok() && a1.ok() && b1.ok() && c1.ok() && a2.ok() && b2.ok() && c2.ok() && dout.ok());
    const Geod* pg = dg.ptr; const double *q1 = a1.ptr, *s1 = b1.ptr, *t1 = c1.ptr, *q2 = a2.ptr, *s2 = b2.ptr, *t2 = c2.ptr;
    double* po = dout.ptr;
    S5_RUN(n, "geodesic_timedelay", [=] __device__(size_t i) { po[i] = timedelay(pg[i], q1[i], s1[i], t1[i], q2[i], s2[i], t2[i]); });
    S5_HIP(dout.to_host(dt));
    return SIM5GPU_OK;
}

// ------------------------------------------------------------------------------------------
// thin disk
// ------------------------------------------------------------------------------------------
int sim5gpu_disk_nt_flux(size_t n, const double* r, double* flux)
{
    S5_NEED("disk_nt_flux", r && flux);
    if (!g_disk.ready) { snprintf(g_err, sizeof g_err, "disk_nt_setup has not been called"); return SIM5GPU_E_NOT_SETUP; }
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dr(r, n), df(n);
    S5_BUFS_OK("disk_nt_flux", dr.ok() && df.ok());
    const double* pr = dr.ptr; double* pf = df.ptr; const DiskConsts d = g_disk;
    S5_RUN(n, "disk_nt_flux", [=] __device__(size_t i) { pf[i] = disk_flux(d, pr[i]); });
    S5_HIP(df.to_host(flux));
    return SIM5GPU_OK;
}

int sim5gpu_disk_nt_sigma(size_t n, const double* r, double* sigma)
{
    S5_NEED("disk_nt_sigma", r && sigma);
    if (!g_disk.ready) { snprintf(g_err, sizeof g_err, "disk_nt_setup has not been called"); return SIM5GPU_E_NOT_SETUP; }
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dr(r, n), ds(n);
    S5_BUFS_OK("disk_nt_sigma", dr.ok() && ds.ok());
    const double* pr = dr.ptr; double* ps = ds.ptr; const DiskConsts d = g_disk;
    S5_RUN(n, "disk_nt_sigma", [=] __device__(size_t i) { ps[i] = disk_sigma(d, pr[i]); });
    S5_HIP(ds.to_host(sigma));
    return SIM5GPU_OK;
}

int sim5gpu_disk_nt_ell(size_t n, const double* r, double* ell)
{
    S5_NEED("disk_nt_ell", r && ell);
    if (!g_disk.ready) { snprintf(g_err, sizeof g_err, "disk_nt_setup has not been called"); return SIM5GPU_E_NOT_SETUP; }
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dr(r, n), dl(n);
    S5_BUFS_OK("disk_nt_ell", dr.ok() && dl.ok());
    const double* pr = dr.ptr; double* pl = dl.ptr; const DiskConsts d = g_disk;
    S5_RUN(n, "disk_nt_ell", [=] __device__(size_t i) { pl[i] = disk_ell(d, pr[i]); });
    S5_HIP(dl.to_host(ell));
    return SIM5GPU_OK;
}

// ------------------------------------------------------------------------------------------
// step-wise integrator
// ------------------------------------------------------------------------------------------
int sim5gpu_raytrace_prepare(size_t n, const double* bh_spin, const double* x, const double* k,
                             const double* precision, const int* options, sim5gpu_raytrace_data* rtd)
{
    S5_NEED("raytrace_prepare", bh_spin && x && k && precision && options && rtd);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> da(bh_spin, n), dx(x, 4 * n), dk(k, 4 * n), dp(precision, n); DevBuf<int> dopt(options, n);
    DevBuf<RayState> ds((const RayState*)rtd, n);
    S5_BUFS_OK("raytrace_prepare", da.ok() && dx.ok() && dk.ok() && dp.ok() && dopt.ok() && ds.ok());
    const double *pa = da.ptr, *px = dx.ptr, *pk = dk.ptr, *pp = dp.ptr; const int* po = dopt.ptr; RayState* ps = ds.ptr;
    S5_RUN(n, "raytrace_prepare", [=] __device__(size_t i) {
        RayState s = ps[i];
        const double xx[4] = { px[4 * i], px[4 * i + 1], px[4 * i + 2], px[4 * i + 3] };
        const double kk[4] = { pk[4 * i], pk[4 * i + 1], pk[4 * i + 2], pk[4 * i + 3] };
        raytrace_prepare(pa[i], xx, kk, pp[i], po[i], s);
        ps[i] = s;
    });
    S5_HIP(ds.to_host((RayState*)rtd));
    return SIM5GPU_OK;
}

int sim5gpu_raytrace(size_t n, double* x, double* k, double* step, sim5gpu_raytrace_data* rtd, int nsteps)
{
    S5_NEED("raytrace", x && k && step && rtd);
    if (nsteps < 1) { snprintf(g_err, sizeof g_err, "raytrace: nsteps must be >= 1"); return SIM5GPU_E_ARG; }
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dx(x, 4 * n), dk(k, 4 * n), dst(step, n); DevBuf<RayState> ds((const RayState*)rtd, n);
    S5_BUFS_OK("raytrace", dx.ok() && dk.ok() && dst.ok() && ds.ok());
    double *px = dx.ptr, *pk = dk.ptr, *pst = dst.ptr; RayState* ps = ds.ptr;
    S5_RUN(n, "raytrace", [=] __device__(size_t i) {
        RayState s = ps[i];
        double xx[4] = { px[4 * i], px[4 * i + 1], px[4 * i + 2], px[4 * i + 3] };
        double kk[4] = { pk[4 * i], pk[4 * i + 1], pk[4 * i + 2], pk[4 * i + 3] };
        const double cap = pst[i];
        double taken = cap;
        for (int it = 0; it < nsteps; ++it) { taken = cap; raytrace_step(xx, kk, taken, s); }
        ps[i] = s; pst[i] = taken;
        px[4 * i] = xx[0]; px[4 * i + 1] = xx[1]; px[4 * i + 2] = xx[2]; px[4 * i + 3] = xx[3];
        pk[4 * i] = kk[0]; pk[4 * i + 1] = kk[1]; pk[4 * i + 2] = kk[2]; pk[4 * i + 3] = kk[3];
    });
    S5_HIP(dx.to_host(x)); S5_HIP(dk.to_host(k)); S5_HIP(dst.to_host(step)); S5_HIP(ds.to_host((RayState*)rtd));
    return SIM5GPU_OK;
}

// ONE ray, `nsteps` consecutive raytrace() calls with the same cap re-applied at each, every intermediate result kept: record j
// holds what the j-th call leaves in x, k, *step and *rtd.  For callers that make the calls one by one (the scalar API of
// sim5_amd/host/sim5lib.c: ref README.md:184-193, src/sim5unittests.c:116-127) -- one launch per `nsteps` calls instead of one per call.
int sim5gpu_raytrace_record(const double* x, const double* k, double step_cap, const sim5gpu_raytrace_data* rtd, int nsteps,
                            sim5gpu_raytrace_step* records)
{
    S5_NEED("raytrace_record", x && k && rtd && records);
    if (nsteps < 1 || nsteps > 4096) { snprintf(g_err, sizeof g_err, "raytrace_record: nsteps must be in 1 .. 4096"); return SIM5GPU_E_ARG; }
    static_assert(sizeof(sim5gpu_raytrace_step) == 216 && offsetof(sim5gpu_raytrace_step, rtd) == 72, "raytrace_step layout");
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dx(x, 4), dk(k, 4); DevBuf<RayState> ds((const RayState*)rtd, 1); DevBuf<sim5gpu_raytrace_step> dr((size_t)nsteps);
    S5_BUFS_OK("raytrace_record", dx.ok() && dk.ok() && ds.ok() && dr.ok());
    const double *px = dx.ptr, *pk = dk.ptr; const RayState* ps = ds.ptr; sim5gpu_raytrace_step* pr = dr.ptr;
    S5_RUN(1, "raytrace_record", [=] __device__(size_t) {
        RayState s = ps[0];
        double xx[4] = { px[0], px[1], px[2], px[3] };
        double kk[4] = { pk[0], pk[1], pk[2], pk[3] };
        for (int it = 0; it < nsteps; ++it) {
            double taken = step_cap;
            raytrace_step(xx, kk, taken, s);
            sim5gpu_raytrace_step& o = pr[it];
            o.x[0] = xx[0]; o.x[1] = xx[1]; o.x[2] = xx[2]; o.x[3] = xx[3];
            o.k[0] = kk[0]; o.k[1] = kk[1]; o.k[2] = kk[2]; o.k[3] = kk[3];
            o.step = taken;
            *(RayState*)&o.rtd = s;
        }
    });
    S5_HIP(dr.to_host(records));
    return SIM5GPU_OK;
}

int sim5gpu_raytrace_error(size_t n, const double* x, const double* k, const sim5gpu_raytrace_data* rtd, double* err)
{
    S5_NEED("raytrace_error", x && k && rtd && err);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dx(x, 4 * n), dk(k, 4 * n), de(n); DevBuf<RayState> ds((const RayState*)rtd, n);
    S5_BUFS_OK("raytrace_error", dx.ok() && dk.ok() && de.ok() && ds.ok());
    const double *px = dx.ptr, *pk = dk.ptr; double* pe = de.ptr; const RayState* ps = ds.ptr;
    S5_RUN(n, "raytrace_error", [=] __device__(size_t i) {
        const double xx[4] = { px[4 * i], px[4 * i + 1], px[4 * i + 2], px[4 * i + 3] };
        const double kk[4] = { pk[4 * i], pk[4 * i + 1], pk[4 * i + 2], pk[4 * i + 3] };
        pe[i] = raytrace_error(xx, kk, ps[i]);
    });
    S5_HIP(de.to_host(err));
    return SIM5GPU_OK;
}

// ------------------------------------------------------------------------------------------
// polarization and radiation
// ------------------------------------------------------------------------------------------
int sim5gpu_polarization_constant(size_t n, const double* k, const double* f, const sim5gpu_metric* metric, double* wp)
{
    S5_NEED("polarization_constant", k && f && metric && wp);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dk(k, 4 * n), df(f, 4 * n), dw(2 * n); DevBuf<Metric> dmt((const Metric*)metric, n);
    S5_BUFS_OK("polarization_constant", dk.ok() && df.ok() && dw.ok() && dmt.ok());
    const double *pk = dk.ptr, *pf = df.ptr; double* pw = dw.ptr; const Metric* pg = dmt.ptr;
    S5_RUN(n, "polarization_constant", [=] __device__(size_t i) {
        const double kk[4] = { pk[4 * i], pk[4 * i + 1], pk[4 * i + 2], pk[4 * i + 3] };
        const double ff[4] = { pf[4 * i], pf[4 * i + 1], pf[4 * i + 2], pf[4 * i + 3] };
        double w2[2];
        polarization_constant(kk, ff, pg[i], w2);
        pw[2 * i] = w2[0]; pw[2 * i + 1] = w2[1];
    });
    S5_HIP(dw.to_host(wp));
    return SIM5GPU_OK;
}

int sim5gpu_polarization_vector(size_t n, const double* k, const double* wp, const sim5gpu_metric* metric, double* f)
{
    S5_NEED("polarization_vector", k && wp && metric && f);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dk(k, 4 * n), dw(wp, 2 * n), df(4 * n); DevBuf<Metric> dmt((const Metric*)metric, n);
    S5_BUFS_OK("polarization_vector", dk.ok() && dw.ok() && df.ok() && dmt.ok());
    const double *pk = dk.ptr, *pw = dw.ptr; double* pf = df.ptr; const Metric* pg = dmt.ptr;
    S5_RUN(n, "polarization_vector", [=] __device__(size_t i) {
        const double kk[4] = { pk[4 * i], pk[4 * i + 1], pk[4 * i + 2], pk[4 * i + 3] };
        const double w2[2] = { pw[2 * i], pw[2 * i + 1] };
        double ff[4];
        polarization_vector(kk, w2, pg[i], ff);
        pf[4 * i] = ff[0]; pf[4 * i + 1] = ff[1]; pf[4 * i + 2] = ff[2]; pf[4 * i + 3] = ff[3];
    });
    S5_HIP(df.to_host(f));
    return SIM5GPU_OK;
}

int sim5gpu_polarization_constant_infinity(size_t n, const double* a, const double* alpha, const double* beta,
                                           const double* incl, double* wp)
{
    S5_NEED("polarization_constant_infinity", a && alpha && beta && incl && wp);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> da(a, n), dal(alpha, n), dbe(beta, n), di(incl, n), dw(2 * n);
    S5_BUFS_OK("polarization_constant_infinity", da.ok() && dal.ok() && dbe.ok() && di.ok() && dw.ok());
    const double *pa = da.ptr, *pal = dal.ptr, *pbe = dbe.ptr, *pi = di.ptr; double* pw = dw.ptr;
    S5_RUN(n, "polarization_constant_infinity", [=] __device__(size_t i) {
        double w2[2];
        polarization_constant_infinity(pa[i], pal[i], pbe[i], sin(pi[i]), w2);
        pw[2 * i] = w2[0]; pw[2 * i + 1] = w2[1];
    });
    S5_HIP(dw.to_host(wp));
    return SIM5GPU_OK;
}

int sim5gpu_polarization_angle_rotation(size_t n, const double* a, const double* inc, const double* alpha,
                                        const double* beta, const double* wp, double* angle)
{
    S5_NEED("polarization_angle_rotation", a && inc && alpha && beta && wp && angle);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> da(a, n), di(inc, n), dal(alpha, n), dbe(beta, n), dw(wp, 2 * n), dout(n);
    S5_BUFS_OK("polarization_angle_rotation", da.ok() && di.ok() && dal.ok() && dbe.ok() && dw.ok() && dout.ok());
    const double *pa = da.ptr, *pi = di.ptr, *pal = dal.ptr, *pbe = dbe.ptr, *pw = dw.ptr; double* po = dout.ptr;
    S5_RUN(n, "polarization_angle_rotation", [=] __device__(size_t i) {
        const double w2[2] = { pw[2 * i], pw[2 * i + 1] };
        po[i] = polarization_angle_rotation(pa[i], sin(pi[i]), pal[i], pbe[i], w2);
    });
    S5_HIP(dout.to_host(angle));
    return SIM5GPU_OK;
}

int sim5gpu_blackbody_Iv(size_t n, const double* T, const double* hardf, const double* cos_mu, const double* E, double* Iv)
{
    S5_NEED("blackbody_Iv", T && hardf && cos_mu && E && Iv);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dT(T, n), dh(hardf, n), dc(cos_mu, n), dE(E, n), dI(n);
    S5_BUFS_OK("blackbody_Iv", dT.ok() && dh.ok() && dc.ok() && dE.ok() && dI.ok());
    const double *pT = dT.ptr, *ph = dh.ptr, *pc = dc.ptr, *pE = dE.ptr; double* pI = dI.ptr;
    S5_RUN(n, "blackbody_Iv", [=] __device__(size_t i) { pI[i] = blackbody_Iv(pT[i], ph[i], pc[i], pE[i]); });
    S5_HIP(dI.to_host(Iv));
    return SIM5GPU_OK;
}

} // extern "C"
