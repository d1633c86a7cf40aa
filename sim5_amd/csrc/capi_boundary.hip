// capi_boundary.hip -- batch forms of the remaining public SIM5 prototypes of the headers SURVEY.md 8(b) cites
// (group (1b) of include/sim5gpu.h): metric helpers, Gamma, the vector helpers that do arithmetic, tetrad_general /
// tetrad_radial, epicyclic frequencies, four-velocities, Legendre integrals by angle / sine, black-body photon
// counts.  Same scheme as capi_batch.hip: validate -> upload the caller's host arrays -> one kernel, one lane per
// element, calling the device routines of s5_boundary.hpp -> download.
#include "capi_util.hpp"
#include "s5_polar.hpp"
#include "s5_boundary.hpp"

namespace s5 {

template <typename F>
static int run_elems(size_t n, F body, const char* what) { return run_batch(n, body, what); }

static int null_arg(const char* fn)
{
    snprintf(g_err, sizeof g_err, "%s: NULL pointer argument", fn);
    return SIM5GPU_E_ARG;
}

#define S5_NEED(fn, cond) do { if (!(cond)) return null_arg(fn); } while (0)
#define S5_DEVICE_OR_FAIL() do { if (!have_device()) return SIM5GPU_E_NO_DEVICE; } while (0)
#define S5_BUFS_OK(fn, cond) do { if (!(cond)) { snprintf(g_err, sizeof g_err, "%s: device allocation/copy failed", fn); return SIM5GPU_E_HIP; } } while (0)
#define S5_RUN(n, what, ...) do { int rc_ = run_elems(n, __VA_ARGS__, what); if (rc_) return rc_; } while (0)
#define S5_LOAD4(v, p, i) const double v[4] = { p[4 * (i)], p[4 * (i) + 1], p[4 * (i) + 2], p[4 * (i) + 3] }
#define S5_STORE4(p, i, v) do { p[4 * (i)] = v[0]; p[4 * (i) + 1] = v[1]; p[4 * (i) + 2] = v[2]; p[4 * (i) + 3] = v[3]; } while (0)

} // namespace s5

using namespace s5;

extern "C" {

// ------------------------------------------------------------------------------------------
// metrics and connection of Minkowski space, contravariant Kerr metric
// ------------------------------------------------------------------------------------------
#define S5_FLAT_METRIC_FN(NAME, DEVFN)                                                         \
int NAME(size_t n, const double* r, const double* m, sim5gpu_metric* metric)                   \
{                                                                                              \
    S5_NEED(#NAME, r && m && metric);                                                          \
    if (n == 0) return SIM5GPU_OK;                                                             \
    S5_DEVICE_OR_FAIL();                                                                       \
    DevBuf<double> dr(r, n), dm(m, n); DevBuf<Metric> dmt(n);                                  \
    S5_BUFS_OK(#NAME, dr.ok() && dm.ok() && dmt.ok());                                         \
    const double *pr = dr.ptr, *pm = dm.ptr; Metric* pg = dmt.ptr;                             \
    S5_RUN(n, #NAME, [=] __device__(size_t i) { Metric g; DEVFN(pr[i], pm[i], g); pg[i] = g; }); \
    S5_HIP(dmt.to_host((Metric*)metric));                                                      \
    return SIM5GPU_OK;                                                                         \
}
S5_FLAT_METRIC_FN(sim5gpu_flat_metric, flat_metric)
S5_FLAT_METRIC_FN(sim5gpu_flat_metric_contravariant, flat_metric_contravariant)
#undef S5_FLAT_METRIC_FN

int sim5gpu_kerr_metric_contravariant(size_t n, const double* a, const double* r, const double* m, sim5gpu_metric* metric)
{
    S5_NEED("kerr_metric_contravariant", a && r && m && metric);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> da(a, n), dr(r, n), dm(m, n); DevBuf<Metric> dmt(n);
    S5_BUFS_OK("kerr_metric_contravariant", da.ok() && dr.ok() && dm.ok() && dmt.ok());
    const double *pa = da.ptr, *pr = dr.ptr, *pm = dm.ptr; Metric* pg = dmt.ptr;
    S5_RUN(n, "kerr_metric_contravariant", [=] __device__(size_t i) { Metric g; kerr_metric_contravariant(pa[i], pr[i], pm[i], g); pg[i] = g; });
    S5_HIP(dmt.to_host((Metric*)metric));
    return SIM5GPU_OK;
}

// Kerr-Newman (charge Q): metric, contravariant metric, connection
#define S5_KN_METRIC_FN(NAME, DEVFN)                                                                         \
int NAME(size_t n, const double* a, const double* Q, const double* r, const double* m, sim5gpu_metric* metric) \
{                                                                                                            \
    S5_NEED(#NAME, a && Q && r && m && metric);                                                              \
    if (n == 0) return SIM5GPU_OK;                                                                           \
    S5_DEVICE_OR_FAIL();                                                                                     \
    DevBuf<double> da(a, n), dq(Q, n), dr(r, n), dm(m, n); DevBuf<Metric> dmt(n);                            \
    S5_BUFS_OK(#NAME, da.ok() && dq.ok() && dr.ok() && dm.ok() && dmt.ok());                                 \
    const double *pa = da.ptr, *pq = dq.ptr, *pr = dr.ptr, *pm = dm.ptr; Metric* pg = dmt.ptr;              \
    S5_RUN(n, #NAME, [=] __device__(size_t i) { Metric g; DEVFN(pa[i], pq[i], pr[i], pm[i], g); pg[i] = g; }); \
    S5_HIP(dmt.to_host((Metric*)metric));                                                                    \
    return SIM5GPU_OK;                                                                                       \
}
S5_KN_METRIC_FN(sim5gpu_kerr_newman_metric, kerr_newman_metric)
S5_KN_METRIC_FN(sim5gpu_kerr_newman_metric_contravariant, kerr_newman_metric_contravariant)
#undef S5_KN_METRIC_FN

int sim5gpu_kerr_newman_connection(size_t n, const double* a, const double* Q, const double* r, const double* m, double* G)
{
    S5_NEED("kerr_newman_connection", a && Q && r && m && G);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> da(a, n), dq(Q, n), dr(r, n), dm(m, n), dG(64 * n);
    S5_BUFS_OK("kerr_newman_connection", da.ok() && dq.ok() && dr.ok() && dm.ok() && dG.ok());
    const double *pa = da.ptr, *pq = dq.ptr, *pr = dr.ptr, *pm = dm.ptr; double* pG = dG.ptr;
    S5_RUN(n, "kerr_newman_connection", [=] __device__(size_t i) { Conn c; kerr_newman_connection(pa[i], pq[i], pr[i], pm[i], c); conn_to_dense(c, pG + 64 * i); });
    S5_HIP(dG.to_host(G));
    return SIM5GPU_OK;
}

int sim5gpu_flat_connection(size_t n, const double* r, const double* m, double* G)
{
    S5_NEED("flat_connection", r && m && G);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dr(r, n), dm(m, n), dG(64 * n);
    S5_BUFS_OK("flat_connection", dr.ok() && dm.ok() && dG.ok());
    const double *pr = dr.ptr, *pm = dm.ptr; double* pG = dG.ptr;
    S5_RUN(n, "flat_connection", [=] __device__(size_t i) { Conn c; flat_connection(pr[i], pm[i], c); conn_to_dense(c, pG + 64 * i); });
    S5_HIP(dG.to_host(G));
    return SIM5GPU_OK;
}

int sim5gpu_Gamma(size_t n, const double* G, const double* U, const double* V, double* result)
{
    S5_NEED("Gamma", G && U && V && result);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dG(G, 64 * n), dU(U, 4 * n), dV(V, 4 * n), dout(4 * n);
    S5_BUFS_OK("Gamma", dG.ok() && dU.ok() && dV.ok() && dout.ok());
    const double *pG = dG.ptr, *pU = dU.ptr, *pV = dV.ptr; double* po = dout.ptr;
    S5_RUN(n, "Gamma", [=] __device__(size_t i) {
        S5_LOAD4(u, pU, i); S5_LOAD4(v, pV, i);
        double w[4];
        gamma_dense(pG + 64 * i, u, v, w);
        S5_STORE4(po, i, w);
    });
    S5_HIP(dout.to_host(result));
    return SIM5GPU_OK;
}

// ------------------------------------------------------------------------------------------
// vector helpers that do arithmetic (metric == NULL: Minkowski, as in the reference)
// ------------------------------------------------------------------------------------------
int sim5gpu_vector_covariant(size_t n, const double* v1, double* v2, const sim5gpu_metric* metric)
{
    S5_NEED("vector_covariant", v1 && v2);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> d1(v1, 4 * n), d2(4 * n); DevBuf<Metric> dmt((const Metric*)metric, metric ? n : 0);
    S5_BUFS_OK("vector_covariant", d1.ok() && d2.ok() && dmt.ok());
    const double* p1 = d1.ptr; double* p2 = d2.ptr; const Metric* pg = dmt.ptr;
    S5_RUN(n, "vector_covariant", [=] __device__(size_t i) {
        S5_LOAD4(u, p1, i);
        double w[4];
        vector_covariant(u, w, pg ? &pg[i] : nullptr);
        S5_STORE4(p2, i, w);
    });
    S5_HIP(d2.to_host(v2));
    return SIM5GPU_OK;
}

int sim5gpu_vector_norm(size_t n, const double* v, const sim5gpu_metric* metric, double* out)
{
    S5_NEED("vector_norm", v && out);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dv(v, 4 * n), dout(n); DevBuf<Metric> dmt((const Metric*)metric, metric ? n : 0);
    S5_BUFS_OK("vector_norm", dv.ok() && dout.ok() && dmt.ok());
    const double* pv = dv.ptr; double* po = dout.ptr; const Metric* pg = dmt.ptr;
    S5_RUN(n, "vector_norm", [=] __device__(size_t i) { S5_LOAD4(u, pv, i); po[i] = vector_norm(u, pg ? &pg[i] : nullptr); });
    S5_HIP(dout.to_host(out));
    return SIM5GPU_OK;
}

int sim5gpu_vector_3norm(size_t n, const double* v, double* out)
{
    S5_NEED("vector_3norm", v && out);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dv(v, 4 * n), dout(n);
    S5_BUFS_OK("vector_3norm", dv.ok() && dout.ok());
    const double* pv = dv.ptr; double* po = dout.ptr;
    S5_RUN(n, "vector_3norm", [=] __device__(size_t i) { S5_LOAD4(u, pv, i); po[i] = vector_3norm(u); });
    S5_HIP(dout.to_host(out));
    return SIM5GPU_OK;
}

int sim5gpu_vector_norm_to_null(size_t n, double* v, const double* V0, const sim5gpu_metric* metric)
{
    S5_NEED("vector_norm_to_null", v && V0);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dv(v, 4 * n), d0(V0, n); DevBuf<Metric> dmt((const Metric*)metric, metric ? n : 0);
    S5_BUFS_OK("vector_norm_to_null", dv.ok() && d0.ok() && dmt.ok());
    double* pv = dv.ptr; const double* p0 = d0.ptr; const Metric* pg = dmt.ptr;
    S5_RUN(n, "vector_norm_to_null", [=] __device__(size_t i) {
        double u[4] = { pv[4 * i], pv[4 * i + 1], pv[4 * i + 2], pv[4 * i + 3] };
        vector_norm_to_null(u, p0[i], pg ? &pg[i] : nullptr);
        S5_STORE4(pv, i, u);
    });
    S5_HIP(dv.to_host(v));
    return SIM5GPU_OK;
}

// ------------------------------------------------------------------------------------------
// tetrads of a general and of a radially moving observer
// ------------------------------------------------------------------------------------------
int sim5gpu_tetrad_general(size_t n, const sim5gpu_metric* metric, const double* U, sim5gpu_tetrad* t)
{
    S5_NEED("tetrad_general", metric && U && t);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<Metric> dmt((const Metric*)metric, n); DevBuf<double> dU(U, 4 * n); DevBuf<Tetrad> dt(n);
    S5_BUFS_OK("tetrad_general", dmt.ok() && dU.ok() && dt.ok());
    const Metric* pg = dmt.ptr; const double* pU = dU.ptr; Tetrad* pt = dt.ptr;
    S5_RUN(n, "tetrad_general", [=] __device__(size_t i) { S5_LOAD4(u, pU, i); Tetrad t_; tetrad_general(pg[i], u, t_); pt[i] = t_; });
    S5_HIP(dt.to_host((Tetrad*)t));
    return SIM5GPU_OK;
}

int sim5gpu_tetrad_radial(size_t n, const sim5gpu_metric* metric, const double* v_r, sim5gpu_tetrad* t)
{
    S5_NEED("tetrad_radial", metric && v_r && t);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<Metric> dmt((const Metric*)metric, n); DevBuf<double> dv(v_r, n); DevBuf<Tetrad> dt(n);
    S5_BUFS_OK("tetrad_radial", dmt.ok() && dv.ok() && dt.ok());
    const Metric* pg = dmt.ptr; const double* pv = dv.ptr; Tetrad* pt = dt.ptr;
    S5_RUN(n, "tetrad_radial", [=] __device__(size_t i) { Tetrad t_; tetrad_radial(pg[i], pv[i], t_); pt[i] = t_; });
    S5_HIP(dt.to_host((Tetrad*)t));
    return SIM5GPU_OK;
}

// ------------------------------------------------------------------------------------------
// orbital frequencies
// ------------------------------------------------------------------------------------------
#define S5_RA_FN(NAME, DEVFN)                                                                  \
int NAME(size_t n, const double* r, const double* a, double* out)                              \
{                                                                                              \
    S5_NEED(#NAME, r && a && out);                                                             \
    if (n == 0) return SIM5GPU_OK;                                                             \
    S5_DEVICE_OR_FAIL();                                                                       \
    DevBuf<double> dr(r, n), da(a, n), dout(n);                                                \
    S5_BUFS_OK(#NAME, dr.ok() && da.ok() && dout.ok());                                        \
    const double *pr = dr.ptr, *pa = da.ptr; double* po = dout.ptr;                            \
    S5_RUN(n, #NAME, [=] __device__(size_t i) { po[i] = DEVFN(pr[i], pa[i]); });               \
    S5_HIP(dout.to_host(out));                                                                 \
    return SIM5GPU_OK;                                                                         \
}
S5_RA_FN(sim5gpu_omega_r, omega_r)
S5_RA_FN(sim5gpu_omega_z, omega_z)
#undef S5_RA_FN

int sim5gpu_ell_from_Omega(size_t n, const double* Omega, const sim5gpu_metric* metric, double* ell)
{
    S5_NEED("ell_from_Omega", Omega && metric && ell);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dO(Omega, n), dout(n); DevBuf<Metric> dmt((const Metric*)metric, n);
    S5_BUFS_OK("ell_from_Omega", dO.ok() && dout.ok() && dmt.ok());
    const double* pO = dO.ptr; double* po = dout.ptr; const Metric* pg = dmt.ptr;
    S5_RUN(n, "ell_from_Omega", [=] __device__(size_t i) { po[i] = ell_from_omega(pO[i], pg[i]); });
    S5_HIP(dout.to_host(ell));
    return SIM5GPU_OK;
}

// ------------------------------------------------------------------------------------------
// four-velocities
// ------------------------------------------------------------------------------------------
int sim5gpu_fourvelocity_zamo(size_t n, const sim5gpu_metric* metric, double* U)
{
    S5_NEED("fourvelocity_zamo", metric && U);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<Metric> dmt((const Metric*)metric, n); DevBuf<double> dU(4 * n);
    S5_BUFS_OK("fourvelocity_zamo", dmt.ok() && dU.ok());
    const Metric* pg = dmt.ptr; double* pU = dU.ptr;
    S5_RUN(n, "fourvelocity_zamo", [=] __device__(size_t i) { double u[4]; fourvelocity_zamo(pg[i], u); S5_STORE4(pU, i, u); });
    S5_HIP(dU.to_host(U));
    return SIM5GPU_OK;
}

#define S5_FOURVEL1_FN(NAME, DEVFN)                                                            \
int NAME(size_t n, const double* x, const sim5gpu_metric* metric, double* U)                   \
{                                                                                              \
    S5_NEED(#NAME, x && metric && U);                                                          \
    if (n == 0) return SIM5GPU_OK;                                                             \
    S5_DEVICE_OR_FAIL();                                                                       \
    DevBuf<double> dx(x, n), dU(4 * n); DevBuf<Metric> dmt((const Metric*)metric, n);          \
    S5_BUFS_OK(#NAME, dx.ok() && dU.ok() && dmt.ok());                                         \
    const double* px = dx.ptr; const Metric* pg = dmt.ptr; double* pU = dU.ptr;                \
    S5_RUN(n, #NAME, [=] __device__(size_t i) { double u[4]; DEVFN(px[i], pg[i], u); S5_STORE4(pU, i, u); }); \
    S5_HIP(dU.to_host(U));                                                                     \
    return SIM5GPU_OK;                                                                         \
}
S5_FOURVEL1_FN(sim5gpu_fourvelocity_azimuthal, fourvelocity_azimuthal)
S5_FOURVEL1_FN(sim5gpu_fourvelocity_radial, fourvelocity_radial)
#undef S5_FOURVEL1_FN

int sim5gpu_fourvelocity_norm(size_t n, const double* U1, const double* U2, const double* U3, const sim5gpu_metric* metric, double* out)
{
    S5_NEED("fourvelocity_norm", U1 && U2 && U3 && metric && out);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> d1(U1, n), d2(U2, n), d3(U3, n), dout(n); DevBuf<Metric> dmt((const Metric*)metric, n);
    S5_BUFS_OK("fourvelocity_norm", d1.ok() && d2.ok() && d3.ok() && dout.ok() && dmt.ok());
    const double *p1 = d1.ptr, *p2 = d2.ptr, *p3 = d3.ptr; const Metric* pg = dmt.ptr; double* po = dout.ptr;
    S5_RUN(n, "fourvelocity_norm", [=] __device__(size_t i) { po[i] = fourvelocity_norm(p1[i], p2[i], p3[i], pg[i]); });
    S5_HIP(dout.to_host(out));
    return SIM5GPU_OK;
}

int sim5gpu_fourvelocity(size_t n, const double* U1, const double* U2, const double* U3, const sim5gpu_metric* metric, double* U)
{
    S5_NEED("fourvelocity", U1 && U2 && U3 && metric && U);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> d1(U1, n), d2(U2, n), d3(U3, n), dU(4 * n); DevBuf<Metric> dmt((const Metric*)metric, n);
    S5_BUFS_OK("fourvelocity", d1.ok() && d2.ok() && d3.ok() && dU.ok() && dmt.ok());
    const double *p1 = d1.ptr, *p2 = d2.ptr, *p3 = d3.ptr; const Metric* pg = dmt.ptr; double* pU = dU.ptr;
    S5_RUN(n, "fourvelocity", [=] __device__(size_t i) { double u[4]; fourvelocity(p1[i], p2[i], p3[i], pg[i], u); S5_STORE4(pU, i, u); });
    S5_HIP(dU.to_host(U));
    return SIM5GPU_OK;
}

// ------------------------------------------------------------------------------------------
// geodesics: sign of k^theta
// ------------------------------------------------------------------------------------------
int sim5gpu_geodesic_position_pol_sign_k_theta(size_t n, const sim5gpu_geodesic* g, const double* P, double* sign)
{
    S5_NEED("geodesic_position_pol_sign_k_theta", g && P && sign);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<Geod> dg((const Geod*)g, n); DevBuf<double> dP(P, n), dout(n);
    S5_BUFS_OK("geodesic_position_pol_sign_k_theta", dg.ok() && dP.ok() && dout.ok());
    const Geod* pg = dg.ptr; const double* pP = dP.ptr; double* po = dout.ptr;
    S5_RUN(n, "geodesic_position_pol_sign_k_theta", [=] __device__(size_t i) { po[i] = position_pol_sign_k_theta(pg[i], pP[i]); });
    S5_HIP(dout.to_host(sign));
    return SIM5GPU_OK;
}

// ------------------------------------------------------------------------------------------
// Legendre integrals by angle / by sine
// ------------------------------------------------------------------------------------------
int sim5gpu_legendre(int which, size_t n, const double* x, const double* nn, const double* m, double* out)
{
    S5_NEED("legendre", x && m && out);
    if (which < 0 || which > 3) { snprintf(g_err, sizeof g_err, "legendre: unknown selector %d", which); return SIM5GPU_E_ARG; }
    const bool need_n = (which >= 2);
    S5_NEED("legendre", !need_n || nn);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    const size_t nout = (which == 3) ? 2 * n : n;
    DevBuf<double> dx(x, n), dn(need_n ? nn : nullptr, need_n ? n : 0), dm(m, n), dout(nout);
    S5_BUFS_OK("legendre", dx.ok() && dn.ok() && dm.ok() && dout.ok());
    const double *px = dx.ptr, *pn = dn.ptr, *pm = dm.ptr; double* po = dout.ptr;
    S5_RUN(n, "legendre", [=] __device__(size_t i) {
        switch (which) {                         // wave-uniform selector
        case 0: po[i] = ell_F(px[i], pm[i]); break;
        case 1: po[i] = ell_E_sin(px[i], pm[i]); break;
        case 2: po[i] = ell_Pi_sin(px[i], pn[i], pm[i]); break;
        case 3: { double w[2]; ell_Pi(px[i], pn[i], pm[i], w); po[2 * i] = w[0]; po[2 * i + 1] = w[1]; } break;
        }
    });
    S5_HIP(dout.to_host(out));
    return SIM5GPU_OK;
}

// ------------------------------------------------------------------------------------------
// black body: spectrum form and photon counts
// ------------------------------------------------------------------------------------------
int sim5gpu_blackbody(double T, double hardf, double cos_mu, size_t n_energies, const double* E, double* Iv)
{
    S5_NEED("blackbody", E && Iv);
    if (n_energies == 0 || T <= 0.0) return SIM5GPU_OK;          // T <= 0: Iv[] is left untouched (ref src/sim5radiation.c:70)
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dE(E, n_energies), dI(n_energies);
    S5_BUFS_OK("blackbody", dE.ok() && dI.ok());
    const double* pE = dE.ptr; double* pI = dI.ptr;
    S5_RUN(n_energies, "blackbody", [=] __device__(size_t i) {
        double BB1, BB2;
        blackbody_factors(T, hardf, cos_mu, BB1, BB2);
        pI[i] = BB1 * (pE[i] * pE[i] * pE[i]) / expm1(BB2 * pE[i]);
    });
    S5_HIP(dI.to_host(Iv));
    return SIM5GPU_OK;
}

int sim5gpu_blackbody_photons(size_t n, const double* T, const double* hardf, const double* cos_mu, const double* E, double* out)
{
    S5_NEED("blackbody_photons", T && hardf && cos_mu && E && out);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dT(T, n), dh(hardf, n), dc(cos_mu, n), dE(E, n), dout(n);
    S5_BUFS_OK("blackbody_photons", dT.ok() && dh.ok() && dc.ok() && dE.ok() && dout.ok());
    const double *pT = dT.ptr, *ph = dh.ptr, *pc = dc.ptr, *pE = dE.ptr; double* po = dout.ptr;
    S5_RUN(n, "blackbody_photons", [=] __device__(size_t i) { po[i] = blackbody_photons(pT[i], ph[i], pc[i], pE[i]); });
    S5_HIP(dout.to_host(out));
    return SIM5GPU_OK;
}

int sim5gpu_blackbody_photons_total(size_t n, const double* T, const double* hardf, double* out)
{
    S5_NEED("blackbody_photons_total", T && hardf && out);
    if (n == 0) return SIM5GPU_OK;
    S5_DEVICE_OR_FAIL();
    DevBuf<double> dT(T, n), dh(hardf, n), dout(n);
    S5_BUFS_OK("blackbody_photons_total", dT.ok() && dh.ok() && dout.ok());
    const double *pT = dT.ptr, *ph = dh.ptr; double* po = dout.ptr;
    S5_RUN(n, "blackbody_photons_total", [=] __device__(size_t i) { po[i] = blackbody_photons_total(pT[i], ph[i]); });
    S5_HIP(dout.to_host(out));
    return SIM5GPU_OK;
}

} // extern "C"
