"""A stand-in for sim5_amd.capi's batch calls made of per-ray calls into the CPU oracle, used ONLY to
check the HOST-SIDE LOGIC of sim5_amd/diskraytrace.py (masking, formula order) without a GPU.
TEST INFRASTRUCTURE: never imported by the product."""
import ctypes as C

import numpy as np

import oraclelib as ol
import sim5_amd.capi as capi

GEODESIC_DTYPE, METRIC_DTYPE, TETRAD_DTYPE = capi.GEODESIC_DTYPE, capi.METRIC_DTYPE, capi.TETRAD_DTYPE
_o = None


def _orc():
    global _o
    if _o is None:
        _o = ol.Oracle()
    return _o


def _b(x, n):
    return np.broadcast_to(np.asarray(x, dtype=np.float64), (n,))


def disk_nt_setup(M, a, mdot, alpha, options=0):
    _orc().disk_nt_setup(M, a, mdot, alpha, options)


def disk_nt_mdot():
    return _orc().disk_nt_mdot()


def disk_nt_lumi():
    return _orc().disk_nt_lumi()


def disk_nt_sigma(r):
    return np.array([_orc().disk_nt_sigma(v) for v in np.atleast_1d(r)])


def r_ms(a):
    return np.array([_orc().r_ms(v) for v in np.atleast_1d(a)])


def r_bh(a):
    return np.array([_orc().r_bh(v) for v in np.atleast_1d(a)])


def r_mb(a):
    return np.array([_orc().r_mb(v) for v in np.atleast_1d(a)])


def r_ph(a):
    return np.array([_orc().r_ph(v) for v in np.atleast_1d(a)])


def disk_nt_r_min():
    return _orc().disk_nt_r_min()


def disk_nt_flux(r):
    return np.array([_orc().disk_nt_flux(v) for v in np.atleast_1d(r)])


def disk_nt_ell(r):
    return np.array([_orc().disk_nt_ell(v) for v in np.atleast_1d(r)])


def geodesic_init_inf(incl, a, alpha, beta):
    alpha = np.atleast_1d(alpha); n = alpha.size
    incl, a, beta = _b(incl, n), _b(a, n), _b(beta, n)
    out = np.zeros(n, dtype=GEODESIC_DTYPE); err = np.zeros(n, np.int32); ok = np.zeros(n, np.int32)
    for i in range(n):
        g = ol.Geodesic(); e = C.c_int(0)
        ok[i] = _orc().geodesic_init_inf(incl[i], a[i], alpha[i], beta[i], C.byref(g), C.byref(e))
        err[i] = e.value
        out[i:i + 1] = np.frombuffer(ol.struct_bytes(g), dtype=GEODESIC_DTYPE)
    return out, err, ok


def _gd(rec):
    return ol.Geodesic.from_buffer_copy(rec.tobytes())


def geodesic_find_midplane_crossing(g, order):
    return np.array([_orc().geodesic_find_midplane_crossing(C.byref(_gd(r)), int(order)) for r in np.atleast_1d(g)])


def geodesic_position_rad(g, P):
    g = np.atleast_1d(g)
    return np.array([_orc().geodesic_position_rad(C.byref(_gd(r)), p) for r, p in zip(g, _b(P, g.size))])


def geodesic_position_pol(g, P):
    g = np.atleast_1d(g)
    return np.array([_orc().geodesic_position_pol(C.byref(_gd(r)), p) for r, p in zip(g, _b(P, g.size))])


def geodesic_P_int(g, r, ppc):
    g = np.atleast_1d(g)
    return np.array([_orc().geodesic_P_int(C.byref(_gd(rec)), rr, int(pp)) for rec, rr, pp in zip(g, _b(r, g.size), _b(ppc, g.size))])


def photon_momentum(a, r, m, l, q, rs, ms):
    r = np.atleast_1d(r); n = r.size
    a, m, l, q, rs, ms = (_b(v, n) for v in (a, m, l, q, rs, ms))
    k = np.zeros((n, 4))
    for i in range(n):
        kk = ol.D4(); _orc().photon_momentum(a[i], r[i], m[i], l[i], q[i], rs[i], ms[i], kk); k[i] = list(kk)
    return k


def kerr_metric(a, r, m):
    r = np.atleast_1d(r); n = r.size
    a, m = _b(a, n), _b(m, n)
    out = np.zeros(n, dtype=METRIC_DTYPE)
    for i in range(n):
        g = ol.Metric(); _orc().kerr_metric(a[i], r[i], m[i], C.byref(g))
        out[i:i + 1] = np.frombuffer(ol.struct_bytes(g), dtype=METRIC_DTYPE)
    return out


def _met(rec):
    return ol.Metric.from_buffer_copy(rec.tobytes())


def Omega_from_ell(ell, metric):
    metric = np.atleast_1d(metric)
    ell = _b(ell, metric.size)
    return np.array([_orc().Omega_from_ell(ell[i], C.byref(_met(metric[i]))) for i in range(metric.size)])


def tetrad_surface(metric, Om, V, dhdr):
    metric = np.atleast_1d(metric)
    n = metric.size
    Om, V, dhdr = _b(Om, n), _b(V, n), _b(dhdr, n)
    out = np.zeros(n, dtype=TETRAD_DTYPE)
    for i in range(n):
        t = ol.Tetrad(); _orc().tetrad_surface(C.byref(_met(metric[i])), Om[i], V[i], dhdr[i], C.byref(t))
        out[i:i + 1] = np.frombuffer(ol.struct_bytes(t), dtype=TETRAD_DTYPE)
    return out


def on2bl(v, tetrad):
    v = np.asarray(v, dtype=np.float64).reshape(-1, 4)
    tetrad = np.atleast_1d(tetrad)
    out = np.zeros_like(v)
    for i in range(v.shape[0]):
        t = ol.Tetrad.from_buffer_copy(tetrad[i].tobytes()); vo = ol.D4()
        _orc().on2bl(ol.D4(*v[i]), vo, C.byref(t)); out[i] = list(vo)
    return out


def dotprod(v1, v2, metric):
    v1 = np.asarray(v1).reshape(-1, 4); v2 = np.asarray(v2).reshape(-1, 4)
    metric = np.atleast_1d(metric)
    return np.array([_orc().dotprod(ol.D4(*v1[i]), ol.D4(*v2[i]), C.byref(_met(metric[i]))) for i in range(v1.shape[0])])
