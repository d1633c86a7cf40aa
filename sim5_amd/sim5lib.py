"""sim5lib -- the names of SIM5's SWIG module (ref: src/sim5lib.swig:11-47), served by libsim5gpu.so.

Python callers written against `from sim5lib import *` (ref: python/sim5diskraytrace.py:13,
python/sim5diskmodel.py:15) keep working when this module is the `sim5lib` they import, e.g.

    import sys, sim5_amd.sim5lib as sim5lib;  sys.modules["sim5lib"] = sim5lib

Each call is one batch call with n = 1 through the C-ABI (sim5_amd/capi.py); no ray arithmetic happens
here and there is no CPU fallback.  Throughput work should use sim5_amd.diskraytrace (batched) or the
whole-job kernels instead; this module is the compatibility layer.

The calls of the image loop (ref python/sim5diskraytrace.py:163-172, 228-250: geodesic_init_inf, then the crossing, the
radius, g-factor and flux of the ray) go through the C host shim built as a shared library (sim5_amd/host/sim5lib.c ->
lib/libsim5shim.so): its per-ray record and its look-ahead by image rows serve Python callers exactly as they serve C
callers -- same code, same bit-for-bit argument checks.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import capi as _c

_HERE = os.path.dirname(os.path.abspath(__file__))
_shim_lib = None


def _shim():
    """libsim5shim.so (built by sim5_amd/host/Makefile; on demand here if the tree was not built), bound to the very
    libsim5gpu.so that sim5_amd.capi has loaded"""
    global _shim_lib
    if _shim_lib is None:
        path = os.path.join(_HERE, "lib", "libsim5shim.so")
        src = os.path.join(_HERE, "host", "sim5lib.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.run(["make", "-s", "-C", os.path.join(_HERE, "host")], check=True)
        os.environ.setdefault("SIM5GPU_LIB", _c.LIB_PATH)
        L = C.CDLL(path)
        D, I, P = C.c_double, C.c_int, C.c_void_p
        L.geodesic_init_inf.restype = I; L.geodesic_init_inf.argtypes = [D, D, D, D, P, C.POINTER(I)]
        L.geodesic_find_midplane_crossing.restype = D; L.geodesic_find_midplane_crossing.argtypes = [P, I]
        L.geodesic_position_rad.restype = D; L.geodesic_position_rad.argtypes = [P, D]
        L.gfactorK.restype = D; L.gfactorK.argtypes = [D, D, D]
        L.disk_nt_flux.restype = D; L.disk_nt_flux.argtypes = [D]
        L.disk_nt_setup.restype = I; L.disk_nt_setup.argtypes = [D, D, D, D, I]
        _shim_lib = L
    return _shim_lib

# ---- constants (ref: src/sim5const.h:24-95) --------------------------------------------------------
TINY = 1e-40
grav_radius = 1.476716e+05
speed_of_light = 2.997925e+10
speed_of_light2 = 8.987554e+20
boltzmann_k = 1.380650e-16
sb_sigma = 5.670400e-05
sigma_thomson = 6.652458e-25
parsec = 3.085680e+18
mass_proton = 1.672622e-24
mass_electron = 9.109382e-28
solar_mass = 1.988920e+33
grav_const = 6.673000e-08
planck_h = 6.626069e-27
Mdot_Edd = 2.225475942e+18
L_Edd = 1.257142540e+38
kev2freq = 2.417990e+17
freq2kev = 4.135667e-18
kev2erg = 1.602177e-09
erg2kev = 6.241507e+08

GEOD_TYPE_RR, GEOD_TYPE_RR_DBL, GEOD_TYPE_RR_BH, GEOD_TYPE_RC, GEOD_TYPE_CC = 40, 41, 42, 2, 0
GD_OK = 0
RTOPT_NONE, RTOPT_FLAT, RTOPT_POLARIZATION = 0, 1, 2


# ---- the SWIG helper types (cpointer.i / carrays.i: ref src/sim5lib.swig:20-27) -------------------------
class intp:
    def __init__(self):
        self._v = 0

    def assign(self, v):
        self._v = int(v)

    def value(self):
        return self._v


class doublep:
    def __init__(self):
        self._v = 0.0

    def assign(self, v):
        self._v = float(v)

    def value(self):
        return self._v


class doubleArray:
    def __init__(self, n):
        self._a = np.zeros(int(n))

    def __getitem__(self, i):
        return float(self._a[i])

    def __setitem__(self, i, v):
        self._a[i] = v

    def __len__(self):
        return self._a.size


class intArray(doubleArray):
    def __init__(self, n):
        self._a = np.zeros(int(n), dtype=np.int32)


def double_array_getitem(a, i):
    return float(_arr(a)[i])


def double_array_setitem(a, i, v):
    _arr(a)[i] = v


def sim5vector(components):
    v = doubleArray(4)
    for i in range(4):
        v[i] = components[i]
    return v


def _arr(v):
    return v._a if isinstance(v, doubleArray) else np.asarray(v, dtype=np.float64)


class _Record:
    """attribute access to a one-element numpy record with a SIM5 struct layout"""
    _dtype = None

    def __init__(self):
        object.__setattr__(self, "_rec", np.zeros(1, dtype=self._dtype))

    def __getattr__(self, name):
        rec = object.__getattribute__(self, "_rec")
        if name in rec.dtype.names:
            v = rec[name][0]
            return v.item() if np.ndim(v) == 0 else v
        raise AttributeError(name)

    def __setattr__(self, name, value):
        self._rec[name][0] = value


class geodesic(_Record):
    _dtype = _c.GEODESIC_DTYPE


class raytrace_data(_Record):
    _dtype = _c.RAYTRACE_DTYPE


class sim5metric(_Record):
    _dtype = _c.METRIC_DTYPE


class _MetricView:
    def __init__(self, rec):
        self._m = rec

    def __getattr__(self, name):
        return float(self._m[name])


class sim5tetrad(_Record):
    _dtype = _c.TETRAD_DTYPE

    @property
    def metric(self):
        return _MetricView(self._rec["metric"][0])

    @property
    def e(self):
        return self._rec["e"][0]


def _met(m):
    return m._rec if isinstance(m, sim5metric) else (m._m.reshape(1) if isinstance(m, _MetricView) else m)


# ---- functions ----------------------------------------------------------------------------------------
def r_bh(a):
    return float(_c.r_bh([a])[0])


def r_ms(a):
    return float(_c.r_ms([a])[0])


def r_mb(a):
    return float(_c.r_mb([a])[0])


def r_ph(a):
    return float(_c.r_ph([a])[0])


def OmegaK(r, a):
    return float(_c.OmegaK([r], a)[0])


def ellK(r, a):
    return float(_c.ellK([r], a)[0])


def gfactorK(r, a, l):
    return _shim().gfactorK(r, a, l)


def kerr_metric(a, r, m, metric):
    metric._rec[:] = _c.kerr_metric(a, [r], m)


def kerr_connection(a, r, m, G=None):
    return _c.kerr_connection(a, [r], m)[0]


def Omega_from_ell(ell, metric):
    return float(_c.Omega_from_ell(ell, _met(metric))[0])


def dotprod(v1, v2, metric):
    return float(_c.dotprod(_arr(v1), _arr(v2), None if metric is None else _met(metric))[0])


def tetrad_zamo(metric, tetrad):
    tetrad._rec[:] = _c.tetrad_zamo(_met(metric))


def tetrad_azimuthal(metric, Omega, tetrad):
    tetrad._rec[:] = _c.tetrad_azimuthal(_met(metric), Omega)


def tetrad_surface(metric, Omega, V, dhdr, tetrad):
    tetrad._rec[:] = _c.tetrad_surface(_met(metric), Omega, V, dhdr)


def bl2on(vin, vout, tetrad):
    _arr(vout)[:] = _c.bl2on(_arr(vin), tetrad._rec)[0]


def on2bl(vin, vout, tetrad):
    _arr(vout)[:] = _c.on2bl(_arr(vin), tetrad._rec)[0]


def photon_momentum(a, r, m, l, q, r_sign, m_sign, k):
    _arr(k)[:] = _c.photon_momentum(a, [r], m, l, q, r_sign, m_sign)[0]


def photon_carter_const(k, metric):
    return float(_c.photon_carter_const(_arr(k), _met(metric))[0])


def geodesic_init_inf(i, a, alpha, beta, g, status=None):
    err = C.c_int(0)
    ok = _shim().geodesic_init_inf(i, a, alpha, beta, g._rec.ctypes.data, C.byref(err))
    if status is not None:
        status.assign(err.value)
    return int(ok)


def geodesic_init_src(a, r, m, k, ppc, g, status=None):
    rec, err, ok = _c.geodesic_init_src(a, r, m, _arr(k), ppc)
    g._rec[:] = rec
    if status is not None:
        status.assign(int(err[0]))
    return int(ok[0])


def geodesic_find_midplane_crossing(g, order):
    return _shim().geodesic_find_midplane_crossing(g._rec.ctypes.data, int(order))


def geodesic_P_int(g, r, ppc):
    return float(_c.geodesic_P_int(g._rec, r, ppc)[0])


def geodesic_position_rad(g, P):
    return _shim().geodesic_position_rad(g._rec.ctypes.data, P)


def geodesic_position_pol(g, P):
    return float(_c.geodesic_position_pol(g._rec, P)[0])


def geodesic_dm_sign(g, P):
    return float(_c.geodesic_dm_sign(g._rec, P)[0])


def geodesic_position_azm(g, r, m, P):
    return float(_c.geodesic_position_azm(g._rec, r, m, P)[0])


def geodesic_timedelay(g, P1, r1, m1, P2, r2, m2):
    return float(_c.geodesic_timedelay(g._rec, P1, r1, m1, P2, r2, m2)[0])


def geodesic_momentum(g, P, r, m, k):
    _arr(k)[:] = _c.geodesic_momentum(g._rec, P, r, m)[0]


def geodesic_follow(g, step, P, r, m, status=None):
    P2, r2, m2, st = _c.geodesic_follow(g._rec, step, P.value(), r.value(), m.value())
    P.assign(P2[0]); r.assign(r2[0]); m.assign(m2[0])
    if status is not None:
        status.assign(int(st[0]))


def disk_nt_setup(M, a, mdot_or_L, alpha, options=0):
    _shim().disk_nt_setup(M, a, mdot_or_L, alpha, int(options))        # (through the shim: its records know the disk they were made for)
    return 0


def disk_nt_done():
    pass


def disk_nt_r_min():
    return _c.disk_nt_r_min()


def disk_nt_flux(r):
    return _shim().disk_nt_flux(r)


def disk_nt_lumi():
    return _c.disk_nt_lumi()


def disk_nt_mdot():
    return _c.disk_nt_mdot()


def disk_nt_sigma(r):
    return float(_c.disk_nt_sigma([r])[0])


def disk_nt_ell(r):
    return float(_c.disk_nt_ell([r])[0])


def disk_nt_vr(r):
    return 0.0


def disk_nt_h(r):
    return 0.0


def disk_nt_dhdr(r):
    return 0.0


def raytrace_prepare(bh_spin, x, k, precision, options, rtd):
    rtd._rec[:] = _c.raytrace_prepare(bh_spin, _arr(x), _arr(k), precision, options)


def raytrace(x, k, step, rtd):
    x2, k2, st, r2 = _c.raytrace(_arr(x), _arr(k), step.value(), rtd._rec, 1)
    _arr(x)[:] = x2[0]; _arr(k)[:] = k2[0]; step.assign(st[0]); rtd._rec[:] = r2


def raytrace_error(x, k, rtd):
    return float(_c.raytrace_error(_arr(x), _arr(k), rtd._rec)[0])


def polarization_constant(k, f, metric):
    w = _c.polarization_constant(_arr(k), _arr(f), _met(metric))[0]
    return complex(w[0], w[1])


def polarization_vector(k, wp, metric, f):
    _arr(f)[:] = _c.polarization_vector(_arr(k), [[wp.real, wp.imag]], _met(metric))[0]


def polarization_constant_infinity(a, alpha, beta, incl):
    w = _c.polarization_constant_infinity(a, [alpha], beta, incl)[0]
    return complex(w[0], w[1])


def polarization_angle_rotation(a, inc, alpha, beta, kappa):
    return float(_c.polarization_angle_rotation(a, inc, [alpha], beta, [[kappa.real, kappa.imag]])[0])


def blackbody_Iv(T, hardf, cos_mu, E):
    return float(_c.blackbody_Iv(T, hardf, cos_mu, [E])[0])


def rf(x, y, z):
    return float(_c.elliptic("rf", [x], y, z)[0])


def elliptic_k(m):
    return float(_c.elliptic("elliptic_k", [m])[0])


def jacobi_isn(z, m):
    return float(_c.elliptic("jacobi_isn", [z], m)[0])


def jacobi_icn(z, m):
    return float(_c.elliptic("jacobi_icn", [z], m)[0])


def jacobi_sn(u, m):
    return float(_c.elliptic("jacobi_sn", [u], m)[0])


def jacobi_cn(u, m):
    return float(_c.elliptic("jacobi_cn", [u], m)[0])


# ---- the rest of the headers the SWIG interface exports (ref src/sim5lib.swig:33-39: sim5elliptic.h, sim5kerr.h,
#      sim5kerr-geod.h, ...) ------------------------------------------------------------------------------------
def flat_metric(r, m, metric):
    metric._rec[:] = _c.flat_metric([r], m)


def flat_metric_contravariant(r, m, metric):
    metric._rec[:] = _c.flat_metric_contravariant([r], m)


def kerr_metric_contravariant(a, r, m, metric):
    metric._rec[:] = _c.kerr_metric_contravariant(a, [r], m)


def kerr_newman_metric(a, Q, r, m, metric):
    metric._rec[:] = _c.kerr_newman_metric(a, Q, [r], m)


def kerr_newman_metric_contravariant(a, Q, r, m, metric):
    metric._rec[:] = _c.kerr_newman_metric_contravariant(a, Q, [r], m)


def kerr_newman_connection(a, Q, r, m, G=None):
    return _c.kerr_newman_connection(a, Q, [r], m)[0]


def flat_connection(r, m, G=None):
    return _c.flat_connection([r], m)[0]


def Gamma(G, U, V, result):
    _arr(result)[:] = _c.Gamma(np.asarray(G, dtype=np.float64), _arr(U), _arr(V))[0]


def vector_set(x, x0, x1, x2, x3):
    _arr(x)[:] = (x0, x1, x2, x3)


def vector_copy(src, dst):
    _arr(dst)[:] = _arr(src)


def vector_multiply(v, factor):
    _arr(v)[:] = _arr(v) * factor


def vector_covariant(v1, v2, metric):
    _arr(v2)[:] = _c.vector_covariant(_arr(v1), None if metric is None else _met(metric))[0]


def vector_norm(v, metric):
    return float(_c.vector_norm(_arr(v), None if metric is None else _met(metric))[0])


def vector_3norm(v):
    return float(_c.vector_3norm(_arr(v))[0])


def vector_norm_to(v, norm, metric):
    _arr(v)[:] = _c.vector_norm_to(_arr(v), norm, None if metric is None else _met(metric))[0]


def vector_norm_to_null(v, V0, metric):
    _arr(v)[:] = _c.vector_norm_to_null(_arr(v), V0, None if metric is None else _met(metric))[0]


def tetrad_general(metric, U, tetrad):
    tetrad._rec[:] = _c.tetrad_general(_met(metric), _arr(U))


def tetrad_radial(metric, v_r, tetrad):
    tetrad._rec[:] = _c.tetrad_radial(_met(metric), v_r)


def omega_r(r, a):
    return float(_c.omega_r([r], a)[0])


def omega_z(r, a):
    return float(_c.omega_z([r], a)[0])


def ell_from_Omega(Omega, metric):
    return float(_c.ell_from_Omega(Omega, _met(metric))[0])


def fourvelocity_zamo(metric, U):
    _arr(U)[:] = _c.fourvelocity_zamo(_met(metric))[0]


def fourvelocity_azimuthal(Omega, metric, U):
    _arr(U)[:] = _c.fourvelocity_azimuthal(Omega, _met(metric))[0]


def fourvelocity_radial(vr, metric, U):
    _arr(U)[:] = _c.fourvelocity_radial(vr, _met(metric))[0]


def fourvelocity_norm(U1, U2, U3, metric):
    return float(_c.fourvelocity_norm(U1, U2, U3, _met(metric))[0])


def fourvelocity(U1, U2, U3, metric, U):
    _arr(U)[:] = _c.fourvelocity(U1, U2, U3, _met(metric))[0]


def photon_motion_constants(a, r, m, k, L, Q):
    l_, q_ = _c.photon_motion_constants(a, [r], m, _arr(k))
    L.assign(l_[0]); Q.assign(q_[0])


def geodesic_position(g, P, x):
    pass                                       # an empty stub in the reference too (ref src/sim5kerr-geod.c:266-283)


def geodesic_position_pol_sign_k_theta(g, P):
    return float(_c.geodesic_position_pol_sign_k_theta(g._rec, P)[0])


def elliptic_f(phi, m):
    return float(_c.legendre("elliptic_f", [phi], m)[0])


def elliptic_e_sin(sin_phi, m):
    return float(_c.legendre("elliptic_e_sin", [sin_phi], m)[0])


def elliptic_pi_sin(sin_phi, n, m):
    return float(_c.legendre("elliptic_pi_sin", [sin_phi], m, nn=n)[0])


def elliptic_pi(phi, n, m):
    return complex(_c.legendre("elliptic_pi", [phi], m, nn=n)[0])


def blackbody(T, hardf, cos_mu, E, Iv, en_bins):
    if T > 0.0 and en_bins > 0:
        _arr(Iv)[:en_bins] = _c.blackbody(T, hardf, cos_mu, _arr(E)[:en_bins])


def blackbody_photons(T, hardf, cos_mu, E):
    return float(_c.blackbody_photons(T, hardf, cos_mu, [E])[0])


def blackbody_photons_total(T, hardf):
    return float(_c.blackbody_photons_total([T], hardf)[0])
