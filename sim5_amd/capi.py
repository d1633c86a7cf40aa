"""ctypes binding of include/sim5gpu.h (the C-ABI of libsim5gpu.so).

Fails loudly: a missing library raises ImportError at import; any non-zero status from the
library raises Sim5GpuError with the library's own message.  Nothing here computes rays.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SIM5GPU_LIB (the variable the C host shim reads too) selects another build of the library, e.g. an experiment
# variant sim5_amd/lib/ab_<name>.so of tests/tools/ab_build.sh: the in-tree library is never overwritten
LIB_PATH = os.environ.get("SIM5GPU_LIB") or os.path.join(_HERE, "lib", "libsim5gpu.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "sim5gpu.h")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "sim5_amd: %s is missing -- build it with `python -m sim5_amd.build` "
        "(there is no CPU fallback)" % LIB_PATH)

_lib = C.CDLL(LIB_PATH)

D = C.c_double
I = C.c_int
SZ = C.c_size_t
VP = C.c_void_p


class Sim5GpuError(RuntimeError):
    pass


class Geodesic(C.Structure):
    _fields_ = [("a", D), ("alpha", D), ("beta", D), ("incl", D), ("cos_i", D),
                ("l", D), ("q", D),
                ("r1", D * 2), ("r2", D * 2), ("r3", D * 2), ("r4", D * 2),
                ("nrr", I), ("type", I),
                ("m2p", D), ("m2m", D), ("mm", D), ("mK", D),
                ("rp", D), ("dmdp_inf", D),
                ("Rpc", D), ("Tpp", D), ("Tip", D),
                ("k", D * 4), ("p", D)]


class Metric(C.Structure):
    _fields_ = [("a", D), ("r", D), ("m", D), ("g00", D), ("g11", D), ("g22", D),
                ("g33", D), ("g03", D)]


class Tetrad(C.Structure):
    _fields_ = [("e", (D * 4) * 4), ("metric", Metric)]


class RaytraceData(C.Structure):
    _fields_ = [("opt_gr", I), ("opt_pol", I), ("step_epsilon", D),
                ("bh_spin", D), ("E", D), ("Q", D), ("WP", D * 2),
                ("pass_", I), ("refines", I), ("dk", D * 4), ("df", D * 4),
                ("kt", D), ("error", C.c_float)]


class Stokes(C.Structure):
    _fields_ = [("i", D), ("q", D), ("u", D), ("v", D), ("tau", D)]


class ImageDesc(C.Structure):
    _fields_ = [("nx", I), ("ny", I), ("y0", I), ("y1", I),
                ("a", D), ("incl", D), ("rmax", D), ("rms", D),
                ("bh_mass", D), ("mdot", D), ("alpha_visc", D),
                ("max_order", I), ("flags", I), ("pol_degree", D), ("stripe_rows", I), ("stripe_step", I),
                ("disk_spin", D)]


class ImageAux(C.Structure):
    _fields_ = [("cls", VP), ("gtype", VP), ("r", VP), ("g", VP), ("flux", VP)]


class TorusDesc(C.Structure):
    _fields_ = [("img", ImageDesc), ("r0", D), ("dl_max", D), ("precision", D),
                ("options", I), ("max_steps", I), ("max_error", D),
                ("r_stop_in", D), ("r_stop_out", D), ("shape", I),
                ("torus_r", D), ("torus_w", D), ("torus_l", D),
                ("emis0", D), ("absorb0", D)]


class TorusAux(C.Structure):
    _fields_ = [("steps", VP), ("max_step_error", VP), ("carter_error", VP),
                ("x_end", VP), ("k_end", VP)]


assert C.sizeof(Geodesic) == 240 and C.sizeof(Metric) == 64 and C.sizeof(Tetrad) == 192
assert C.sizeof(RaytraceData) == 144 and C.sizeof(Stokes) == 40

GEODESIC_DTYPE = np.dtype(Geodesic)
METRIC_DTYPE = np.dtype(Metric)
TETRAD_DTYPE = np.dtype(Tetrad)
RAYTRACE_DTYPE = np.dtype(RaytraceData)
STOKES_DTYPE = np.dtype(Stokes)

PX_ERROR, PX_NAN0, PX_HIT0, PX_NAN1, PX_HIT1, PX_MISS = range(6)

_lib.sim5gpu_last_error.restype = C.c_char_p
_lib.sim5gpu_version.restype = C.c_char_p


def _check(rc, what):
    if rc != 0:
        msg = _lib.sim5gpu_last_error().decode(errors="replace")
        raise Sim5GpuError("%s failed (status %d): %s" % (what, rc, msg))


def _f64(x, n=None, cols=None):
    a = np.ascontiguousarray(x, dtype=np.float64)
    if n is not None:
        a = np.ascontiguousarray(np.broadcast_to(a, (n,) if cols is None else (n, cols)))
    return a


def _i32(x, n):
    return np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=np.int32), (n,)))


def _p(a):
    return a.ctypes.data_as(VP)


def device_count():
    _lib.sim5gpu_device_count.restype = I
    return _lib.sim5gpu_device_count()


def set_device(dev):
    _check(_lib.sim5gpu_set_device(I(dev)), "sim5gpu_set_device")


def device_bus_id(dev):
    """PCI bus id of a HIP device ("0000:05:00.0")"""
    buf = C.create_string_buffer(64)
    _check(_lib.sim5gpu_device_bus_id(I(dev), buf, I(64)), "sim5gpu_device_bus_id")
    return buf.value.decode()


def version():
    return _lib.sim5gpu_version().decode()


def synchronize(stream=None):
    _check(_lib.sim5gpu_synchronize(VP(stream or 0)), "sim5gpu_synchronize")


class Event:
    """HIP event recorded on the stream the kernels are launched on."""

    def __init__(self):
        p = VP()
        _check(_lib.sim5gpu_event_create(C.byref(p)), "sim5gpu_event_create")
        self.ptr = p.value

    def record(self, stream=None):
        _check(_lib.sim5gpu_event_record(VP(self.ptr), VP(stream or 0)), "sim5gpu_event_record")

    def elapsed_ms(self, stop):
        ms = C.c_float(0.0)
        _check(_lib.sim5gpu_event_elapsed_ms(VP(self.ptr), VP(stop.ptr), C.byref(ms)), "sim5gpu_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            if self.ptr:
                _lib.sim5gpu_event_destroy(VP(self.ptr))
        except Exception:
            pass


# ---- raw device memory (for hosts that do not use torch) -----------------------------------
class DeviceBuffer:
    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = VP()
        _check(_lib.sim5gpu_malloc(C.byref(p), SZ(self.nbytes)), "sim5gpu_malloc")
        self.ptr = p.value

    def to_numpy(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        _check(_lib.sim5gpu_memcpy_d2h(_p(out), VP(self.ptr), SZ(out.nbytes)), "sim5gpu_memcpy_d2h")
        return out

    def from_numpy(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _check(_lib.sim5gpu_memcpy_h2d(VP(self.ptr), _p(arr), SZ(arr.nbytes)), "sim5gpu_memcpy_h2d")

    def zero(self):
        self.fill(0)

    def fill(self, byte):
        _check(_lib.sim5gpu_memset(VP(self.ptr), I(int(byte)), SZ(self.nbytes)), "sim5gpu_memset")

    def free(self):
        if self.ptr:
            _lib.sim5gpu_free(VP(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def image_view(desc):
    """(rmax, rms, sin i, cos i) a job description resolves to (host arithmetic, no GPU)"""
    rmax, rms, si, ci = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
    _check(_lib.sim5gpu_image_view(C.byref(desc), C.byref(rmax), C.byref(rms), C.byref(si), C.byref(ci)), "sim5gpu_image_view")
    return rmax.value, rms.value, si.value, ci.value


IPC_HANDLE_BYTES = 64


def ipc_export(d_ptr):
    """inter-process handle (bytes) of a device allocation of this process (its start address)"""
    h = (C.c_char * IPC_HANDLE_BYTES)()
    _check(_lib.sim5gpu_ipc_export(VP(d_ptr), h), "sim5gpu_ipc_export")
    return bytes(h.raw)


def ipc_open(handle):
    """device pointer (int) under which another process's allocation is mapped here; close with ipc_close"""
    p = C.c_void_p(0)
    buf = (C.c_char * IPC_HANDLE_BYTES).from_buffer_copy(handle)
    _check(_lib.sim5gpu_ipc_open(buf, C.byref(p)), "sim5gpu_ipc_open")
    return int(p.value)


def ipc_close(d_ptr):
    _check(_lib.sim5gpu_ipc_close(VP(d_ptr)), "sim5gpu_ipc_close")


def words_differ(d_a, d_b, n_words):
    """number of 32-bit words in which two device buffers differ (bit comparison on the device, synchronous)"""
    n = C.c_ulonglong(0)
    _check(_lib.sim5gpu_words_differ(VP(d_a), VP(d_b), SZ(int(n_words)), C.byref(n)), "sim5gpu_words_differ")
    return int(n.value)


# ---- batch forms of the SIM5 per-ray API -----------------------------------------------------
def geodesic_init_inf(incl, a, alpha, beta):
    alpha = _f64(alpha).ravel()
    n = alpha.size
    incl, a, beta = _f64(incl, n), _f64(a, n), _f64(beta, n)
    g = np.zeros(n, dtype=GEODESIC_DTYPE)
    err = np.zeros(n, dtype=np.int32)
    ok = np.zeros(n, dtype=np.int32)
    _check(_lib.sim5gpu_geodesic_init_inf(SZ(n), _p(incl), _p(a), _p(alpha), _p(beta), _p(g), _p(err), _p(ok)),
           "sim5gpu_geodesic_init_inf")
    return g, err, ok


CHAIN_DTYPE = np.dtype([("P", "f8", 2), ("r", "f8", 2), ("g", "f8", 2), ("flux", "f8", 2), ("a", "f8"), ("l", "f8"),
                        ("have_r", "i4", 2), ("valid", "i4"), ("flux_valid", "i4")])


def geodesic_init_inf_chain(incl, a, alpha, beta, fast=False):
    """geodesic_init_inf and the example-04 chain of every ray in one launch (sim5gpu_geodesic_init_inf_chain; fast=True:
    the record in the library's fast arithmetic, sim5gpu_geodesic_init_inf_chain_fast).  Returns (geodesics, err, ok, chain)."""
    alpha = _f64(alpha).ravel()
    n = alpha.size
    incl, a, beta = _f64(incl, n), _f64(a, n), _f64(beta, n)
    g = np.zeros(n, dtype=GEODESIC_DTYPE)
    err = np.zeros(n, dtype=np.int32)
    ok = np.zeros(n, dtype=np.int32)
    ch = np.zeros(n, dtype=CHAIN_DTYPE)
    assert CHAIN_DTYPE.itemsize == 96
    fn = _lib.sim5gpu_geodesic_init_inf_chain_fast if fast else _lib.sim5gpu_geodesic_init_inf_chain
    _check(fn(SZ(n), _p(incl), _p(a), _p(alpha), _p(beta), _p(g), _p(err), _p(ok), _p(ch)), "sim5gpu_geodesic_init_inf_chain")
    return g, err, ok, ch


def geodesic_init_src(a, r, m, k, ppc):
    k = _f64(k).reshape(-1, 4)
    n = k.shape[0]
    a, r, m, ppc = _f64(a, n), _f64(r, n), _f64(m, n), _i32(ppc, n)
    g = np.zeros(n, dtype=GEODESIC_DTYPE)
    err = np.zeros(n, dtype=np.int32)
    ok = np.zeros(n, dtype=np.int32)
    _check(_lib.sim5gpu_geodesic_init_src(SZ(n), _p(a), _p(r), _p(m), _p(k), _p(ppc), _p(g), _p(err), _p(ok)),
           "sim5gpu_geodesic_init_src")
    return g, err, ok


def _geod(g):
    g = np.ascontiguousarray(g, dtype=GEODESIC_DTYPE).ravel()
    return g, g.size


def geodesic_find_midplane_crossing(g, order):
    g, n = _geod(g)
    order = _i32(order, n)
    P = np.empty(n)
    _check(_lib.sim5gpu_geodesic_find_midplane_crossing(SZ(n), _p(g), _p(order), _p(P)),
           "sim5gpu_geodesic_find_midplane_crossing")
    return P


def geodesic_P_int(g, r, ppc):
    g, n = _geod(g)
    r, ppc = _f64(r, n), _i32(ppc, n)
    P = np.empty(n)
    _check(_lib.sim5gpu_geodesic_P_int(SZ(n), _p(g), _p(r), _p(ppc), _p(P)), "sim5gpu_geodesic_P_int")
    return P


def _geod_P(fn, name, g, P):
    g, n = _geod(g)
    P = _f64(P, n)
    out = np.empty(n)
    _check(fn(SZ(n), _p(g), _p(P), _p(out)), name)
    return out


def geodesic_position_rad(g, P):
    return _geod_P(_lib.sim5gpu_geodesic_position_rad, "sim5gpu_geodesic_position_rad", g, P)


def geodesic_position_pol(g, P):
    return _geod_P(_lib.sim5gpu_geodesic_position_pol, "sim5gpu_geodesic_position_pol", g, P)


def geodesic_dm_sign(g, P):
    return _geod_P(_lib.sim5gpu_geodesic_dm_sign, "sim5gpu_geodesic_dm_sign", g, P)


def geodesic_momentum(g, P, r=0.0, m=0.0):
    g, n = _geod(g)
    P, r, m = _f64(P, n), _f64(r, n), _f64(m, n)
    k = np.zeros((n, 4))
    _check(_lib.sim5gpu_geodesic_momentum(SZ(n), _p(g), _p(P), _p(r), _p(m), _p(k)), "sim5gpu_geodesic_momentum")
    return k


def geodesic_position_azm(g, r, m, P):
    g, n = _geod(g)
    r, m, P = _f64(r, n), _f64(m, n), _f64(P, n)
    out = np.empty(n)
    _check(_lib.sim5gpu_geodesic_position_azm(SZ(n), _p(g), _p(r), _p(m), _p(P), _p(out)),
           "sim5gpu_geodesic_position_azm")
    return out


def geodesic_timedelay(g, P1, r1, m1, P2, r2, m2):
    g, n = _geod(g)
    P1, r1, m1, P2, r2, m2 = (_f64(v, n) for v in (P1, r1, m1, P2, r2, m2))
    out = np.empty(n)
    _check(_lib.sim5gpu_geodesic_timedelay(SZ(n), _p(g), _p(P1), _p(r1), _p(m1), _p(P2), _p(r2), _p(m2), _p(out)),
           "sim5gpu_geodesic_timedelay")
    return out


INTEGRALS = ["elliptic_f_cos", "elliptic_e_cos", "elliptic_pi_complete", "elliptic_pi_cos", "integral_C2",
             "integral_C2_cos", "integral_Z1", "integral_Z2", "integral_Rm1", "integral_Rm2", "integral_R1",
             "integral_R2", "integral_R_r0_re", "integral_R_r0_re_inf", "integral_R_r1_re", "integral_R_r2_re",
             "integral_R_rp_re", "integral_R_rp_re_inf", "integral_R_r0_cc", "integral_R_r0_cc_inf",
             "integral_R_r1_cc", "integral_R_r2_cc", "integral_R_rp_cc2", "integral_R_rp_cc2_inf",
             "integral_T_m0", "integral_T_m2", "integral_T_mp"]


def integral(name, *args):
    """One of the 27 Legendre / Byrd & Friedman integrals (reference argument order; a complex root is two
    arguments re, im).  Arguments broadcast to a common length."""
    cols = np.broadcast_arrays(*[np.asarray(a, dtype=np.float64).ravel() for a in args])
    n = cols[0].size
    packed = np.ascontiguousarray(np.stack(cols, axis=0))
    out = np.empty(n)
    _check(_lib.sim5gpu_integral(I(INTEGRALS.index(name)), SZ(n), I(len(cols)), _p(packed), _p(out)),
           "sim5gpu_integral(%s)" % name)
    return out


def vector_norm_to(v, norm, metric=None):
    v = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1, 4)).copy()
    n = v.shape[0]
    norm = _f64(norm, n)
    _check(_lib.sim5gpu_vector_norm_to(SZ(n), _p(v), _p(norm), _p(metric) if metric is not None else None),
           "sim5gpu_vector_norm_to")
    return v


def geodesic_follow(g, step, P, r, m):
    g, n = _geod(g)
    step = _f64(step, n)
    P, r, m = _f64(P, n).copy(), _f64(r, n).copy(), _f64(m, n).copy()
    st = np.zeros(n, dtype=np.int32)
    _check(_lib.sim5gpu_geodesic_follow(SZ(n), _p(g), _p(step), _p(P), _p(r), _p(m), _p(st)),
           "sim5gpu_geodesic_follow")
    return P, r, m, st


def photon_momentum(a, r, m, l, q, r_sign, m_sign):
    r = _f64(r).ravel()
    n = r.size
    a, m, l, q, rs, ms = (_f64(v, n) for v in (a, m, l, q, r_sign, m_sign))
    k = np.zeros((n, 4))
    _check(_lib.sim5gpu_photon_momentum(SZ(n), _p(a), _p(r), _p(m), _p(l), _p(q), _p(rs), _p(ms), _p(k)),
           "sim5gpu_photon_momentum")
    return k


def photon_motion_constants(a, r, m, k):
    k = _f64(k).reshape(-1, 4)
    n = k.shape[0]
    a, r, m = _f64(a, n), _f64(r, n), _f64(m, n)
    L, Q = np.empty(n), np.empty(n)
    _check(_lib.sim5gpu_photon_motion_constants(SZ(n), _p(a), _p(r), _p(m), _p(k), _p(L), _p(Q)),
           "sim5gpu_photon_motion_constants")
    return L, Q


def photon_carter_const(k, metric):
    k = _f64(k).reshape(-1, 4)
    n = k.shape[0]
    metric = np.ascontiguousarray(metric, dtype=METRIC_DTYPE).ravel()
    Q = np.empty(n)
    _check(_lib.sim5gpu_photon_carter_const(SZ(n), _p(k), _p(metric), _p(Q)), "sim5gpu_photon_carter_const")
    return Q


def _unary(fn, name, x):
    x = _f64(x).ravel()
    out = np.empty(x.size)
    _check(fn(SZ(x.size), _p(x), _p(out)), name)
    return out


def r_bh(a):
    return _unary(_lib.sim5gpu_r_bh, "sim5gpu_r_bh", a)


def r_ms(a):
    return _unary(_lib.sim5gpu_r_ms, "sim5gpu_r_ms", a)


def r_mb(a):
    return _unary(_lib.sim5gpu_r_mb, "sim5gpu_r_mb", a)


def r_ph(a):
    return _unary(_lib.sim5gpu_r_ph, "sim5gpu_r_ph", a)


def OmegaK(r, a):
    r = _f64(r).ravel()
    a = _f64(a, r.size)
    out = np.empty(r.size)
    _check(_lib.sim5gpu_OmegaK(SZ(r.size), _p(r), _p(a), _p(out)), "sim5gpu_OmegaK")
    return out


def ellK(r, a):
    r = _f64(r).ravel()
    a = _f64(a, r.size)
    out = np.empty(r.size)
    _check(_lib.sim5gpu_ellK(SZ(r.size), _p(r), _p(a), _p(out)), "sim5gpu_ellK")
    return out


def Omega_from_ell(ell, metric):
    metric = np.ascontiguousarray(metric, dtype=METRIC_DTYPE).ravel()
    ell = _f64(ell, metric.size)
    out = np.empty(metric.size)
    _check(_lib.sim5gpu_Omega_from_ell(SZ(metric.size), _p(ell), _p(metric), _p(out)), "sim5gpu_Omega_from_ell")
    return out


def dotprod(v1, v2, metric=None):
    v1 = _f64(v1).reshape(-1, 4)
    n = v1.shape[0]
    v2 = _f64(v2, n, 4)
    mp = None
    if metric is not None:
        metric = np.ascontiguousarray(metric, dtype=METRIC_DTYPE).ravel()
        mp = _p(metric)
    out = np.empty(n)
    _check(_lib.sim5gpu_dotprod(SZ(n), _p(v1), _p(v2), mp, _p(out)), "sim5gpu_dotprod")
    return out


def gfactorK(r, a, l):
    r = _f64(r).ravel()
    n = r.size
    a, l = _f64(a, n), _f64(l, n)
    g = np.empty(n)
    _check(_lib.sim5gpu_gfactorK(SZ(n), _p(r), _p(a), _p(l), _p(g)), "sim5gpu_gfactorK")
    return g


def kerr_metric(a, r, m):
    r = _f64(r).ravel()
    n = r.size
    a, m = _f64(a, n), _f64(m, n)
    out = np.zeros(n, dtype=METRIC_DTYPE)
    _check(_lib.sim5gpu_kerr_metric(SZ(n), _p(a), _p(r), _p(m), _p(out)), "sim5gpu_kerr_metric")
    return out


def kerr_connection(a, r, m):
    r = _f64(r).ravel()
    n = r.size
    a, m = _f64(a, n), _f64(m, n)
    G = np.zeros((n, 4, 4, 4))
    _check(_lib.sim5gpu_kerr_connection(SZ(n), _p(a), _p(r), _p(m), _p(G)), "sim5gpu_kerr_connection")
    return G


def tetrad_zamo(metric):
    metric = np.ascontiguousarray(metric, dtype=METRIC_DTYPE).ravel()
    n = metric.size
    t = np.zeros(n, dtype=TETRAD_DTYPE)
    _check(_lib.sim5gpu_tetrad_zamo(SZ(n), _p(metric), _p(t)), "sim5gpu_tetrad_zamo")
    return t


def tetrad_azimuthal(metric, Omega):
    metric = np.ascontiguousarray(metric, dtype=METRIC_DTYPE).ravel()
    n = metric.size
    Omega = _f64(Omega, n)
    t = np.zeros(n, dtype=TETRAD_DTYPE)
    _check(_lib.sim5gpu_tetrad_azimuthal(SZ(n), _p(metric), _p(Omega), _p(t)), "sim5gpu_tetrad_azimuthal")
    return t


def tetrad_surface(metric, Omega, V, dhdr):
    metric = np.ascontiguousarray(metric, dtype=METRIC_DTYPE).ravel()
    n = metric.size
    Omega, V, dhdr = _f64(Omega, n), _f64(V, n), _f64(dhdr, n)
    t = np.zeros(n, dtype=TETRAD_DTYPE)
    _check(_lib.sim5gpu_tetrad_surface(SZ(n), _p(metric), _p(Omega), _p(V), _p(dhdr), _p(t)),
           "sim5gpu_tetrad_surface")
    return t


def _frame(fn, name, v, t):
    v = _f64(v).reshape(-1, 4)
    n = v.shape[0]
    t = np.ascontiguousarray(t, dtype=TETRAD_DTYPE).ravel()
    out = np.zeros((n, 4))
    _check(fn(SZ(n), _p(v), _p(out), _p(t)), name)
    return out


def bl2on(v, t):
    return _frame(_lib.sim5gpu_bl2on, "sim5gpu_bl2on", v, t)


def on2bl(v, t):
    return _frame(_lib.sim5gpu_on2bl, "sim5gpu_on2bl", v, t)


ELLIPTIC = {"rf": 0, "elliptic_k": 1, "jacobi_isn": 2, "jacobi_icn": 3, "jacobi_itn": 4,
            "jacobi_sn": 5, "jacobi_cn": 6, "jacobi_dn": 7, "rd": 8, "rc": 9, "rj": 10, "elliptic_f_sin": 11}


def elliptic(name, x, y=None, z=None, w=None):
    x = _f64(x).ravel()
    n = x.size
    y = _f64(y, n) if y is not None else None
    z = _f64(z, n) if z is not None else None
    w = _f64(w, n) if w is not None else None
    out = np.empty(n)
    _check(_lib.sim5gpu_elliptic(I(ELLIPTIC[name]), SZ(n), _p(x),
                                 _p(y) if y is not None else None,
                                 _p(z) if z is not None else None,
                                 _p(w) if w is not None else None, _p(out)), "sim5gpu_elliptic(%s)" % name)
    return out


def disk_nt_setup(M, a, mdot, alpha, options=0):
    _check(_lib.sim5gpu_disk_nt_setup(D(M), D(a), D(mdot), D(alpha), I(options)), "sim5gpu_disk_nt_setup")


def disk_nt_r_min():
    v = D(0.0)
    _check(_lib.sim5gpu_disk_nt_r_min(C.byref(v)), "sim5gpu_disk_nt_r_min")
    return v.value


def disk_nt_mdot():
    v = D(0.0)
    _check(_lib.sim5gpu_disk_nt_mdot(C.byref(v)), "sim5gpu_disk_nt_mdot")
    return v.value


def disk_nt_lumi():
    v = D(0.0)
    _check(_lib.sim5gpu_disk_nt_lumi(C.byref(v)), "sim5gpu_disk_nt_lumi")
    return v.value


def disk_nt_sigma(r):
    r = _f64(r).ravel()
    out = np.empty(r.size)
    _check(_lib.sim5gpu_disk_nt_sigma(SZ(r.size), _p(r), _p(out)), "sim5gpu_disk_nt_sigma")
    return out


def disk_nt_flux(r):
    r = _f64(r).ravel()
    out = np.empty(r.size)
    _check(_lib.sim5gpu_disk_nt_flux(SZ(r.size), _p(r), _p(out)), "sim5gpu_disk_nt_flux")
    return out


def disk_nt_ell(r):
    r = _f64(r).ravel()
    out = np.empty(r.size)
    _check(_lib.sim5gpu_disk_nt_ell(SZ(r.size), _p(r), _p(out)), "sim5gpu_disk_nt_ell")
    return out


def raytrace_prepare(bh_spin, x, k, precision, options):
    x = _f64(x).reshape(-1, 4)
    n = x.shape[0]
    k = _f64(k, n, 4)
    a, prec, opt = _f64(bh_spin, n), _f64(precision, n), _i32(options, n)
    rtd = np.zeros(n, dtype=RAYTRACE_DTYPE)
    _check(_lib.sim5gpu_raytrace_prepare(SZ(n), _p(a), _p(x), _p(k), _p(prec), _p(opt), _p(rtd)),
           "sim5gpu_raytrace_prepare")
    return rtd


def raytrace(x, k, step, rtd, nsteps=1):
    x = _f64(x).reshape(-1, 4).copy()
    n = x.shape[0]
    k = _f64(k, n, 4).copy()
    step = _f64(step, n).copy()
    rtd = np.ascontiguousarray(rtd, dtype=RAYTRACE_DTYPE).ravel().copy()
    _check(_lib.sim5gpu_raytrace(SZ(n), _p(x), _p(k), _p(step), _p(rtd), I(nsteps)), "sim5gpu_raytrace")
    return x, k, step, rtd


def raytrace_error(x, k, rtd):
    x = _f64(x).reshape(-1, 4)
    n = x.shape[0]
    k = _f64(k, n, 4)
    rtd = np.ascontiguousarray(rtd, dtype=RAYTRACE_DTYPE).ravel()
    out = np.empty(n)
    _check(_lib.sim5gpu_raytrace_error(SZ(n), _p(x), _p(k), _p(rtd), _p(out)), "sim5gpu_raytrace_error")
    return out


def polarization_constant(k, f, metric):
    k = _f64(k).reshape(-1, 4)
    n = k.shape[0]
    f = _f64(f, n, 4)
    metric = np.ascontiguousarray(metric, dtype=METRIC_DTYPE).ravel()
    wp = np.empty((n, 2))
    _check(_lib.sim5gpu_polarization_constant(SZ(n), _p(k), _p(f), _p(metric), _p(wp)),
           "sim5gpu_polarization_constant")
    return wp


def polarization_vector(k, wp, metric):
    k = _f64(k).reshape(-1, 4)
    n = k.shape[0]
    wp = _f64(wp, n, 2)
    metric = np.ascontiguousarray(metric, dtype=METRIC_DTYPE).ravel()
    f = np.empty((n, 4))
    _check(_lib.sim5gpu_polarization_vector(SZ(n), _p(k), _p(wp), _p(metric), _p(f)),
           "sim5gpu_polarization_vector")
    return f


def polarization_constant_infinity(a, alpha, beta, incl):
    alpha = _f64(alpha).ravel()
    n = alpha.size
    a, beta, incl = _f64(a, n), _f64(beta, n), _f64(incl, n)
    wp = np.empty((n, 2))
    _check(_lib.sim5gpu_polarization_constant_infinity(SZ(n), _p(a), _p(alpha), _p(beta), _p(incl), _p(wp)),
           "sim5gpu_polarization_constant_infinity")
    return wp


def polarization_angle_rotation(a, inc, alpha, beta, wp):
    alpha = _f64(alpha).ravel()
    n = alpha.size
    a, inc, beta = _f64(a, n), _f64(inc, n), _f64(beta, n)
    wp = _f64(wp, n, 2)
    out = np.empty(n)
    _check(_lib.sim5gpu_polarization_angle_rotation(SZ(n), _p(a), _p(inc), _p(alpha), _p(beta), _p(wp), _p(out)),
           "sim5gpu_polarization_angle_rotation")
    return out


def blackbody_Iv(T, hardf, cos_mu, E):
    E = _f64(E).ravel()
    n = E.size
    T, hardf, cos_mu = _f64(T, n), _f64(hardf, n), _f64(cos_mu, n)
    out = np.empty(n)
    _check(_lib.sim5gpu_blackbody_Iv(SZ(n), _p(T), _p(hardf), _p(cos_mu), _p(E), _p(out)), "sim5gpu_blackbody_Iv")
    return out


# ---- (1b) the remaining public SIM5 prototypes (include/sim5gpu.h group (1b)) -------------------
def _metric_or_none(metric):
    if metric is None:
        return None, None
    metric = np.ascontiguousarray(metric, dtype=METRIC_DTYPE).ravel()
    return metric, _p(metric)


def flat_metric(r, m, contravariant=False):
    r = _f64(r).ravel()
    n = r.size
    m = _f64(m, n)
    out = np.zeros(n, dtype=METRIC_DTYPE)
    fn = _lib.sim5gpu_flat_metric_contravariant if contravariant else _lib.sim5gpu_flat_metric
    _check(fn(SZ(n), _p(r), _p(m), _p(out)), "sim5gpu_flat_metric")
    return out


def flat_metric_contravariant(r, m):
    return flat_metric(r, m, contravariant=True)


def kerr_metric_contravariant(a, r, m):
    r = _f64(r).ravel()
    n = r.size
    a, m = _f64(a, n), _f64(m, n)
    out = np.zeros(n, dtype=METRIC_DTYPE)
    _check(_lib.sim5gpu_kerr_metric_contravariant(SZ(n), _p(a), _p(r), _p(m), _p(out)), "sim5gpu_kerr_metric_contravariant")
    return out


def kerr_newman_metric(a, Q, r, m, contravariant=False):
    r = _f64(r).ravel()
    n = r.size
    a, Q, m = _f64(a, n), _f64(Q, n), _f64(m, n)
    out = np.zeros(n, dtype=METRIC_DTYPE)
    fn = _lib.sim5gpu_kerr_newman_metric_contravariant if contravariant else _lib.sim5gpu_kerr_newman_metric
    _check(fn(SZ(n), _p(a), _p(Q), _p(r), _p(m), _p(out)), "sim5gpu_kerr_newman_metric")
    return out


def kerr_newman_metric_contravariant(a, Q, r, m):
    return kerr_newman_metric(a, Q, r, m, contravariant=True)


def kerr_newman_connection(a, Q, r, m):
    r = _f64(r).ravel()
    n = r.size
    a, Q, m = _f64(a, n), _f64(Q, n), _f64(m, n)
    G = np.zeros((n, 4, 4, 4))
    _check(_lib.sim5gpu_kerr_newman_connection(SZ(n), _p(a), _p(Q), _p(r), _p(m), _p(G)), "sim5gpu_kerr_newman_connection")
    return G


def flat_connection(r, m):
    r = _f64(r).ravel()
    n = r.size
    m = _f64(m, n)
    G = np.zeros((n, 4, 4, 4))
    _check(_lib.sim5gpu_flat_connection(SZ(n), _p(r), _p(m), _p(G)), "sim5gpu_flat_connection")
    return G


def Gamma(G, U, V):
    G = _f64(G).reshape(-1, 4, 4, 4)
    n = G.shape[0]
    U, V = _f64(U, n, 4), _f64(V, n, 4)
    out = np.zeros((n, 4))
    _check(_lib.sim5gpu_Gamma(SZ(n), _p(G), _p(U), _p(V), _p(out)), "sim5gpu_Gamma")
    return out


def vector_covariant(v, metric=None):
    v = _f64(v).reshape(-1, 4)
    n = v.shape[0]
    metric, mp = _metric_or_none(metric)
    out = np.zeros((n, 4))
    _check(_lib.sim5gpu_vector_covariant(SZ(n), _p(v), _p(out), mp), "sim5gpu_vector_covariant")
    return out


def vector_norm(v, metric=None):
    v = _f64(v).reshape(-1, 4)
    n = v.shape[0]
    metric, mp = _metric_or_none(metric)
    out = np.empty(n)
    _check(_lib.sim5gpu_vector_norm(SZ(n), _p(v), mp, _p(out)), "sim5gpu_vector_norm")
    return out


def vector_3norm(v):
    v = _f64(v).reshape(-1, 4)
    out = np.empty(v.shape[0])
    _check(_lib.sim5gpu_vector_3norm(SZ(v.shape[0]), _p(v), _p(out)), "sim5gpu_vector_3norm")
    return out


def vector_norm_to_null(v, V0, metric=None):
    v = np.array(_f64(v).reshape(-1, 4))
    n = v.shape[0]
    V0 = _f64(V0, n)
    metric, mp = _metric_or_none(metric)
    _check(_lib.sim5gpu_vector_norm_to_null(SZ(n), _p(v), _p(V0), mp), "sim5gpu_vector_norm_to_null")
    return v


def tetrad_general(metric, U):
    metric = np.ascontiguousarray(metric, dtype=METRIC_DTYPE).ravel()
    n = metric.size
    U = _f64(U, n, 4)
    t = np.zeros(n, dtype=TETRAD_DTYPE)
    _check(_lib.sim5gpu_tetrad_general(SZ(n), _p(metric), _p(U), _p(t)), "sim5gpu_tetrad_general")
    return t


def tetrad_radial(metric, v_r):
    metric = np.ascontiguousarray(metric, dtype=METRIC_DTYPE).ravel()
    n = metric.size
    v_r = _f64(v_r, n)
    t = np.zeros(n, dtype=TETRAD_DTYPE)
    _check(_lib.sim5gpu_tetrad_radial(SZ(n), _p(metric), _p(v_r), _p(t)), "sim5gpu_tetrad_radial")
    return t


def _ra(fn, name, r, a):
    r = _f64(r).ravel()
    a = _f64(a, r.size)
    out = np.empty(r.size)
    _check(fn(SZ(r.size), _p(r), _p(a), _p(out)), name)
    return out


def omega_r(r, a):
    return _ra(_lib.sim5gpu_omega_r, "sim5gpu_omega_r", r, a)


def omega_z(r, a):
    return _ra(_lib.sim5gpu_omega_z, "sim5gpu_omega_z", r, a)


def ell_from_Omega(Omega, metric):
    metric = np.ascontiguousarray(metric, dtype=METRIC_DTYPE).ravel()
    Omega = _f64(Omega, metric.size)
    out = np.empty(metric.size)
    _check(_lib.sim5gpu_ell_from_Omega(SZ(metric.size), _p(Omega), _p(metric), _p(out)), "sim5gpu_ell_from_Omega")
    return out


def fourvelocity_zamo(metric):
    metric = np.ascontiguousarray(metric, dtype=METRIC_DTYPE).ravel()
    U = np.zeros((metric.size, 4))
    _check(_lib.sim5gpu_fourvelocity_zamo(SZ(metric.size), _p(metric), _p(U)), "sim5gpu_fourvelocity_zamo")
    return U


def _fourvel1(fn, name, x, metric):
    metric = np.ascontiguousarray(metric, dtype=METRIC_DTYPE).ravel()
    x = _f64(x, metric.size)
    U = np.zeros((metric.size, 4))
    _check(fn(SZ(metric.size), _p(x), _p(metric), _p(U)), name)
    return U


def fourvelocity_azimuthal(Omega, metric):
    return _fourvel1(_lib.sim5gpu_fourvelocity_azimuthal, "sim5gpu_fourvelocity_azimuthal", Omega, metric)


def fourvelocity_radial(vr, metric):
    return _fourvel1(_lib.sim5gpu_fourvelocity_radial, "sim5gpu_fourvelocity_radial", vr, metric)


def fourvelocity_norm(U1, U2, U3, metric):
    metric = np.ascontiguousarray(metric, dtype=METRIC_DTYPE).ravel()
    n = metric.size
    U1, U2, U3 = _f64(U1, n), _f64(U2, n), _f64(U3, n)
    out = np.empty(n)
    _check(_lib.sim5gpu_fourvelocity_norm(SZ(n), _p(U1), _p(U2), _p(U3), _p(metric), _p(out)), "sim5gpu_fourvelocity_norm")
    return out


def fourvelocity(U1, U2, U3, metric):
    metric = np.ascontiguousarray(metric, dtype=METRIC_DTYPE).ravel()
    n = metric.size
    U1, U2, U3 = _f64(U1, n), _f64(U2, n), _f64(U3, n)
    U = np.zeros((n, 4))
    _check(_lib.sim5gpu_fourvelocity(SZ(n), _p(U1), _p(U2), _p(U3), _p(metric), _p(U)), "sim5gpu_fourvelocity")
    return U


def geodesic_position_pol_sign_k_theta(g, P):
    return _geod_P(_lib.sim5gpu_geodesic_position_pol_sign_k_theta, "sim5gpu_geodesic_position_pol_sign_k_theta", g, P)


LEGENDRE = {"elliptic_f": 0, "elliptic_e_sin": 1, "elliptic_pi_sin": 2, "elliptic_pi": 3}


def legendre(name, x, m, nn=None):
    """elliptic_f(phi, m), elliptic_e_sin(s, m), elliptic_pi_sin(s, nn, m) -> float64[n]; elliptic_pi(phi, nn, m) ->
    complex128[n]"""
    which = LEGENDRE[name]
    x = _f64(x).ravel()
    n = x.size
    m = _f64(m, n)
    nn = _f64(nn, n) if nn is not None else None
    out = np.empty(2 * n if which == 3 else n)
    _check(_lib.sim5gpu_legendre(I(which), SZ(n), _p(x), _p(nn) if nn is not None else None, _p(m), _p(out)),
           "sim5gpu_legendre(%s)" % name)
    return out.view(np.complex128) if which == 3 else out


def blackbody(T, hardf, cos_mu, E):
    """spectrum of ONE black body over the energies E [keV] (the reference's array form)"""
    E = _f64(E).ravel()
    out = np.zeros(E.size)
    _check(_lib.sim5gpu_blackbody(D(T), D(hardf), D(cos_mu), SZ(E.size), _p(E), _p(out)), "sim5gpu_blackbody")
    return out


def blackbody_photons(T, hardf, cos_mu, E):
    E = _f64(E).ravel()
    n = E.size
    T, hardf, cos_mu = _f64(T, n), _f64(hardf, n), _f64(cos_mu, n)
    out = np.empty(n)
    _check(_lib.sim5gpu_blackbody_photons(SZ(n), _p(T), _p(hardf), _p(cos_mu), _p(E), _p(out)), "sim5gpu_blackbody_photons")
    return out


def blackbody_photons_total(T, hardf):
    T = _f64(T).ravel()
    hardf = _f64(hardf, T.size)
    out = np.empty(T.size)
    _check(_lib.sim5gpu_blackbody_photons_total(SZ(T.size), _p(T), _p(hardf), _p(out)), "sim5gpu_blackbody_photons_total")
    return out


# ---- whole-job kernels --------------------------------------------------------------------------
IMG_DEFAULT, IMG_STRICT, IMG_MIRROR, IMG_INPLACE, IMG_DIRECT = 0, 1, 2, 4, 8


def image_desc(nx, ny, a, incl_rad, y0=0, y1=None, rmax=0.0, rms=0.0, bh_mass=10.0, mdot=0.1,
               alpha_visc=0.1, max_order=2, pol_degree=0.0, strict=False, stripe_rows=0, stripe_step=0,
               disk_spin=-1.0, mirror=False, inplace=False, direct=False):
    """Job description; defaults are those of the reference example (disk-image.c:41-45).
    strict=True selects the reference-parameter arithmetic variant (SIM5GPU_IMG_STRICT); mirror=True adds the mirror
    images ny-1-y of the named rows, which must lie in the upper half (SIM5GPU_IMG_MIRROR); inplace=True: the outputs are
    whole-image planes and every traced row is written at its image row (SIM5GPU_IMG_INPLACE); direct=True: every ray through
    the reference's own sequence with the fast arithmetic (SIM5GPU_IMG_DIRECT)."""
    return ImageDesc(nx=nx, ny=ny, y0=y0, y1=ny if y1 is None else y1, a=a, incl=incl_rad,
                     rmax=rmax, rms=rms, bh_mass=bh_mass, mdot=mdot, alpha_visc=alpha_visc,
                     max_order=max_order, flags=(IMG_STRICT if strict else IMG_DEFAULT) | (IMG_MIRROR if mirror else 0) | (IMG_INPLACE if inplace else 0) | (IMG_DIRECT if direct else 0),
                     pol_degree=pol_degree,
                     stripe_rows=stripe_rows, stripe_step=stripe_step, disk_spin=disk_spin)


def image_rows(desc):
    _lib.sim5gpu_image_rows.restype = I
    return _lib.sim5gpu_image_rows(C.byref(desc))


def image_row_map(desc):
    """image row of every packed output row of the job (host arithmetic of the library, no GPU)"""
    n = image_rows(desc)
    rows = np.zeros(max(n, 1), dtype=np.int32)
    _check(_lib.sim5gpu_image_row_map(C.byref(desc), _p(rows), I(n)), "sim5gpu_image_row_map")
    return rows[:n]


def image_place_shares(descs, d_shares, share_rows, d_image_f, d_image_g, stream=None):
    """rows of len(descs) shares (consecutive [2][share_rows][nx] float blocks at d_shares) to their image rows, one launch"""
    arr = (ImageDesc * len(descs))(*descs)
    _check(_lib.sim5gpu_image_place_shares(I(len(descs)), arr, VP(d_shares), SZ(share_rows), VP(d_image_f), VP(d_image_g),
                                           VP(stream or 0)), "sim5gpu_image_place_shares")


def disk_image_device(desc, d_image_f, d_image_g, aux=None, stream=None):
    """Asynchronous launch on device pointers (ints).  aux: dict of device pointers or None."""
    a = None
    if aux:
        a = ImageAux(**{k: aux.get(k) for k in ("cls", "gtype", "r", "g", "flux")})
    _check(_lib.sim5gpu_disk_image(C.byref(desc), VP(d_image_f), VP(d_image_g),
                                   C.byref(a) if a is not None else None, VP(stream or 0)),
           "sim5gpu_disk_image")


def disk_image_jobs(descs, d_images_f, d_images_g, stream=None):
    """Several image jobs with as few launches as they allow (sim5gpu_disk_image_jobs): descs[j] into the device planes
    d_images_f[j], d_images_g[j] (ints).  Asynchronous on `stream`."""
    n = len(descs)
    arr = (ImageDesc * n)(*descs)
    pf = (VP * n)(*[VP(int(x)) for x in d_images_f])
    pg = (VP * n)(*[VP(int(x)) for x in d_images_g])
    _check(_lib.sim5gpu_disk_image_jobs(I(n), arr, pf, pg, VP(stream or 0)), "sim5gpu_disk_image_jobs")


def disk_image(desc, full=False):
    """Host-buffer convenience: returns dict(image_f, image_g[, cls, gtype, r, g, flux])."""
    rows, nx = max(image_rows(desc), 0), max(desc.nx, 0)       # bad geometry is rejected by the library
    out = {"image_f": np.zeros((rows, nx), np.float32), "image_g": np.zeros((rows, nx), np.float32)}
    aux = None
    if full:
        out.update(cls=np.zeros((rows, nx), np.uint8), gtype=np.zeros((rows, nx), np.int8),
                   r=np.zeros((rows, nx)), g=np.zeros((rows, nx)), flux=np.zeros((rows, nx)))
        aux = ImageAux(cls=out["cls"].ctypes.data, gtype=out["gtype"].ctypes.data, r=out["r"].ctypes.data,
                       g=out["g"].ctypes.data, flux=out["flux"].ctypes.data)
    _check(_lib.sim5gpu_disk_image_host(C.byref(desc), _p(out["image_f"]), _p(out["image_g"]),
                                        C.byref(aux) if aux is not None else None), "sim5gpu_disk_image_host")
    return out


def disk_rays_device(desc, n, d_alpha, d_beta, d_image_f, d_image_g, aux=None, stream=None):
    a = None
    if aux:
        a = ImageAux(**{k: aux.get(k) for k in ("cls", "gtype", "r", "g", "flux")})
    _check(_lib.sim5gpu_disk_rays(C.byref(desc), SZ(n), VP(d_alpha), VP(d_beta), VP(d_image_f), VP(d_image_g),
                                  C.byref(a) if a is not None else None, VP(stream or 0)), "sim5gpu_disk_rays")


def disk_image_polarized_device(desc, d_stokes, d_chi=None, aux=None, stream=None):
    a = None
    if aux:
        a = ImageAux(**{k: aux.get(k) for k in ("cls", "gtype", "r", "g", "flux")})
    _check(_lib.sim5gpu_disk_image_polarized(C.byref(desc), VP(d_stokes), VP(d_chi or 0),
                                             C.byref(a) if a is not None else None, VP(stream or 0)),
           "sim5gpu_disk_image_polarized")


def disk_spectrum(desc, energies, hardening=1.7, limb_darkening=1):
    """Sum over the pixels of `desc` of I_nu(E/g) g^3 for each energy [keV]; host arrays in and out."""
    E = np.ascontiguousarray(energies, dtype=np.float64).ravel()
    _lib.sim5gpu_disk_spectrum_workspace.restype = SZ
    ws_bytes = _lib.sim5gpu_disk_spectrum_workspace(C.byref(desc), I(E.size))
    dE = DeviceBuffer(E.nbytes); dS = DeviceBuffer(E.nbytes); ws = DeviceBuffer(max(ws_bytes, 8))
    dE.from_numpy(E)
    _check(_lib.sim5gpu_disk_spectrum(C.byref(desc), I(E.size), VP(dE.ptr), D(hardening), I(limb_darkening),
                                      VP(dS.ptr), VP(ws.ptr), VP(0)), "sim5gpu_disk_spectrum")
    synchronize()
    return dS.to_numpy(np.float64, (E.size,))


SURFACE_TABLE_CHECKED = 2


def _surface_flags(tR, strict, checked):
    """bit 0: strict arithmetic; SURFACE_TABLE_CHECKED when the caller vouches for the table (checked=True) -- the library
    then skips its blocking read-back of the table.  checked=None: vouch for it if this host copy passes the same test."""
    if checked is None:
        checked = bool(tR.size >= 2 and np.all(np.diff(tR) > 0) and not np.isnan(tR[0]))
    return (1 if strict else 0) | (SURFACE_TABLE_CHECKED if checked else 0)


def disk_surface_rays(a, incl_rad, table_R, table_H, alpha, beta, strict=False, checked=False):
    """Surface search for a thick disk H(R) (host arrays in and out): dict(P, r, m, k[n,4], status)."""
    tR = np.ascontiguousarray(table_R, dtype=np.float64).ravel()
    tH = np.ascontiguousarray(table_H, dtype=np.float64).ravel()
    al = np.ascontiguousarray(alpha, dtype=np.float64).ravel()
    be = np.ascontiguousarray(beta, dtype=np.float64).ravel()
    n = al.size
    bufs = {}
    for name, arr in (("tR", tR), ("tH", tH), ("al", al), ("be", be)):
        bufs[name] = DeviceBuffer(max(arr.nbytes, 8)); bufs[name].from_numpy(arr)
    out = {k: DeviceBuffer(max(n * s, 8)) for k, s in (("P", 8), ("r", 8), ("m", 8), ("k", 32), ("st", 4))}
    _check(_lib.sim5gpu_disk_surface_rays(D(a), D(incl_rad), I(tR.size), VP(bufs["tR"].ptr), VP(bufs["tH"].ptr),
                                          SZ(n), VP(bufs["al"].ptr), VP(bufs["be"].ptr), VP(out["P"].ptr),
                                          VP(out["r"].ptr), VP(out["m"].ptr), VP(out["k"].ptr), VP(out["st"].ptr),
                                          I(_surface_flags(tR, strict, checked)), VP(0)), "sim5gpu_disk_surface_rays")
    synchronize()
    return {"P": out["P"].to_numpy(np.float64, (n,)), "r": out["r"].to_numpy(np.float64, (n,)),
            "m": out["m"].to_numpy(np.float64, (n,)), "k": out["k"].to_numpy(np.float64, (n, 4)),
            "status": out["st"].to_numpy(np.int32, (n,))}


def disk_surface_frame(a, incl_rad, bh_mass, mdot, table_R, table_H, alpha, beta, table_vr=None, disk_spin=-1.0,
                       strict=False, checked=False):
    """Surface search + local frame in one kernel (host arrays in and out):
    dict(P, r, m, k[n,4], status, g, mue, flux)."""
    tR = np.ascontiguousarray(table_R, dtype=np.float64).ravel()
    tH = np.ascontiguousarray(table_H, dtype=np.float64).ravel()
    al = np.ascontiguousarray(alpha, dtype=np.float64).ravel()
    be = np.ascontiguousarray(beta, dtype=np.float64).ravel()
    n = al.size
    arrays = [("tR", tR), ("tH", tH), ("al", al), ("be", be)]
    if table_vr is not None:
        tV = np.ascontiguousarray(table_vr, dtype=np.float64).ravel()
        if tV.size != tR.size:
            raise ValueError("table_vr must have the length of table_R")
        arrays.append(("tV", tV))
    bufs = {}
    for name, arr in arrays:
        bufs[name] = DeviceBuffer(max(arr.nbytes, 8)); bufs[name].from_numpy(arr)
    out = {k: DeviceBuffer(max(n * s, 8)) for k, s in (("P", 8), ("r", 8), ("m", 8), ("k", 32), ("st", 4),
                                                       ("g", 8), ("mue", 8), ("flux", 8))}
    _check(_lib.sim5gpu_disk_surface_frame(D(a), D(incl_rad), D(bh_mass), D(mdot), D(disk_spin), I(tR.size),
                                           VP(bufs["tR"].ptr), VP(bufs["tH"].ptr),
                                           VP(bufs["tV"].ptr if "tV" in bufs else 0), SZ(n),
                                           VP(bufs["al"].ptr), VP(bufs["be"].ptr), VP(out["P"].ptr), VP(out["r"].ptr),
                                           VP(out["m"].ptr), VP(out["k"].ptr), VP(out["st"].ptr), VP(out["g"].ptr),
                                           VP(out["mue"].ptr), VP(out["flux"].ptr), I(_surface_flags(tR, strict, checked)), VP(0)),
           "sim5gpu_disk_surface_frame")
    synchronize()
    res = {"P": out["P"].to_numpy(np.float64, (n,)), "r": out["r"].to_numpy(np.float64, (n,)),
           "m": out["m"].to_numpy(np.float64, (n,)), "k": out["k"].to_numpy(np.float64, (n, 4)),
           "status": out["st"].to_numpy(np.int32, (n,))}
    for k in ("g", "mue", "flux"):
        res[k] = out[k].to_numpy(np.float64, (n,))
    return res


def release_workspaces():
    """give back the grow-only workspaces of the surface-search and torus jobs; returns the bytes freed"""
    n = SZ(0)
    _check(_lib.sim5gpu_release_workspaces(C.byref(n)), "sim5gpu_release_workspaces")
    return int(n.value)


def torus_image_device(desc, d_stokes, aux=None, stream=None):
    a = None
    if aux:
        a = TorusAux(**{k: aux.get(k) for k in ("steps", "max_step_error", "carter_error", "x_end", "k_end")})
    _check(_lib.sim5gpu_torus_image(C.byref(desc), VP(d_stokes), C.byref(a) if a is not None else None,
                                    VP(stream or 0)), "sim5gpu_torus_image")
