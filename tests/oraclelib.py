"""ctypes bindings for the CPU checker libraries (TEST INFRASTRUCTURE ONLY).

* ``oracle/liboracle.so``      -- our C restatement (symbols ``orc_*``)
* ``oracle/_ref/libsim5ref.so`` -- the unmodified reference, when it has been built
  (``make -C oracle``; needs /root/reference, so only in the build container -- the
  prebuilt file travels to the GPU box).

Struct layouts are shared by both libraries (reference src/sim5kerr-geod.h:42-68,
src/sim5kerr.h:18-31, src/sim5raytrace.h:26-43).
"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
DRIVER_SO = os.path.join(ORACLE_DIR, "libcpudriver.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libsim5ref.so")

D = C.c_double
I = C.c_int
D4 = D * 4


class Cplx(C.Structure):
    _fields_ = [("re", D), ("im", D)]


class Geodesic(C.Structure):
    _fields_ = [("a", D), ("alpha", D), ("beta", D), ("incl", D), ("cos_i", D),
                ("l", D), ("q", D),
                ("r1", Cplx), ("r2", Cplx), ("r3", Cplx), ("r4", Cplx),
                ("nrr", I), ("type", I),
                ("m2p", D), ("m2m", D), ("mm", D), ("mK", D),
                ("rp", D), ("dmdp_inf", D),
                ("Rpc", D), ("Tpp", D), ("Tip", D),
                ("k", D4), ("p", D)]


class Metric(C.Structure):
    _fields_ = [("a", D), ("r", D), ("m", D), ("g00", D), ("g11", D), ("g22", D),
                ("g33", D), ("g03", D)]


class Tetrad(C.Structure):
    _fields_ = [("e", D4 * 4), ("metric", Metric)]


class RaytraceData(C.Structure):
    _fields_ = [("opt_gr", I), ("opt_pol", I), ("step_epsilon", D),
                ("bh_spin", D), ("E", D), ("Q", D), ("WP", Cplx),
                ("pass_", I), ("refines", I), ("dk", D4), ("df", D4),
                ("kt", D), ("error", C.c_float)]


class DiskNT(C.Structure):
    _fields_ = [("mass", C.c_float), ("spin", C.c_float), ("mdot", C.c_float),
                ("rms", C.c_float), ("alpha", C.c_float), ("options", I)]


class Pixel(C.Structure):
    _fields_ = [("cls", I), ("gtype", I), ("err", I), ("r", D), ("g", D), ("flux", D),
                ("image_f", C.c_float), ("image_g", C.c_float)]


assert C.sizeof(Geodesic) == 240 and C.sizeof(Metric) == 64
assert C.sizeof(Tetrad) == 192 and C.sizeof(RaytraceData) == 144

G444 = (D * 4) * 4 * 4
PG = C.POINTER(Geodesic)
PM = C.POINTER(Metric)
PT = C.POINTER(Tetrad)
PR = C.POINTER(RaytraceData)
PD = C.POINTER(D)
PI = C.POINTER(I)

# name -> (restype, argtypes); names are the SIM5 ones, the oracle adds the orc_ prefix
_COMMON = {
    "rf": (D, [D, D, D]), "rd": (D, [D, D, D]), "rc": (D, [D, D]), "rj": (D, [D, D, D, D]),
    "elliptic_k": (D, [D]), "elliptic_f_sin": (D, [D, D]),
    "jacobi_isn": (D, [D, D]), "jacobi_icn": (D, [D, D]), "jacobi_itn": (D, [D, D]),
    "jacobi_sncndn": (None, [D, D, PD, PD, PD]),
    "jacobi_sn": (D, [D, D]), "jacobi_cn": (D, [D, D]), "jacobi_dn": (D, [D, D]),
    "r_bh": (D, [D]), "r_ms": (D, [D]), "r_mb": (D, [D]), "r_ph": (D, [D]),
    "flat_metric": (None, [D, D, PM]),
    "kerr_metric": (None, [D, D, D, PM]),
    "kerr_metric_contravariant": (None, [D, D, D, PM]),
    "kerr_newman_metric": (None, [D, D, D, D, PM]),
    "kerr_newman_metric_contravariant": (None, [D, D, D, D, PM]),
    "kerr_newman_connection": (None, [D, D, D, D, G444]),
    "flat_connection": (None, [D, D, G444]),
    "kerr_connection": (None, [D, D, D, G444]),
    "Gamma": (None, [G444, D4, D4, D4]),
    "dotprod": (D, [D4, D4, PM]),
    "vector_norm_to": (None, [D4, D, PM]),
    "tetrad_zamo": (None, [PM, PT]),
    "tetrad_azimuthal": (None, [PM, D, PT]),
    "tetrad_surface": (None, [PM, D, D, D, PT]),
    "bl2on": (None, [D4, D4, PT]), "on2bl": (None, [D4, D4, PT]),
    "OmegaK": (D, [D, D]), "ellK": (D, [D, D]),
    "Omega_from_ell": (D, [D, PM]),
    "gfactorK": (D, [D, D, D]),
    "photon_momentum": (None, [D, D, D, D, D, D, D, D4]),
    "photon_motion_constants": (None, [D, D, D, D4, PD, PD]),
    "photon_carter_const": (D, [D4, PM]),
    "geodesic_init_inf": (I, [D, D, D, D, PG, PI]),
    "geodesic_init_src": (I, [D, D, D, D4, I, PG, PI]),
    "geodesic_P_int": (D, [PG, D, I]),
    "geodesic_position_rad": (D, [PG, D]),
    "geodesic_position_pol": (D, [PG, D]),
    "geodesic_dm_sign": (D, [PG, D]),
    "geodesic_momentum": (None, [PG, D, D, D, D4]),
    "geodesic_find_midplane_crossing": (D, [PG, I]),
    "geodesic_follow": (None, [PG, D, PD, PD, PD, PI]),
    "raytrace_prepare": (None, [D, D4, D4, D, I, PR]),
    "raytrace": (None, [D4, D4, PD, PR]),
    "raytrace_error": (D, [D4, D4, PR]),
    "polarization_constant": (Cplx, [D4, D4, PM]),
    "polarization_vector": (None, [D4, Cplx, PM, D4]),
    "polarization_constant_infinity": (Cplx, [D, D, D, D]),
    "polarization_angle_rotation": (D, [D, D, D, D, Cplx]),
    "blackbody_Iv": (D, [D, D, D, D]),
    # azimuth / light-travel time and the integrals below them (ref src/sim5elliptic.c:255-1161).
    # A `sim5complex` (double _Complex) argument passed by value travels as two consecutive doubles
    # (re, im) in the SysV x86-64 calling convention, which is how it is declared here.
    "elliptic_f_cos": (D, [D, D]), "elliptic_e_cos": (D, [D, D]),
    "elliptic_pi_complete": (D, [D, D]), "elliptic_pi_cos": (D, [D, D, D]),
    "integral_C2": (D, [D, D]), "integral_C2_cos": (D, [D, D]),
    "integral_Z1": (D, [D, D, D, D]), "integral_Z2": (D, [D, D, D, D]),
    "integral_Rm1": (D, [D, D, D]), "integral_Rm2": (D, [D, D, D]),
    "integral_R1": (D, [D, D, D]), "integral_R2": (D, [D, D, D]),
    "integral_R_r0_re": (D, [D] * 5), "integral_R_r0_re_inf": (D, [D] * 4),
    "integral_R_r1_re": (D, [D] * 5), "integral_R_r2_re": (D, [D] * 5),
    "integral_R_rp_re": (D, [D] * 6), "integral_R_rp_re_inf": (D, [D] * 5),
    "integral_R_r0_cc": (D, [D] * 5), "integral_R_r0_cc_inf": (D, [D] * 4),
    "integral_R_r1_cc": (D, [D] * 6), "integral_R_r2_cc": (D, [D] * 6),
    "integral_R_rp_cc2": (D, [D] * 7), "integral_R_rp_cc2_inf": (D, [D] * 6),
    "integral_T_m0": (D, [D] * 3), "integral_T_m2": (D, [D] * 3), "integral_T_mp": (D, [D] * 4),
    "geodesic_position_azm": (D, [PG, D, D, D]),
    "geodesic_timedelay": (D, [PG, D, D, D, D, D, D]),
}


def build_oracle():
    """(Re)build the checker libraries; building the checker is not using it."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


class _Lib:
    def __init__(self, path, prefix):
        self.path = path
        self.lib = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0))
        for name, (res, args) in _COMMON.items():
            fn = getattr(self.lib, prefix + name)
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)


class Oracle(_Lib):
    """Our restatement; the disk keeps its state in an explicit object."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        super().__init__(ORACLE_SO, "orc_")
        L = self.lib
        PDN = C.POINTER(DiskNT)
        L.orc_disk_nt_setup.argtypes = [PDN, D, D, D, D]
        L.orc_disk_nt_setup.restype = None
        L.orc_disk_nt_r_min.argtypes = [PDN]
        L.orc_disk_nt_r_min.restype = D
        L.orc_disk_nt_flux.argtypes = [PDN, D]
        L.orc_disk_nt_flux.restype = D
        L.orc_disk_nt_ell.argtypes = [PDN, D]
        L.orc_disk_nt_ell.restype = D
        L.orc_disk_pixel.argtypes = [PDN, D, D, D, D, D, C.POINTER(Pixel)]
        L.orc_disk_pixel.restype = None
        L.orc_disk_nt_setup_opt.argtypes = [PDN, D, D, D, D, I]
        L.orc_disk_nt_setup_opt.restype = None
        for name in ("orc_disk_nt_mdot", "orc_disk_nt_lumi"):
            getattr(L, name).argtypes = [PDN]
            getattr(L, name).restype = D
        L.orc_disk_nt_sigma.argtypes = [PDN, D]
        L.orc_disk_nt_sigma.restype = D
        self.disk = DiskNT()

    def disk_nt_setup(self, M, a, mdot_or_L, alpha, options=0):
        self.lib.orc_disk_nt_setup_opt(C.byref(self.disk), M, a, mdot_or_L, alpha, options)
        return 0

    def disk_nt_mdot(self):
        return self.lib.orc_disk_nt_mdot(C.byref(self.disk))

    def disk_nt_lumi(self):
        return self.lib.orc_disk_nt_lumi(C.byref(self.disk))

    def disk_nt_sigma(self, r):
        return self.lib.orc_disk_nt_sigma(C.byref(self.disk), r)

    def disk_nt_r_min(self):
        return self.lib.orc_disk_nt_r_min(C.byref(self.disk))

    def disk_nt_flux(self, r):
        return self.lib.orc_disk_nt_flux(C.byref(self.disk), r)

    def disk_nt_ell(self, r):
        return self.lib.orc_disk_nt_ell(C.byref(self.disk), r)

    def disk_pixel(self, inc, a, rms, alpha, beta):
        px = Pixel()
        self.lib.orc_disk_pixel(C.byref(self.disk), inc, a, rms, alpha, beta, C.byref(px))
        return px


class Reference(_Lib):
    """The unmodified reference library (process-global disk state, as shipped)."""

    def __init__(self):
        super().__init__(REF_SO, "")
        L = self.lib
        L.disk_nt_setup.argtypes = [D, D, D, D, I]
        L.disk_nt_setup.restype = I
        L.disk_nt_r_min.restype = D
        L.disk_nt_flux.argtypes = [D]
        L.disk_nt_flux.restype = D
        L.disk_nt_ell.argtypes = [D]
        L.disk_nt_ell.restype = D
        L.disk_nt_mdot.restype = D
        L.disk_nt_lumi.restype = D
        L.disk_nt_sigma.argtypes = [D]
        L.disk_nt_sigma.restype = D
        self.disk_nt_setup = L.disk_nt_setup
        self.disk_nt_r_min = L.disk_nt_r_min
        self.disk_nt_flux = L.disk_nt_flux
        self.disk_nt_ell = L.disk_nt_ell
        self.disk_nt_mdot = L.disk_nt_mdot
        self.disk_nt_lumi = L.disk_nt_lumi
        self.disk_nt_sigma = L.disk_nt_sigma


def have_reference():
    return os.path.exists(REF_SO)


def struct_bytes(s):
    return bytes(memoryview(s))


def cpu_disk_image(kind, nx, ny, a, inc_deg, y0=0, y1=None, ystride=1, xstride=1, nthreads=1,
                   full=True, M=10.0, mdot=0.1, alpha_visc=0.1):
    """Trace (a sample of) the thin-disk image on the host cores.

    kind = "reference" (oracle/_ref) or "port" (oracle/liboracle.so).  Returns a dict of
    packed numpy arrays over the sampled pixels plus the wall time of the pixel loop.
    Pixel loop and parameters: reference examples/04-disk-image-eqplane/disk-image.c:41-105.
    """
    import math
    import numpy as np
    if not os.path.exists(DRIVER_SO) or not os.path.exists(ORACLE_SO):
        build_oracle()
    drv = C.CDLL(DRIVER_SO)
    fn = drv.cpu_disk_image
    VP = C.c_void_p
    fn.argtypes = [C.c_char_p, I, I, I, D, D, D, D, D, I, I, I, I, I,
                   VP, VP, VP, VP, VP, VP, VP, PD]
    fn.restype = I
    y1 = ny if y1 is None else y1
    path = {"reference": REF_SO, "port": ORACLE_SO}[kind]
    onx = (nx + xstride - 1) // xstride
    ony = (y1 - y0 + ystride - 1) // ystride
    out = {
        "image_f": np.zeros((ony, onx), np.float32),
        "image_g": np.zeros((ony, onx), np.float32),
        "cls": np.zeros((ony, onx), np.uint8),
    }
    if full:
        out["gtype"] = np.zeros((ony, onx), np.int8)
        out["r"] = np.zeros((ony, onx), np.float64)
        out["g"] = np.zeros((ony, onx), np.float64)
        out["flux"] = np.zeros((ony, onx), np.float64)

    def p(name):
        return out[name].ctypes.data if name in out else None

    sec = D(0.0)
    rc = fn(path.encode(), 0 if kind == "reference" else 1, nx, ny, a, inc_deg / 180.0 * math.pi,
            M, mdot, alpha_visc, y0, y1, ystride, xstride, nthreads,
            p("image_f"), p("image_g"), p("cls"), p("gtype"), p("r"), p("g"), p("flux"),
            C.byref(sec))
    if rc != 0:
        raise RuntimeError("cpu_disk_image failed rc=%d (%s)" % (rc, path))
    out["seconds"] = sec.value
    out["rays"] = onx * ony
    return out


def cpu_disk_flux(r, a, kind=None, M=10.0, mdot=0.1, alpha_visc=0.1):
    """disk_nt_flux of the live reference (kind "reference"; default when oracle/_ref is there) or of our restatement
    ("port", pinned to it byte for byte) at the radii `r` (any shape, NaNs pass through), for the disk model
    disk_nt_setup(M, a, mdot, alpha_visc, 0) -- the reference's flux AT GIVEN INPUT BITS (ref src/sim5disk-nt.c:110-146)."""
    import numpy as np
    if not os.path.exists(DRIVER_SO) or not os.path.exists(ORACLE_SO):
        build_oracle()
    kind = kind or ("reference" if have_reference() else "port")
    drv = C.CDLL(DRIVER_SO)
    fn = drv.cpu_disk_flux
    fn.argtypes = [C.c_char_p, I, D, D, D, D, C.c_long, C.c_void_p, C.c_void_p]
    fn.restype = I
    rr = np.ascontiguousarray(np.asarray(r, np.float64)).ravel()
    nan = np.isnan(rr)
    rin = np.where(nan, 1e3, rr)
    out = np.zeros_like(rin)
    rc = fn({"reference": REF_SO, "port": ORACLE_SO}[kind].encode(), 0 if kind == "reference" else 1, M, a, mdot, alpha_visc,
            rin.size, rin.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("cpu_disk_flux failed rc=%d" % rc)
    out[nan] = np.nan
    return out.reshape(np.shape(r))
