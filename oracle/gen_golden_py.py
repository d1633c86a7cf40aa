#!/usr/bin/env python3
"""Golden vectors of the reference's PYTHON ray tracer (python/sim5diskraytrace.py), captured by
importing it in the build container with a throw-away `sim5lib` module made of ctypes calls into the
unmodified reference build (oracle/_ref/libsim5ref.so) -- the reference's own SWIG module cannot be
built here (no swig).  Output: tests/golden/py_diskraytrace.npz (inputs + the reference's outputs).

TEST INFRASTRUCTURE ONLY; needs /root/reference; nothing of the reference is copied.
"""
import ctypes as C
import logging
import math
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oraclelib as ol  # noqa: E402

REFPY = "/root/reference/python"


def make_shim(ref):
    m = types.ModuleType("sim5lib")

    class intp:
        def __init__(self): self.c = C.c_int(0)
        def value(self): return self.c.value
        def assign(self, v): self.c.value = int(v)

    class doublep:
        def __init__(self): self.c = C.c_double(0.0)
        def value(self): return self.c.value
        def assign(self, v): self.c.value = float(v)

    m.intp, m.doublep = intp, doublep
    class geodesic(ol.Geodesic):
        # the reference's python code reads gd.i (python/sim5diskraytrace.py:265); the C struct member is incl
        @property
        def i(self):
            return self.incl
    m.geodesic = geodesic
    m.sim5metric = ol.Metric
    m.sim5tetrad = ol.Tetrad
    m.doubleArray = lambda n: (C.c_double * n)()
    m.double_array_getitem = lambda a, i: a[i]

    def sim5vector(c):
        v = (C.c_double * 4)(); v[0], v[1], v[2], v[3] = c; return v
    m.sim5vector = sim5vector
    m.grav_radius, m.parsec, m.solar_mass, m.grav_const = 1.476716e+05, 3.085680e+18, 1.988920e+33, 6.673000e-08
    m.Mdot_Edd = 2.225475942e+18
    L = ref.lib
    for name, res, args in [("disk_nt_mdot", C.c_double, []), ("disk_nt_lumi", C.c_double, []),
                            ("disk_nt_sigma", C.c_double, [C.c_double])]:
        f = getattr(L, name); f.restype = res; f.argtypes = args; setattr(m, name, f)
    m.r_bh, m.r_ms = ref.r_bh, ref.r_ms
    m.disk_nt_setup = ref.disk_nt_setup
    m.disk_nt_r_min, m.disk_nt_flux, m.disk_nt_ell = ref.disk_nt_r_min, ref.disk_nt_flux, ref.disk_nt_ell
    m.geodesic_init_inf = lambda i, a, al, be, gd, st: ref.geodesic_init_inf(i, a, al, be, C.byref(gd), C.byref(st.c))
    m.geodesic_find_midplane_crossing = lambda gd, o: ref.geodesic_find_midplane_crossing(C.byref(gd), o)
    m.geodesic_position_rad = lambda gd, P: ref.geodesic_position_rad(C.byref(gd), P)
    m.geodesic_position_pol = lambda gd, P: ref.geodesic_position_pol(C.byref(gd), P)
    m.geodesic_P_int = lambda gd, r, ppc: ref.geodesic_P_int(C.byref(gd), r, ppc)
    m.geodesic_follow = lambda gd, step, P, r, mm, st: ref.geodesic_follow(C.byref(gd), step, C.byref(P.c), C.byref(r.c), C.byref(mm.c), C.byref(st.c))
    m.photon_momentum = lambda a, r, mm, l, q, rs, ms, k: ref.photon_momentum(a, r, mm, l, q, rs, ms, k)
    m.kerr_metric = lambda a, r, mm, met: ref.kerr_metric(a, r, mm, C.byref(met))
    m.tetrad_surface = lambda met, Om, V, dh, t: ref.tetrad_surface(C.byref(met), Om, V, dh, C.byref(t))
    m.Omega_from_ell = lambda ell, met: ref.Omega_from_ell(ell, C.byref(met))
    m.on2bl = lambda vin, vout, t: ref.on2bl(vin, vout, C.byref(t))
    m.dotprod = lambda a, b, met: ref.dotprod(a, b, C.byref(met))
    return m


def main():
    if not (ol.have_reference() and os.path.isdir(REFPY)):
        sys.exit("needs oracle/_ref/libsim5ref.so and /root/reference/python")
    ref = ol.Reference()
    sys.modules["sim5lib"] = make_shim(ref)
    np.float = float                       # the reference predates numpy 1.24 (python/sim5diskraytrace.py:154)
    logging.disable(logging.CRITICAL)
    sys.path.insert(0, REFPY)
    import sim5diskmodel
    import sim5diskraytrace
    import sim5diskspectrum
    devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(2); os.dup2(devnull, 2)
    out = {}
    try:
        cases = [(a, inc) for a in (0.0, 0.9, 0.998) for inc in (60.0, 70.0)]
        out["cases"] = np.array(cases)
        N = 16
        for ci, (a, inc) in enumerate(cases):
            disk = sim5diskmodel.DiskModel_ThinDisk(10.0, a, 0.1, 0.1)
            rt = sim5diskraytrace.DiskRaytrace(10.0, a, 10.0, disk, None)
            rmax = ref.r_ms(a) + 8.0
            img = rt.image(inc, rmax, N)
            for k, v in img.items():
                out["img%d_%s" % (ci, k)] = np.array(v, dtype=np.float64)      # None -> nan
            # per-pixel geodesic(): r, k
            rr = np.full((N, N), np.nan); kk = np.full((N, N, 4), np.nan)
            for y in range(N):
                for x in range(N):
                    al = ((x + .5) / N - 0.5) * 2.0 * rmax; be = ((y + .5) / N - 0.5) * 2.0 * rmax
                    r, m, gd, k = rt.geodesic(math.radians(inc), al, be, flat=True)
                    if gd is not None:
                        rr[y, x] = r; kk[y, x] = [k[0], k[1], k[2], k[3]]
            out["geo%d_r" % ci] = rr; out["geo%d_k" % ci] = kk
            out["rmax%d" % ci] = np.array([rmax])
            # spectrum of the pixel grid with the reference's black-body class (python/sim5diskspectrum.py:54-88),
            # accumulated as DiskRaytrace.spectrum does (python/sim5diskraytrace.py:122-123): sum Iv(E/g) g^3 per pixel
            bb = sim5diskspectrum.DiskSpectrum_BlackBody()
            E = 10.0 ** np.linspace(-1.5, 1.5, 48)
            for (tag, limb, hard) in (("a", 1, 1.7), ("b", 0, 1.0)):
                spec = np.zeros(len(E))
                g_ = np.array(img["gfactor"], dtype=np.float64); T_ = np.array(img["T"], dtype=np.float64)
                mu_ = np.cos(np.radians(np.array(img["mue"], dtype=np.float64)))
                for y in range(N):
                    for x in range(N):
                        if not np.isfinite(g_[y, x]):
                            continue
                        e = mu_[y, x] if limb > 0 else -1.0
                        spec += bb.spectrum(T_[y, x], e, hard, E / g_[y, x]) * g_[y, x] ** 3
                out["spec%d%s" % (ci, tag)] = spec
            out["spec_E"] = E
    finally:
        os.dup2(saved, 2)
    # ---- thick disk: the surface search (python/sim5diskraytrace.py:257-335) with a tabulated H(R) ------
    os.dup2(devnull, 2)
    try:
        tR = 10.0 ** np.linspace(0.0, 3.0, 256)
        tH = np.where(tR > 2.0, 0.25 * (tR - 2.0), 0.0)
        tV = -0.05 / np.sqrt(tR)                   # a slow radial inflow

        class ThickDisk(sim5diskmodel.DiskModel):
            """H(R): linear interpolation of the table, H[0] below it, constant opening angle beyond it
            (the definition sim5gpu_disk_surface_rays uses)."""
            def h(self, R):
                if not (R > tR[0]): return float(tH[0])
                if R >= tR[-1]: return float(tH[-1] * (R / tR[-1]))
                hi = int(np.searchsorted(tR, R, side="left")); lo = hi - 1
                w = (R - tR[lo]) / (tR[hi] - tR[lo])
                return float(tH[lo] + w * (tH[hi] - tH[lo]))

            # the rest of what image() asks of a model (python/sim5diskraytrace.py:176-179, 340-348): slope of
            # the same piecewise-linear surface, Novikov-Thorne flux and angular momentum of the reference
            # library (disk_nt_setup is called per case below), a slow radial inflow
            def dhdr(self, R):
                if not (R > tR[0]): return 0.0
                if R >= tR[-1]: return float(tH[-1] / tR[-1])
                hi = int(np.searchsorted(tR, R, side="left")); lo = hi - 1
                return float((tH[hi] - tH[lo]) / (tR[hi] - tR[lo]))
            def flux(self, R): return ref.disk_nt_flux(R)
            def l(self, R): return ref.disk_nt_ell(R)
            def vr(self, R):                        # nodes tV, interpolated as the surface itself
                if not (R > tR[0]): return float(tV[0])
                if R >= tR[-1]: return float(tV[-1])
                hi = int(np.searchsorted(tR, R, side="left")); lo = hi - 1
                w = (R - tR[lo]) / (tR[hi] - tR[lo])
                return float(tV[lo] + w * (tV[hi] - tV[lo]))

        out["surf_R"] = tR; out["surf_H"] = tH; out["surf_V"] = tV
        scases = [(a, inc) for a in (0.5, 0.9) for inc in (30.0, 60.0, 80.0)]
        out["surf_cases"] = np.array(scases)
        Ns, rmax_s = 12, 30.0
        c = ((np.arange(Ns) + .5) / Ns - 0.5) * 2.0 * rmax_s
        out["surf_alpha"] = np.tile(c, Ns); out["surf_beta"] = np.repeat(c, Ns)
        for ci, (a, inc) in enumerate(scases):
            rt = sim5diskraytrace.DiskRaytrace(10.0, a, 10.0, ThickDisk(), None)
            rr = np.zeros(Ns * Ns); mm = np.zeros(Ns * Ns); ok = np.zeros(Ns * Ns, np.int32); kk = np.full((Ns * Ns, 4), np.nan)
            PP = np.full(Ns * Ns, np.nan)
            for j in range(Ns * Ns):
                r, m, gd, k = rt.geodesic(math.radians(inc), float(out["surf_alpha"][j]), float(out["surf_beta"][j]), flat=False)
                if gd is not None:
                    rr[j], mm[j], ok[j] = r, m, 1
                    kk[j] = [k[0], k[1], k[2], k[3]]
            out["surf%d_r" % ci] = rr; out["surf%d_m" % ci] = mm; out["surf%d_ok" % ci] = ok; out["surf%d_k" % ci] = kk
            # DiskRaytrace.image() of the thick disk (python/sim5diskraytrace.py:138-210; non-flat branch :176)
            ref.disk_nt_setup(10.0, a, 0.1, 0.1, 0)
            timg = rt.image(inc, rmax_s, Ns)
            for kq, v in timg.items():
                out["thk%d_%s" % (ci, kq)] = np.array(v, dtype=np.float64)
    finally:
        os.dup2(saved, 2)
    path = os.path.join(ROOT, "tests", "golden", "py_diskraytrace.npz")
    np.savez_compressed(path, **out)
    print(path, "%.1f KiB" % (os.path.getsize(path) / 1024.0))
    for ci in range(len(scases)):
        print("thick disk", scases[ci], "rays on the surface:", int(out["surf%d_ok" % ci].sum()), "of", Ns * Ns,
              "above the plane:", int((out["surf%d_m" % ci] > 1e-6).sum()))
    for ci in range(len(cases)):
        print(cases[ci], "pixels with flux:", int(np.isfinite(out["img%d_flux" % ci]).sum()),
              "geodesics:", int(np.isfinite(out["geo%d_r" % ci]).sum()))


if __name__ == "__main__":
    main()
