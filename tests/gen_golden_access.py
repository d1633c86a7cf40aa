"""Thin access to the recipe drivers of oracle/cpu_driver.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import math

import numpy as np

import oraclelib as ol


def _driver():
    if not __import__("os").path.exists(ol.DRIVER_SO):
        ol.build_oracle()
    drv = C.CDLL(ol.DRIVER_SO)
    VP, D, I = C.c_void_p, C.c_double, C.c_int
    drv.cpu_polarized_rays.argtypes = [C.c_char_p, C.c_char_p, D, D, D, I, VP, VP, VP, VP, VP, VP]
    drv.cpu_polarized_rays.restype = I
    drv.cpu_verlet_trace.argtypes = [C.c_char_p, C.c_char_p, D, D, D, D, D, D, I, D, D, D, D, I, VP, VP, VP, VP]
    drv.cpu_verlet_trace.restype = I
    return drv


def verlet_traces(lib, prefix, cases, nmax, dl_max=1e9, max_error=1e-2, r_in_fac=1.05, r_out_fac=1.01):
    """cases: (a, inc_rad, alpha, beta, r0, precision, options).  Returns [(n, trace, x0, k0, carter)]."""
    drv = _driver()
    out = []
    for (a, inc, al, be, r0, prec, opt) in cases:
        tr = np.zeros((nmax, 11)); xs = np.zeros(4); ks = np.zeros(4); car = C.c_double(float("nan"))
        rbh = 1.0 + math.sqrt(1.0 - a * a)
        n = drv.cpu_verlet_trace(lib.encode(), prefix.encode(), a, inc, al, be, r0, prec, int(opt), dl_max,
                                 r_in_fac * rbh, r_out_fac * r0, max_error, nmax, tr.ctypes.data,
                                 xs.ctypes.data, ks.ctypes.data, C.byref(car))
        out.append((n, tr, xs, ks, car.value))
    return out


def polarized_rays(lib, prefix, a, inc_deg, alpha, beta, rms=-1.0):
    drv = _driver()
    al = np.ascontiguousarray(alpha, dtype=np.float64); be = np.ascontiguousarray(beta, dtype=np.float64)
    m = al.size
    chi = np.zeros(m); r = np.zeros(m); g = np.zeros(m); wp = np.zeros((m, 2))
    rc = drv.cpu_polarized_rays(lib.encode(), prefix.encode(), a, inc_deg / 180.0 * math.pi, rms, m,
                                al.ctypes.data, be.ctypes.data, chi.ctypes.data, r.ctypes.data,
                                g.ctypes.data, wp.ctypes.data)
    assert rc == 0
    return chi, r, g, wp


def torus_rays(lib, prefix, a, inc_rad, alpha, beta, r0=100.0, precision=1.0, options=0, dl_max=1e9,
               r_in_fac=1.05, r_out_fac=1.01, max_error=1e-2, max_steps=20000, shape=0, torus_r=8.0, torus_w=2.0,
               torus_l=3.5, emis0=1.0, absorb0=0.0, ulps=None):
    """The C4 job on the host (oracle/cpu_driver.c:cpu_torus_rays): the checker library's raytrace() loop with the
    build-defined transfer accumulated per step.  Returns a dict of per-ray arrays.  `ulps` (8 ints): start state
    (x, k) moved by that many units in the last place before raytrace_prepare() (conditioning probe)."""
    drv = _driver()
    VP, D, I = C.c_void_p, C.c_double, C.c_int
    fn = drv.cpu_torus_rays_perturbed
    fn.argtypes = [C.c_char_p, C.c_char_p, D, D, I, VP, VP, D, D, I, D, D, D, D, I, I, D, D, D, D, D,
                   VP, VP, VP, VP, VP, VP, VP, VP]
    fn.restype = I
    shift = None if ulps is None else np.ascontiguousarray(ulps, dtype=np.int32)
    assert shift is None or shift.size == 8
    al = np.ascontiguousarray(alpha, dtype=np.float64).ravel(); be = np.ascontiguousarray(beta, dtype=np.float64).ravel()
    n = al.size
    out = {"steps": np.zeros(n, np.int32), "x_end": np.zeros((n, 4)), "k_end": np.zeros((n, 4)), "I": np.zeros(n),
           "tau": np.zeros(n), "carter": np.zeros(n), "max_step_error": np.zeros(n, np.float32)}
    rbh = 1.0 + math.sqrt(1.0 - a * a)
    rc = fn(lib.encode(), prefix.encode(), a, inc_rad, n, al.ctypes.data, be.ctypes.data, r0, precision, int(options),
            dl_max, r_in_fac * rbh, r_out_fac * r0, max_error, int(max_steps), int(shape), torus_r, torus_w, torus_l,
            emis0, absorb0, *[out[k].ctypes.data for k in ("steps", "x_end", "k_end", "I", "tau", "carter", "max_step_error")],
            None if shift is None else shift.ctypes.data)
    assert rc == 0, rc
    return out
