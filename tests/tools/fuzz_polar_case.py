"""One case of tests/tools/fuzz_polar.py in detail (run on the GPU box): the pixel where the polarization angle of the fast and
the strict variant differ most, with the CPU checker's angle there and at alpha / beta one unit in the last place away.
usage: python tests/tools/fuzz_polar_case.py <seed> <case>"""
import sys, math, ctypes as C, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
from gpuutil import deg2rad
import oraclelib as ol
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(want + 1):
    a = float(rng.choice([0.0, 0.3, 0.9, 0.998, rng.uniform(0, 0.999)])); inc = float(rng.uniform(10.0, 85.0))
    nx, ny = int(rng.integers(17, 260)), int(rng.integers(2, 260))
print("case %d: a=%r inc=%r %dx%d" % (want, a, inc, nx, ny))
def pol(strict):
    d = capi.image_desc(nx, ny, a, deg2rad(inc), pol_degree=0.1, strict=strict)
    N = nx * ny
    st = capi.DeviceBuffer(3 * N * 8); chi = capi.DeviceBuffer(N * 8); r = capi.DeviceBuffer(N * 8)
    capi.disk_image_polarized_device(d, st.ptr, chi.ptr, aux={"r": r.ptr}); capi.synchronize()
    return st.to_numpy(np.float64, (3, ny, nx)), chi.to_numpy(np.float64, (ny, nx)), r.to_numpy(np.float64, (ny, nx))
S, CH, R = pol(False); Ss, CHs, Rs = pol(True)
both = np.isfinite(CH) & np.isfinite(CHs)
dc = np.where(both, np.abs(np.angle(np.exp(1j * (CH - CHs)))), 0.0)
iy, ix = np.unravel_index(int(np.argmax(dc)), dc.shape)
print("worst pixel (%d,%d): chi fast %.12f strict %.12f (diff %.2e)   r fast %.15g strict %.15g" % (iy, ix, CH[iy, ix], CHs[iy, ix], dc[iy, ix], R[iy, ix], Rs[iy, ix]))
drv = C.CDLL(ol.DRIVER_SO)
D, I, VP = C.c_double, C.c_int, C.c_void_p
drv.cpu_polarized_rays.argtypes = [C.c_char_p, C.c_char_p, D, D, D, I, VP, VP, VP, VP, VP, VP]
drv.cpu_polarized_rays.restype = I
rmax = ol.Oracle().r_ms(a) + 8.0
al0 = ((ix + .5) / nx - 0.5) * 2.0 * rmax; be0 = ((iy + .5) / ny - 0.5) * 2.0 * rmax * (ny / nx)
pts = [("the reference's (alpha, beta)", al0, be0)] + [("%s %s 1 ulp" % (nm, "+" if s > 0 else "-"), float(np.nextafter(al0, al0 + s)) if nm == "alpha" else al0,
        float(np.nextafter(be0, be0 + s)) if nm == "beta" else be0) for nm in ("alpha", "beta") for s in (1, -1)]
al = np.array([p[1] for p in pts]); be = np.array([p[2] for p in pts]); n = len(pts)
rchi = np.zeros(n); rr = np.zeros(n); rg = np.zeros(n); rwp = np.zeros((n, 2))
assert drv.cpu_polarized_rays(ol.ORACLE_SO.encode(), b"orc_", a, deg2rad(inc), -1.0, n, al.ctypes.data, be.ctypes.data,
                              rchi.ctypes.data, rr.ctypes.data, rg.ctypes.data, rwp.ctypes.data) == 0
for k, p in enumerate(pts):
    print("checker at %-28s chi = %.12f  r = %.15g  kappa = (%.6e, %.6e)   chi moved by %.2e" % (p[0], rchi[k], rr[k], rwp[k, 0], rwp[k, 1], abs(np.angle(np.exp(1j * (rchi[k] - rchi[0]))))))
