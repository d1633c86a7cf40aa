"""Randomised campaign for the polarized image (run on the GPU box; not part of the suite): the pairing kernel against the plain
one (bit for bit), and the fast variant against the strict one (Stokes I, Q, U within 1e-6 of the image's peak I, the angle on
the circle within 1e-6 where both have one).  The central column of an odd-width image (alpha = 0) is left out of the
fast/strict comparison.   usage: python tests/tools/fuzz_polar.py [n_cases] [seed]"""
import sys, math, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
from gpuutil import deg2rad

def pol(nx, ny, y0, y1, a, inc, strict):
    d = capi.image_desc(nx, ny, a, deg2rad(inc), y0=y0, y1=y1, pol_degree=0.1, strict=strict)
    rows = y1 - y0; N = rows * nx
    st = capi.DeviceBuffer(3 * N * 8); chi = capi.DeviceBuffer(N * 8); cls = capi.DeviceBuffer(N)
    capi.disk_image_polarized_device(d, st.ptr, chi.ptr, aux={"cls": cls.ptr}); capi.synchronize()
    return st.to_numpy(np.float64, (3, rows, nx)), chi.to_numpy(np.float64, (rows, nx)), cls.to_numpy(np.uint8, (rows, nx))

def probe_checker(a, inc, nx, ny, CH, CHs, both):
    import ctypes as C
    import oraclelib as ol
    dc = np.where(both, np.abs(np.angle(np.exp(1j * (CH - CHs)))), 0.0)
    iy, ix = np.unravel_index(int(np.argmax(dc)), dc.shape)
    drv = C.CDLL(ol.DRIVER_SO)
    D, I, VP = C.c_double, C.c_int, C.c_void_p
    drv.cpu_polarized_rays.argtypes = [C.c_char_p, C.c_char_p, D, D, D, I, VP, VP, VP, VP, VP, VP]
    drv.cpu_polarized_rays.restype = I
    rmax = ol.Oracle().r_ms(a) + 8.0
    al0 = ((ix + .5) / nx - 0.5) * 2.0 * rmax; be0 = ((iy + .5) / ny - 0.5) * 2.0 * rmax * (ny / nx)
    al = np.array([al0, np.nextafter(al0, al0 + 1), np.nextafter(al0, al0 - 1), al0, al0]); be = np.array([be0, be0, be0, np.nextafter(be0, be0 + 1), np.nextafter(be0, be0 - 1)])
    n = 5; rchi = np.zeros(n); rr = np.zeros(n); rg = np.zeros(n); rwp = np.zeros((n, 2))
    if drv.cpu_polarized_rays(ol.ORACLE_SO.encode(), b"orc_", a, deg2rad(inc), -1.0, n, al.ctypes.data, be.ctypes.data,
                              rchi.ctypes.data, rr.ctypes.data, rg.ctypes.data, rwp.ctypes.data) != 0: return 0.0
    d = np.abs(np.angle(np.exp(1j * (rchi[1:] - rchi[0]))))
    return float(np.nanmax(d)) if np.isfinite(d).any() else 0.0


ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
bad = 0; t0 = time.time()
for case in range(ncases):
    a = float(rng.choice([0.0, 0.3, 0.9, 0.998, rng.uniform(0, 0.999)])); inc = float(rng.uniform(10.0, 85.0))
    nx, ny = int(rng.integers(17, 260)), int(rng.integers(2, 260))
    S, CH, CL = pol(nx, ny, 0, ny, a, inc, False)
    cut = ny // 2 + 1 if ny > 2 else 1
    S1, CH1, CL1 = pol(nx, ny, 0, cut, a, inc, False); S2, CH2, CL2 = pol(nx, ny, cut, ny, a, inc, False)
    msg = []
    if not (np.array_equal(S, np.concatenate([S1, S2], axis=1), equal_nan=True) and np.array_equal(CH, np.concatenate([CH1, CH2]), equal_nan=True)
            and np.array_equal(CL, np.concatenate([CL1, CL2]))):
        msg.append("mirror != plain")
    Ss, CHs, CLs = pol(nx, ny, 0, ny, a, inc, True)
    col = np.ones((ny, nx), bool)
    if nx % 2 == 1: col[:, nx // 2] = False
    if not np.array_equal(CL[col], CLs[col]): msg.append("classes differ at %d px" % int((CL != CLs)[col].sum()))
    m = col & (CL == CLs)
    peak = max(float(Ss[0].max()), 1e-300)
    e = max(float(np.abs(S[k][m] - Ss[k][m]).max()) / peak for k in range(3)) if m.any() else 0.0
    both = m & np.isfinite(CH) & np.isfinite(CHs)
    ec = float(np.abs(np.angle(np.exp(1j * (CH[both] - CHs[both])))).max()) if both.any() else 0.0
    note = ""
    if e > 1e-6 or ec > 1e-6:
        # is it the INPUT?  (tests/tools/fuzz_images.py: the fast variant's pixel coordinates differ from the reference's expression
        # by an ulp) -- the CPU checker's own angle at the worst pixel for alpha / beta one unit in the last place away
        moved = probe_checker(a, inc, nx, ny, CH, CHs, both)
        if moved >= 0.5 * ec and e <= 1e-5: note = " [the CHECKER's angle moves by %.1e for one ulp of alpha / beta -- the difference is the input's]" % moved
        else: msg.append("Stokes %.1e of the peak, angle %.1e" % (e, ec))
    print("case %3d a=%.4g inc=%.1f %dx%d lit %d : %s" % (case, a, inc, nx, ny, int((S[0] > 0).sum()), ("ok" if not msg else "; ".join(msg)) + note), flush=True)
    bad += bool(msg)
print("%d cases, %d with findings, %.0f s" % (ncases, bad, time.time() - t0))
