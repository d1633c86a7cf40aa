# RK4-fallback pattern of the reference's raytrace() along C4 rays (CPU, reference library): test infra only
import sys, math, ctypes as C, numpy as np
sys.path.insert(0, 'tests')
import oraclelib as ol
R = ol.Reference()
a, inc = 0.9, math.radians(70.0); r0 = 100.0
n = 1024; rmax = R.r_ms(a) + 8.0
def trace(ix, iy):
    alpha = ((ix + .5) / n - .5) * 2 * rmax; beta = ((iy + .5) / n - .5) * 2 * rmax
    g = ol.Geodesic(); err = C.c_int(0)
    if not R.geodesic_init_inf(inc, a, alpha, beta, C.byref(g), C.byref(err)): return None
    P = R.geodesic_P_int(C.byref(g), r0, 0)
    x = ol.D4(); k = ol.D4()
    x[0] = 0.0; x[1] = r0; x[2] = R.geodesic_position_pol(C.byref(g), P); x[3] = 0.0
    R.geodesic_momentum(C.byref(g), P, x[1], x[2], k)
    rtd = ol.RaytraceData()
    R.raytrace_prepare(a, x, k, 1.0, 0, C.byref(rtd))
    rbh = R.r_bh(a); pat = []; rs = []
    for i in range(5000):
        kt0 = rtd.kt
        dl = C.c_double(1e9)
        R.raytrace(x, k, C.byref(dl), C.byref(rtd))
        pat.append(1 if rtd.kt == kt0 else 0); rs.append(x[1])
        if x[1] < 1.05 * rbh or x[1] > 1.01 * r0 or rtd.error > 1e-2: break
    return np.array(pat), np.array(rs)
for (ix, iy) in [(100, 500), (400, 520), (512, 700), (600, 512), (900, 300)]:
    out = trace(ix, iy)
    if out is None: print(ix, iy, "init failed"); continue
    pat, rs = out
    runs = np.diff(np.flatnonzero(np.diff(np.concatenate([[-1], pat, [-1]])) != 0))
    print(ix, iy, "steps", len(pat), "rk4 frac %.3f" % pat.mean(), "P(rk4|prev rk4) %.3f" % (np.mean(pat[1:][pat[:-1] == 1]) if pat.sum() else 0),
          "P(rk4|prev verlet) %.3f" % np.mean(pat[1:][pat[:-1] == 0]))
    print("   pattern head:", "".join(map(str, pat[:120])))
    # rk4 fraction vs radius bins
    bins = [0, 3, 6, 10, 20, 40, 70, 101]
    idx = np.digitize(rs, bins)
    print("   rk4 frac by r-bin", [(bins[b - 1], round(float(pat[idx == b].mean()), 2), int((idx == b).sum())) for b in range(1, len(bins)) if (idx == b).any()])
