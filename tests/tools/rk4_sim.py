# lane-utilisation model of the torus march kernel from reference step patterns (CPU, test infra only)
import sys, math, ctypes as C, numpy as np
sys.path.insert(0, 'tests')
import oraclelib as ol
R = ol.Reference()
a, inc = 0.9, math.radians(70.0); r0 = 100.0
n = 1024; rmax = R.r_ms(a) + 8.0
rbh = R.r_bh(a)
def trace(ix, iy):
    alpha = ((ix + .5) / n - .5) * 2 * rmax; beta = ((iy + .5) / n - .5) * 2 * rmax
    g = ol.Geodesic(); err = C.c_int(0)
    if not R.geodesic_init_inf(inc, a, alpha, beta, C.byref(g), C.byref(err)): return np.zeros(0, np.int8)
    if not (r0 > g.rp): return np.zeros(0, np.int8)
    P = R.geodesic_P_int(C.byref(g), r0, 0)
    x = ol.D4(); k = ol.D4()
    x[0] = 0.0; x[1] = r0; x[2] = R.geodesic_position_pol(C.byref(g), P); x[3] = 0.0
    R.geodesic_momentum(C.byref(g), P, x[1], x[2], k)
    rtd = ol.RaytraceData()
    R.raytrace_prepare(a, x, k, 1.0, 0, C.byref(rtd))
    pat = []
    for i in range(20000):
        kt0 = rtd.kt
        dl = C.c_double(1e9)
        R.raytrace(x, k, C.byref(dl), C.byref(rtd))
        pat.append(1 if rtd.kt == kt0 else 0)
        if x[1] < 1.05 * rbh or x[1] > 1.01 * r0 or rtd.error > 1e-2: break
    return np.array(pat, np.int8)
waves = []
for (ix0, iy) in [(0, 100), (256, 300), (448, 480), (448, 512), (512, 520), (576, 600), (640, 512), (768, 700), (320, 512), (480, 400), (520, 560), (900, 900)]:
    waves.append([trace(ix0 + j, iy) for j in range(64)])
np.save('/tmp/rk4_waves.npy', np.array([[p for p in w] for w in waves], dtype=object), allow_pickle=True)
for Rc in (2.0, 3.0):
    tw = tt = 0.0
    for w in waves:
        L = max(len(p) for p in w)
        M = np.zeros((64, L), np.int8); A = np.zeros((64, L), bool)
        for j, p in enumerate(w): M[j, :len(p)] = p; A[j, :len(p)] = True
        work = A.sum() * 1.0 + M.sum() * Rc
        time = 64 * (A.any(0).sum() * 1.0 + (M.any(0)).sum() * Rc)
        tw += work; tt += time
        if Rc == 2.0: print("wave: steps min/max %d/%d  rk4 frac %.2f  sync util %.2f" % (min(len(p) for p in w), L, M.sum() / max(A.sum(), 1), work / time))
    print("R cost", Rc, "wave-synchronous utilisation %.3f" % (tw / tt))
