import sys, math, ctypes as C, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, 'scratch')
import os
devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 2)
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
    if k[0] != k[0]: return np.zeros(0, np.int8)
    rtd = ol.RaytraceData()
    R.raytrace_prepare(a, x, k, 1.0, 0, C.byref(rtd))
    pat = []
    for i in range(3000):
        kt0 = rtd.kt
        dl = C.c_double(1e9)
        R.raytrace(x, k, C.byref(dl), C.byref(rtd))
        pat.append(1 if rtd.kt == kt0 else 0)
        if x[1] < 1.05 * rbh or x[1] > 1.01 * r0 or rtd.error > 1e-2: break
    return np.array(pat, np.int8)
rng = np.random.default_rng(3)
def util(w, Rc):
    L = max(len(p) for p in w)
    if L == 0: return 0, 0
    M = np.zeros((len(w), L), np.int8); A = np.zeros((len(w), L), bool)
    for j, p in enumerate(w): M[j, :len(p)] = p; A[j, :len(p)] = True
    work = A.sum() * 1.0 + M.sum() * Rc
    time = 64 * (A.any(0).sum() * 1.0 + (M.any(0)).sum() * Rc)
    return work, time
res = {('row', 2.0): [0, 0], ('tile', 2.0): [0, 0]}
for t in range(24):
    ix0 = int(rng.integers(0, n - 64)); iy0 = int(rng.integers(0, n - 8))
    row = [trace(ix0 + j, iy0) for j in range(64)]
    tile = [trace(ix0 + j % 8, iy0 + j // 8) for j in range(64)]
    for name, w in (('row', row), ('tile', tile)):
        wk, tm = util(w, 2.0); res[(name, 2.0)][0] += wk; res[(name, 2.0)][1] += tm
    M = [p.mean() if len(p) else 0 for p in row]
    print(t, ix0, iy0, "row util %.2f tile util %.2f  rk4 frac %.2f  steps %d..%d" % (util(row, 2.0)[0] / max(util(row, 2.0)[1], 1), util(tile, 2.0)[0] / max(util(tile, 2.0)[1], 1), np.mean(M), min(len(p) for p in row), max(len(p) for p in row)), flush=True)
for k, v in res.items(): print(k, "overall %.3f" % (v[0] / v[1]))
