import sys, math, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import sim5_amd.capi as capi
g = np.load('tests/golden/kat_geodesic.npz')
inp = g["inp"]
rec, err, ok = capi.geodesic_init_inf(inp[:, 0], inp[:, 1], inp[:, 2], inp[:, 3])
ref = np.frombuffer(g["dump"].tobytes(), dtype=capi.GEODESIC_DTYPE)
good = ok == 1
for f in ("l","q","m2p","m2m","mm","mK","rp","Rpc","Tpp","Tip","r1","r2","r3","r4"):
    a = rec[f][good]; b = ref[f][good]
    if a.ndim == 2:
        d = np.abs(a-b).max(axis=1); den = np.maximum(np.abs(b).max(axis=1), 1e-300)
    else:
        d = np.abs(a-b); den = np.maximum(np.abs(b), 1e-300)
    e = d/den
    e[np.isnan(e)] = 0
    idx = np.argsort(-e)[:4]
    print(f, "worst:", [(float("%.2e"%e[i]), int(ref["type"][good][i]), float(inp[good][i,1]), float("%.4g"%np.ravel(b[i])[0])) for i in idx])
# type-wise max error for rp, Rpc
for t in (40,2,0):
    m = good & (ref["type"]==t)
    for f in ("rp","Rpc"):
        a = rec[f][m]; b = ref[f][m]
        e = np.abs(a-b)/np.maximum(np.abs(b),1e-300)
        print("type",t,f,"max rel", np.nanmax(e) if e.size else None)
# flat sphere
sys.path.insert(0,'tests')
import test_gpu_raytrace as T
n, R = 48, 6.0
d = T.torus_desc(capi, n, 0.5, 60.0, options=1, shape=1, torus_w=R, torus_l=0.0, r0=40.0, dl_max=0.02, max_steps=200000, max_error=1.0, r_stop_in=1e-3)
d.img.rmax = 8.0
S, steps, xe, ce, me = T.run_torus(capi, d)
c = ((np.arange(n) + .5) / n - 0.5) * 2.0 * 8.0
b = np.hypot(c[None, :], c[:, None]).ravel()
chord = np.where(b < R, 2.0 * np.sqrt(np.maximum(R * R - b * b, 0.0)), 0.0)
o = np.argsort(b)
for i in o[::60]:
    print("b=%.3f chord=%.4f I=%.4f steps=%d r_end=%.3f maxerr=%.2e" % (b[i], chord[i], S[i,0], steps[i], xe[i,1], me[i]))
