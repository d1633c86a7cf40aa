# ON THE GPU BOX: how well do cheap geometric keys predict the rays of the C4 job that need many raytrace() calls?
# (steps from the kernel; constants of the geodesic of every 4th pixel from the CPU checker)
import sys, math, ctypes as C, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
import oraclelib as ol
import test_gpu_raytrace as T
n = 1024; N = n * n; a = 0.9; inc = math.radians(70.0)
dd = T.torus_desc(capi, n, a, 70.0, r0=100.0, precision=1.0, max_steps=100000)
sb = capi.DeviceBuffer(N * 40); steps = capi.DeviceBuffer(N * 4)
capi.torus_image_device(dd, sb.ptr, aux={"steps": steps.ptr}); capi.synchronize()
s = steps.to_numpy(np.int32, (n, n))
z1 = 1 + (1 - a * a) ** (1 / 3) * ((1 + a) ** (1 / 3) + (1 - a) ** (1 / 3)); z2 = math.sqrt(3 * a * a + z1 * z1)
rmax = 3 + z2 - math.sqrt((3 - z1) * (3 + z1 + 2 * z2)) + 8.0
o = ol.Oracle()
fn = o.lib.orc_geodesic_init_inf
fn.argtypes = [C.c_double] * 4 + [ol.PG, ol.PI]; fn.restype = C.c_int
rows = []
for iy in range(0, n, 4):
    for ix in range(0, n, 4):
        al = ((ix + .5) / n - 0.5) * 2 * rmax; be = ((iy + .5) / n - 0.5) * 2 * rmax
        gd = ol.Geodesic(); e = C.c_int(0)
        okk = fn(inc, a, al, be, C.byref(gd), C.byref(e))
        if not okk: continue
        if gd.nrr == 4: c = (gd.r1.re - gd.r2.re) / gd.r1.re; kind = 0
        elif gd.nrr == 2: c = abs(gd.r3.im) / max(abs(gd.r3.re), 1e-9); kind = 1
        else: c = 9.0; kind = 2
        rows.append((s[iy, ix], c, kind, abs(al) / rmax, math.hypot(al, be), 1.0 - gd.m2p))
R = np.array(rows)
st = R[:, 0]
print("sample %d rays; long (> 800 calls): %d" % (len(R), (st > 800).sum()))
for thr in (0.1, 0.2, 0.3, 0.4, 0.5):
    for pol in (0.002, 0.005, 0.01, 0.02):
        sel = ((R[:, 1] < thr) & (R[:, 2] < 2)) | (R[:, 5] < pol)
        print("crit < %.1f or 1 - m2p < %.3f: selects %.1f %% of the rays, holds %.1f %% of those > 800 calls, %.1f %% of those > 1000, longest left out %d" % (
            thr, pol, 100 * sel.mean(), 100 * (sel & (st > 800)).sum() / max((st > 800).sum(), 1), 100 * (sel & (st > 1000)).sum() / max((st > 1000).sum(), 1), st[~sel].max()))
