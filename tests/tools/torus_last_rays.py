# ON THE GPU BOX, with a -DS5_TORUS_DEBUG build (SIM5GPU_LIB): which rays of the C4 job end last, and what they are --
# call counts, impact parameter, place in the run (write_ray_end stores the wall clock in x_end[.,0] in such a build)
import sys, math, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
import test_gpu_raytrace as T
n = 1024; N = n * n
dd = T.torus_desc(capi, n, 0.9, 70.0, r0=100.0, precision=1.0, max_steps=100000)
sb = capi.DeviceBuffer(N * 40); steps = capi.DeviceBuffer(N * 4); xe = capi.DeviceBuffer(N * 32); dbg = capi.DeviceBuffer(N * 32)
for rep in range(2):
    xe.from_numpy(np.zeros(N * 4)); dbg.from_numpy(np.zeros(N * 4))
    capi.torus_image_device(dd, sb.ptr, aux={"steps": steps.ptr, "x_end": xe.ptr, "k_end": dbg.ptr}); capi.synchronize()
t = xe.to_numpy(np.float64, (N, 4))[:, 0]; s = steps.to_numpy(np.int32, (N,))
ok = t > 0
t0 = dbg.to_numpy(np.uint64, (N * 4,))[16:16 + 3 * 3072].reshape(-1, 3).astype(np.float64); t0 = t0[t0[:, 0] > 0][:, 0].min()
T1 = t[ok].max() - t0
rel = (t - t0) / T1
a = 0.9; z1 = 1 + (1 - a * a) ** (1 / 3) * ((1 + a) ** (1 / 3) + (1 - a) ** (1 / 3)); z2 = math.sqrt(3 * a * a + z1 * z1)
rmax = 3 + z2 - math.sqrt((3 - z1) * (3 + z1 + 2 * z2)) + 8.0
iy, ix = np.divmod(np.arange(N), n)
b = np.hypot(((ix + .5) / n - .5) * 2 * rmax, ((iy + .5) / n - .5) * 2 * rmax)
print("rays ended: %d of %d" % (ok.sum(), N))
for lo, hi in ((0.0, 0.5), (0.5, 0.8), (0.8, 0.9), (0.9, 0.95), (0.95, 0.98), (0.98, 0.99), (0.99, 1.001)):
    m = ok & (rel >= lo) & (rel < hi)
    if m.any():
        print("ended in [%.2f, %.2f) of the run: %7d rays | calls mean %4.0f p50 %4d p99 %4d max %4d | b mean %.1f, b > 8: %.0f %%, b < 4: %.0f %%" % (
            lo, hi, m.sum(), s[m].mean(), np.percentile(s[m], 50), np.percentile(s[m], 99), s[m].max(), b[m].mean(), 100 * (b[m] > 8).mean(), 100 * (b[m] < 4).mean()))
late = ok & (rel > 0.9) & (s > 800)
al = ((ix + .5) / n - .5) * 2 * rmax; be = ((iy + .5) / n - .5) * 2 * rmax
print("rays of more than 800 calls that ended in the last tenth: %d (of %d such rays in the job)" % (late.sum(), (ok & (s > 800)).sum()))
if late.any():
    print("  their b: min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f | |alpha| p50 %.2f | |beta| p50 %.2f" % (
        b[late].min(), *np.percentile(b[late], [10, 50, 90]), b[late].max(), np.median(np.abs(al[late])), np.median(np.abs(be[late]))))
    idx = np.flatnonzero(late)[:: max(1, late.sum() // 12)][:12]
    for i in idx: print("  ray %7d alpha %7.3f beta %7.3f calls %4d ended at %.3f" % (i, al[i], be[i], s[i], rel[i]))
# when were they handed out?  a ray's life is calls x the time per call; the start is the end minus that (unknown per ray) --
# instead: calls per unit of the run it lived, assuming it started at 0
