# ON THE GPU BOX: where in the cursor's order are the long rays of the C4 job?  (raytrace() calls per ray from the kernel)
import sys, math, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
import test_gpu_raytrace as T
n = 1024; N = n * n
dd = T.torus_desc(capi, n, 0.9, 70.0, r0=100.0, precision=1.0, max_steps=100000)
sb = capi.DeviceBuffer(N * 40); steps = capi.DeviceBuffer(N * 4)
capi.torus_image_device(dd, sb.ptr, aux={"steps": steps.ptr}); capi.synchronize()
s = steps.to_numpy(np.int32, (N,)).astype(np.int64)
print("rays %d  total steps %.4e  mean %.1f  median %d  p99 %d  p99.9 %d  max %d" % (N, s.sum(), s.mean(), np.median(s), np.percentile(s, 99), np.percentile(s, 99.9), s.max()))
for lo in (600, 800, 1000, 1200, 1500, 1800):
    idx = np.nonzero(s > lo)[0]
    if len(idx) == 0: continue
    print("steps > %4d: %6d rays (%.3f %%), their share of all steps %.2f %%, positions in the order: first %.3f  median %.3f  last %.3f" % (
        lo, len(idx), 100.0 * len(idx) / N, 100.0 * s[idx].sum() / s.sum(), idx[0] / N, np.median(idx) / N, idx[-1] / N))
rows = s.reshape(n, n)
print("rows holding a ray > 1000 steps: %d .. %d of %d" % (np.nonzero((rows > 1000).any(axis=1))[0].min(), np.nonzero((rows > 1000).any(axis=1))[0].max(), n))
# a lower bound of the kernel time: every ray needs its steps one after the other
cum = np.cumsum(s) / s.sum()          # fraction of the work handed out before ray i, if work is consumed at a constant rate
for tau_us in (4.0, 6.0, 8.0):
    for T_ms in (20.0, 24.0, 29.0):
        end = cum * T_ms + s * tau_us * 1e-3
        print("  tau %.0f us/step, work spread over %.0f ms: last ray would end at %.1f ms (ray %.3f of the order, %d steps)" % (tau_us, T_ms, end.max(), np.argmax(end) / N, s[np.argmax(end)]))
print("max steps per 64 x 64 block (rows = image rows, top to bottom):")
blk = rows.reshape(16, 64, 16, 64).max(axis=(1, 3))
for r in blk: print(" ".join("%5d" % v for v in r))
print("mean steps per block:")
blk = rows.reshape(16, 64, 16, 64).mean(axis=(1, 3))
for r in blk: print(" ".join("%5d" % v for v in r))
