# ON THE GPU BOX: the step-wise job on a few different set-ups (spin, inclination, precision, size): kernel time by HIP events
import sys, math, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
import test_gpu_raytrace as T
out = []
for (n, a, inc, prec) in ((1024, 0.9, 70.0, 1.0), (1024, 0.5, 30.0, 1.0), (1024, 0.998, 80.0, 1.0), (724, 0.9, 70.0, 0.01), (1448, 0.0, 50.0, 1.0)):
    N = n * n
    dd = T.torus_desc(capi, n, a, inc, r0=100.0, precision=prec, max_steps=100000)
    sb = capi.DeviceBuffer(N * 40); steps = capi.DeviceBuffer(N * 4)
    capi.torus_image_device(dd, sb.ptr, aux={"steps": steps.ptr}); capi.synchronize()
    e0 = capi.Event(); e1 = capi.Event(); e0.record()
    for _ in range(3): capi.torus_image_device(dd, sb.ptr, aux={"steps": steps.ptr})
    e1.record(); ms = e0.elapsed_ms(e1) / 3
    s = steps.to_numpy(np.int32, (N,))
    out.append("%d^2 a=%g i=%g prec=%g: %.1f ms (%.2e steps/s, max %d)" % (n, a, inc, prec, ms, s.sum() / ms * 1e3, s.max()))
print(" | ".join(out))
