# ON THE GPU BOX: march-kernel time against the number of rays (same job, n x n grids): T(N) = a + b N, the intercept a is what
# ramp and drain of the persistent grid cost
import sys, math, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
import test_gpu_raytrace as T
rows = []
for n in (512, 724, 1024, 1448, 2048):
    N = n * n
    dd = T.torus_desc(capi, n, 0.9, 70.0, r0=100.0, precision=1.0, max_steps=100000)
    sb = capi.DeviceBuffer(N * 40); steps = capi.DeviceBuffer(N * 4)
    capi.torus_image_device(dd, sb.ptr, aux={"steps": steps.ptr}); capi.synchronize()
    e0 = capi.Event(); e1 = capi.Event(); e0.record()
    for _ in range(3): capi.torus_image_device(dd, sb.ptr, aux={"steps": steps.ptr})
    e1.record(); ms = e0.elapsed_ms(e1) / 3
    s = steps.to_numpy(np.int32, (N,))
    rows.append((N, ms, float(s.sum())))
    print("%4d^2  %8.2f ms  %.4e steps  %.3e steps/s  max steps of a ray %d" % (n, ms, s.sum(), s.sum() / ms * 1e3, s.max()), flush=True)
A = np.array([[1.0, r[2]] for r in rows]); y = np.array([r[1] for r in rows])
(a, b), *_ = np.linalg.lstsq(A, y, rcond=None)
print("fit T = a + b * steps: a = %.2f ms, 1/b = %.3e steps/s (asymptotic rate)" % (a, 1e3 / b))
