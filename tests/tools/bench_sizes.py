"""ON THE GPU BOX: the thin-disk image kernel at the sizes of C2 (1024^2), 2048^2 and the headline (4096^2), at the working
clock (~0.1 s of launches first), with the md5 of the image planes so that variants can be told apart (SIM5GPU_LIB selects
the library).  One line."""
import sys, math, hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sim5_amd.capi as capi
out = []
for n, reps in ((1024, 3000), (2048, 800), (4096, 250)):
    d = capi.image_desc(n, n, 0.998, math.radians(70.0))
    f = capi.DeviceBuffer(n * n * 4); g = capi.DeviceBuffer(n * n * 4)
    for _ in range(reps): capi.disk_image_device(d, f.ptr, g.ptr)
    capi.synchronize()
    e0 = capi.Event(); e1 = capi.Event(); e0.record()
    for _ in range(reps): capi.disk_image_device(d, f.ptr, g.ptr)
    e1.record(); ms = e0.elapsed_ms(e1) / reps
    F = f.to_numpy(np.float32, (n, n)); G = g.to_numpy(np.float32, (n, n))
    out.append("%d^2 %.4f ms frac %.3f hits %d md5 %s" % (n, ms, n * n * 1300 / (ms * 1e-3) / 78.6e12, int((G > 0).sum()),
                                                         hashlib.md5(F.tobytes() + G.tobytes()).hexdigest()[:8]))
print(" | ".join(out))
