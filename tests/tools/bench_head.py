"""ON THE GPU BOX: the headline image kernel only, at the working clock; one line (SIM5GPU_LIB selects the library)."""
import sys, math, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import sim5_amd.capi as capi
n = 4096
d = capi.image_desc(n, n, 0.998, math.radians(70.0))
f = capi.DeviceBuffer(n * n * 4); g = capi.DeviceBuffer(n * n * 4)
for _ in range(250): capi.disk_image_device(d, f.ptr, g.ptr)
capi.synchronize()
e0 = capi.Event(); e1 = capi.Event(); e0.record()
for _ in range(250): capi.disk_image_device(d, f.ptr, g.ptr)
e1.record(); ms = e0.elapsed_ms(e1) / 250
print("4096^2 %.4f ms  hits %d" % (ms, int((g.to_numpy(np.float32, (n, n)) > 0).sum())))
