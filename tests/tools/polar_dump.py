# ON THE GPU BOX: Stokes planes of the C3 job from the library $SIM5GPU_LIB names, saved for a pixel-by-pixel comparison
import sys, math, numpy as np
sys.path.insert(0, '.')
import sim5_amd.capi as capi
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
d = capi.image_desc(n, n, 0.9, 70 / 180 * math.pi, pol_degree=0.1)
st = capi.DeviceBuffer(3 * n * n * 8); ch = capi.DeviceBuffer(n * n * 8)
capi.disk_image_polarized_device(d, st.ptr, ch.ptr)
capi.synchronize()
np.save(sys.argv[1], np.concatenate([st.to_numpy(np.float64, (3, n, n)), ch.to_numpy(np.float64, (1, n, n))]))
