# ON THE GPU BOX: the image job examples/disk_image_sharded.c gives rank 0 of a world of one (mirrored 64-row stripes of the
# whole image, traced in place), through the Python binding; stage by stage
import sys, math, numpy as np
sys.path.insert(0, '.')
import sim5_amd.capi as capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
a, inc = 0.998, math.radians(70.0)
f = capi.DeviceBuffer(n * n * 4); g = capi.DeviceBuffer(n * n * 4)
d0 = capi.image_desc(n, n, a, inc)
capi.disk_image_device(d0, f.ptr, g.ptr); capi.synchronize()
print("whole image: hits", int((g.to_numpy(np.float32, (n, n)) > 0).sum()), flush=True)
d1 = capi.image_desc(n, n, a, inc, y0=0, y1=n // 2, stripe_rows=64, stripe_step=64, mirror=True, inplace=True)
print("rows of the share:", capi.image_rows(d1), flush=True)
capi.disk_image_device(d1, f.ptr, g.ptr); capi.synchronize()
print("in place, mirrored stripes: hits", int((g.to_numpy(np.float32, (n, n)) > 0).sum()), flush=True)
g1 = g.to_numpy(np.float32, (n, n)).copy(); f1 = f.to_numpy(np.float32, (n, n)).copy()
capi.disk_image_device(d0, f.ptr, g.ptr); capi.synchronize()
g0 = g.to_numpy(np.float32, (n, n)).copy(); f0 = f.to_numpy(np.float32, (n, n)).copy()
bad = np.argwhere((g0 != g1) | (f0 != f1))
print("pixels that differ:", len(bad))
full = capi.disk_image(d0, full=True)
for (iy, ix) in bad[:16]:
    print(iy, ix, "plain g %.9g f %.6g | in place g %.9g f %.6g | aux kernel: cls %d gtype %d r %.17g g %.9g" %
          (g0[iy, ix], f0[iy, ix], g1[iy, ix], f1[iy, ix], full["cls"][iy, ix], full["gtype"][iy, ix], full["r"][iy, ix], full["g"][iy, ix]))
# twice more in place: the same pixels every time?
for rep in range(2):
    capi.disk_image_device(d1, f.ptr, g.ptr); capi.synchronize()
    g2 = g.to_numpy(np.float32, (n, n))
    print("in place again: hits", int((g2 > 0).sum()), "differs from the first in-place image at", int((g2 != g1).sum()), "pixels")
