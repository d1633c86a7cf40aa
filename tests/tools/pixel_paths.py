"""ON THE GPU BOX: one pixel of a thin-disk image through the different launch forms of the fast variant (whole image in
mirrored pairs, a band of rows by the plain kernel, the direct routine, the ray alone and with its wave's 63 neighbours
through sim5gpu_disk_rays) against the strict variant -- which path produces an outlier?
usage: python tests/tools/pixel_paths.py a inc_deg nx ny ix iy"""
import sys, math, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
a, inc = float(sys.argv[1]), float(sys.argv[2])
nx, ny, ix, iy = (int(x) for x in sys.argv[3:7])
img = lambda **kw: capi.disk_image(capi.image_desc(nx, ny, a, math.radians(inc), max_order=1, **kw), full=True)
s = img(strict=True)
print("strict            r %.17g g %.17g" % (s["r"][iy, ix], s["g"][iy, ix]))
f = img()
print("fast, whole image r %.17g g %.17g  rel %.2e" % (f["r"][iy, ix], f["g"][iy, ix], abs(f["r"][iy, ix] / s["r"][iy, ix] - 1)))
d = img(direct=True)
print("fast, direct      r %.17g g %.17g  rel %.2e" % (d["r"][iy, ix], d["g"][iy, ix], abs(d["r"][iy, ix] / s["r"][iy, ix] - 1)))
for (y0, y1) in ((iy - iy % 16, iy - iy % 16 + 16), (iy, iy + 1), (iy - 1, iy + 3), (max(iy - 40, 0), min(iy + 40, ny))):
    b = img(y0=y0, y1=y1)
    print("fast, rows %3d..%3d r %.17g  rel %.2e" % (y0, y1, b["r"][iy - y0, ix], abs(b["r"][iy - y0, ix] / s["r"][iy, ix] - 1)))
# the rays of the pixel's 16 x 4 patch through the ray kernel, in lane order; then the pixel's ray alone (64 copies)
z1 = 1 + (1 - a * a) ** (1 / 3) * ((1 + a) ** (1 / 3) + (1 - a) ** (1 / 3)); z2 = math.sqrt(3 * a * a + z1 * z1)
rms = 3 + z2 - math.sqrt((3 - z1) * (3 + z1 + 2 * z2)); rm = rms + 8.0
x0, y0 = ix - ix % 16, iy - iy % 4
X, Y = np.meshgrid(np.arange(x0, x0 + 16), np.arange(y0, y0 + 4))
AL = (((X + .5) / nx - 0.5) * 2.0 * rm).ravel().copy(); BE = (((Y + .5) / ny - 0.5) * 2.0 * rm * (ny / nx)).ravel().copy()
def rays(AL, BE, strict=False):
    N = AL.size
    desc = capi.image_desc(16, 16, a, math.radians(inc), strict=strict, max_order=1)
    b = {k: capi.DeviceBuffer(N * sz) for k, sz in (("al", 8), ("be", 8), ("f", 4), ("g4", 4), ("cls", 1), ("gtype", 1), ("r", 8), ("g", 8), ("flux", 8))}
    b["al"].from_numpy(AL); b["be"].from_numpy(BE)
    capi.disk_rays_device(desc, N, b["al"].ptr, b["be"].ptr, b["f"].ptr, b["g4"].ptr, aux={k: b[k].ptr for k in ("cls", "gtype", "r", "g", "flux")})
    capi.synchronize()
    return b["r"].to_numpy(np.float64, (N,)), b["gtype"].to_numpy(np.int8, (N,)), b["cls"].to_numpy(np.uint8, (N,))
lane = (iy - y0) * 16 + (ix - x0)
r, gt, cl = rays(AL, BE)
rs, _, _ = rays(AL, BE, strict=True)
print("ray kernel, the patch's 64 rays: pixel r %.17g rel %.2e ; worst of the patch %.2e ; classes %s gtypes %s" % (
    r[lane], abs(r[lane] / rs[lane] - 1), np.nanmax(np.abs(r / rs - 1)), sorted(set(cl.tolist())), sorted(set(gt.tolist()))))
r1, _, _ = rays(np.full(64, AL[lane]), np.full(64, BE[lane]))
print("ray kernel, the ray 64 times:    pixel r %.17g rel %.2e" % (r1[0], abs(r1[0] / rs[lane] - 1)))
# which neighbour matters: the pixel's ray everywhere except one lane
for other in range(64):
    if other == lane: continue
    A2 = np.full(64, AL[lane]); B2 = np.full(64, BE[lane]); A2[other] = AL[other]; B2[other] = BE[other]
    r2, gt2, cl2 = rays(A2, B2)
    e = abs(r2[lane] / rs[lane] - 1)
    if e > 1e-9:
        print("   with lane %2d (pixel %d,%d: class %d gtype %d r %.6g) beside it: rel %.2e" % (other, y0 + other // 16, x0 + other % 16, cl2[other], gt2[other], r2[other], e))
