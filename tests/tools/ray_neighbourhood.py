"""ON THE GPU BOX: the thin-disk ray kernel (sim5gpu_disk_rays) on a square of rays around one (alpha, beta): where do the
fast and the strict variant part?  (How the outlier of a campaign -- one pixel 1000 times further from the strict value than
its neighbours -- is localised to a branch of the fast routine.)
usage: python tests/tools/ray_neighbourhood.py a inc_deg alpha beta [half_width] [n]"""
import sys, math, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
a, inc, al0, be0 = (float(x) for x in sys.argv[1:5])
hw = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-3
n = int(sys.argv[6]) if len(sys.argv) > 6 else 201
ax = np.linspace(-hw, hw, n)
AL, BE = np.meshgrid(al0 + ax, be0 + ax)
AL = AL.ravel().copy(); BE = BE.ravel().copy(); N = AL.size
res = {}
for strict in (False, True):
    desc = capi.image_desc(16, 16, a, math.radians(inc), strict=strict)
    b = {k: capi.DeviceBuffer(N * s) for k, s in (("al", 8), ("be", 8), ("f", 4), ("g4", 4), ("cls", 1), ("gtype", 1), ("r", 8), ("g", 8), ("flux", 8))}
    b["al"].from_numpy(AL); b["be"].from_numpy(BE)
    capi.disk_rays_device(desc, N, b["al"].ptr, b["be"].ptr, b["f"].ptr, b["g4"].ptr,
                          aux={k: b[k].ptr for k in ("cls", "gtype", "r", "g", "flux")})
    capi.synchronize()
    res[strict] = {"r": b["r"].to_numpy(np.float64, (n, n)), "cls": b["cls"].to_numpy(np.uint8, (n, n)), "g": b["g"].to_numpy(np.float64, (n, n))}
f, s = res[False], res[True]
print("classes equal:", bool(np.array_equal(f["cls"], s["cls"])), " hits:", int((s["cls"] == 2).sum()), "of", N)
ok = (f["cls"] == s["cls"]) & np.isfinite(s["r"])
er = np.where(ok, np.abs(f["r"] / s["r"] - 1), 0)
print("r error fast vs strict: median %.2e  99%% %.2e  max %.2e ; rays above 1e-8: %d" % (np.median(er[ok]), np.quantile(er[ok], 0.99), er.max(), int((er > 1e-8).sum())))
big = np.argwhere(er > 1e-8)
for (iy, ix) in big[:20]:
    print("  alpha %.17g beta %.17g  r fast %.15g strict %.15g  rel %.2e" % (al0 + ax[ix], be0 + ax[iy], f["r"][iy, ix], s["r"][iy, ix], er[iy, ix]))
if len(big):
    print("bounding box of those rays: alpha offsets %.3e .. %.3e, beta offsets %.3e .. %.3e" % (ax[big[:, 1].min()], ax[big[:, 1].max()], ax[big[:, 0].min()], ax[big[:, 0].max()]))
