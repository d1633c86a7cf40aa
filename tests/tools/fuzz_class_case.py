"""One case of tests/tools/fuzz_images.py whose class maps differ between the variants (run on the GPU box): the pixels, both
variants' classes and records there, the CPU checker's class (reference build and restatement) for the pixel's ray, and the
margins of the decisions that set the class.   usage: python tests/tools/fuzz_class_case.py <seed> <case>"""
import sys, math, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
from gpuutil import deg2rad
import oraclelib as ol
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(want + 1):
    a = float(rng.choice([0.0, 1e-5, 0.3, 0.7, 0.9, 0.998, 0.9999, rng.uniform(0, 0.999)]))
    inc = float(rng.uniform(3.0, 87.0))
    nx, ny = int(rng.integers(17, 300)), int(rng.integers(2, 300))
    order = int(rng.choice([1, 2]))
    rmax = float(rng.choice([0.0, rng.uniform(3.0, 60.0)]))
print("case %d: a=%r inc=%r %dx%d order=%d rmax=%r" % (want, a, inc, nx, ny, order, rmax))
mk = lambda strict, **kw: capi.disk_image(capi.image_desc(nx, ny, a, deg2rad(inc), max_order=order, rmax=rmax, strict=strict, **kw), full=True)
f, s = mk(False), mk(True)
for kind in ("port", "reference"):
    c = ol.cpu_disk_image(kind, nx, ny, a, inc, nthreads=8, full=True) if (order == 2 and rmax == 0.0) else None
    if c is None:
        print(kind, ": the checker's image loop does not take this order / field of view"); continue
    print(kind, ": classes equal to strict:", bool(np.array_equal(c["cls"], s["cls"])), " to fast:", bool(np.array_equal(c["cls"], f["cls"])))
    for (iy, ix) in np.argwhere(f["cls"] != s["cls"]):
        print("   pixel (%d,%d): %s class %d r %.17g" % (iy, ix, kind, c["cls"][iy, ix], c["r"][iy, ix]))
for (iy, ix) in np.argwhere(f["cls"] != s["cls"]):
    print("pixel (%d,%d): fast class %d gtype %d r %.17g g %.17g P %s | strict class %d gtype %d r %.17g g %.17g" % (
        iy, ix, f["cls"][iy, ix], f["gtype"][iy, ix], f["r"][iy, ix], f["g"][iy, ix], f.get("P", np.full((ny, nx), np.nan))[iy, ix],
        s["cls"][iy, ix], s["gtype"][iy, ix], s["r"][iy, ix], s["g"][iy, ix]))
    z1 = 1 + (1 - a * a) ** (1 / 3) * ((1 + a) ** (1 / 3) + (1 - a) ** (1 / 3)); z2 = math.sqrt(3 * a * a + z1 * z1)
    rms = 3 + z2 - math.sqrt((3 - z1) * (3 + z1 + 2 * z2))
    print("   r_ms (double) %.17g ; neighbours' classes fast %s strict %s" % (rms,
          f["cls"][max(iy - 1, 0):iy + 2, max(ix - 1, 0):ix + 2].tolist(), s["cls"][max(iy - 1, 0):iy + 2, max(ix - 1, 0):ix + 2].tolist()))
