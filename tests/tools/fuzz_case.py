"""One case of tests/tools/fuzz_images.py in detail (run on the GPU box): the pixel where fast and strict differ most in r,
with the CPU oracle's value beside them.   usage: python tests/tools/fuzz_case.py <seed> <case>"""
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
mk = lambda strict: capi.disk_image(capi.image_desc(nx, ny, a, deg2rad(inc), max_order=order, rmax=rmax, strict=strict), full=True)
f, s = mk(False), mk(True)
c = ol.cpu_disk_image("port", nx, ny, a, inc, nthreads=8, full=True) if (order == 2 and rmax == 0.0) else None
print("classes of fast and strict equal:", bool(np.array_equal(f["cls"], s["cls"])))
same = (f["cls"] == s["cls"]) & np.isfinite(s["r"])
er = np.where(same, np.abs(f["r"] / s["r"] - 1), 0)
worst_first = sorted(map(tuple, np.argwhere(er > 1e-10)), key=lambda t: -er[t])
for (iy, ix) in worst_first[:12]:
    print("pixel (%d,%d) cls %d gtype %d  r fast %.15g strict %.15g %s  rel %.2e | g fast %.12g strict %.12g" % (
        iy, ix, f["cls"][iy, ix], f["gtype"][iy, ix], f["r"][iy, ix], s["r"][iy, ix],
        ("oracle %.15g (strict-oracle %.1e, fast-oracle %.1e)" % (c["r"][iy, ix], abs(s["r"][iy, ix] / c["r"][iy, ix] - 1), abs(f["r"][iy, ix] / c["r"][iy, ix] - 1))) if c is not None else "",
        er[iy, ix], f["g"][iy, ix], s["g"][iy, ix]))
print("pixels with r rel > 1e-10:", int((er > 1e-10).sum()), "of", int(same.sum()))
# conditioning of the worst pixel in the CHECKER itself: its (alpha, beta) by the reference's expression, then the same ray with
# beta and alpha moved by one unit in the last place -- how far does the checker's own r move?
bad = np.argwhere(er > 1e-10)
if len(bad):
    iy, ix = bad[np.argmax(er[tuple(bad.T)])]
    o = ol.Oracle()
    rms = ol.cpu_r_ms(a) if hasattr(ol, "cpu_r_ms") else None
    z1 = 1 + (1 - a * a) ** (1 / 3) * ((1 + a) ** (1 / 3) + (1 - a) ** (1 / 3)); z2 = math.sqrt(3 * a * a + z1 * z1)
    rms = 3 + z2 - math.sqrt((3 - z1) * (3 + z1 + 2 * z2))
    o.disk_nt_setup(10.0, a, 0.1, 0.1, 0)
    rm = rmax if rmax > 0.0 else rms + 8.0
    al = ((ix + .5) / nx - 0.5) * 2.0 * rm
    be = ((iy + .5) / ny - 0.5) * 2.0 * rm * (ny / nx)
    r0 = o.disk_pixel(deg2rad(inc), a, rms, al, be).r
    print("checker at the reference's (alpha, beta): r = %.15g" % r0)
    def ulps(x, k):
        for _ in range(abs(k)):
            x = float(np.nextafter(x, math.inf if k > 0 else -math.inf))
        return x
    for name, da, db in (("beta + 1 ulp", 0, 1), ("beta - 1 ulp", 0, -1), ("beta + 2 ulp", 0, 2), ("beta - 2 ulp", 0, -2), ("alpha + 1 ulp", 1, 0), ("alpha - 1 ulp", -1, 0)):
        a2 = ulps(al, da)
        b2 = ulps(be, db)
        r1 = o.disk_pixel(deg2rad(inc), a, rms, float(a2), float(b2)).r
        print("  %-14s r = %.15g   moved by %.2e (relative)" % (name, r1, abs(r1 / r0 - 1)))
