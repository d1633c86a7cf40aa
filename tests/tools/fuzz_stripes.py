"""Randomised campaign for the multi-GPU row dealing (run on the GPU box): random image sizes, stripe heights, world sizes and
dealt fractions; every rank's launch (mirrored stripes, SIM5GPU_IMG_MIRROR) and rank 0's band must reproduce the rows of the
whole image bit for bit, and the shares must tile the image exactly once.   usage: python tests/tools/fuzz_stripes.py [n] [seed]"""
import sys, math, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
from gpuutil import deg2rad
from sim5_amd import sharding as sh

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2)
bad = 0
for case in range(ncases):
    nx, ny = int(rng.integers(16, 400)), int(rng.integers(2, 1200))
    world = int(rng.integers(1, 10)); stripe = int(rng.choice([1, 3, 16, 64, 100]))
    half = sh.upper_half(ny); unit = stripe * world
    dealt = None
    if world > 1 and rng.random() < 0.6 and half >= unit:     # (a single GPU keeps everything: no band)
        dealt = int(rng.integers(1, half // unit + 1)) * unit
        if dealt >= half: dealt = None
    a, inc = float(rng.choice([0.3, 0.9, 0.998])), float(rng.uniform(10, 80))
    strict = bool(rng.random() < 0.3)
    whole = capi.disk_image(capi.image_desc(nx, ny, a, deg2rad(inc), strict=strict), full=True)
    seen = np.zeros(ny, int); msg = []
    for r in range(world):
        kw = sh.job_rows(ny, r, world, stripe=stripe, dealt=dealt)
        rows = sh.stripes_for_rank(ny, r, world, stripe=stripe, dealt=dealt)
        for (y0, y1) in rows: seen[y0:y1] += 1
        if kw["y0"] >= kw["y1"]:
            if rows: msg.append("rank %d: rows without a job" % r)
            continue
        d = capi.image_desc(nx, ny, a, deg2rad(inc), strict=strict, **kw)
        if capi.image_rows(d) != sum(y1 - y0 for y0, y1 in rows):
            msg.append("rank %d: library counts %d rows, dealing %d" % (r, capi.image_rows(d), sum(y1 - y0 for y0, y1 in rows))); continue
        t = capi.disk_image(d, full=True)
        for k in ("image_f", "image_g", "cls", "r"):
            if not np.array_equal(t[k], np.concatenate([whole[k][y0:y1] for (y0, y1) in rows]), equal_nan=True):
                msg.append("rank %d: %s differs" % (r, k)); break
    band = sh.root_band(ny, dealt)
    if band:
        seen[band[0]:band[1]] += 1
        t = capi.disk_image(capi.image_desc(nx, ny, a, deg2rad(inc), strict=strict, y0=band[0], y1=band[1]), full=True)
        if not np.array_equal(t["image_g"], whole["image_g"][band[0]:band[1]]): msg.append("band differs")
    if not (seen == 1).all(): msg.append("rows covered %s times" % sorted(set(seen.tolist())))
    print("case %3d %dx%d world %d stripe %d dealt %s %s : %s" % (case, nx, ny, world, stripe, dealt, "strict" if strict else "fast", "ok" if not msg else "; ".join(msg[:4])), flush=True)
    bad += bool(msg)
print("%d cases, %d with findings" % (ncases, bad))
