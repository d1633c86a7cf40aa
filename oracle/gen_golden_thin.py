#!/usr/bin/env python3
"""More golden vectors of the reference's PYTHON thin-disk image and per-ray geodesic() (python/sim5diskraytrace.py:138-252,
flat branch), in the layout of tests/golden/py_diskraytrace.npz, over what oracle/gen_golden_py.py holds fixed: spins 0 .. 0.998
(random ones too), inclinations 5 .. 86 degrees, image sizes 10 .. 22 (odd ones included: the central column / row), fields of view
r_ms + 8 / 20 / 50.  Captured like the other Python goldens (gen_golden_py.make_shim).
Output: tests/golden/py_thin_more.npz.

TEST INFRASTRUCTURE ONLY; needs /root/reference; nothing of the reference is copied.
"""
import logging
import math
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)
import oraclelib as ol  # noqa: E402
import gen_golden_py as G  # noqa: E402


def main():
    ref = ol.Reference()
    sys.modules["sim5lib"] = G.make_shim(ref)
    np.float = float                       # the reference predates numpy 1.24 (python/sim5diskraytrace.py:154)
    sys.path.insert(0, G.REFPY)
    logging.disable(logging.CRITICAL)
    import sim5diskmodel, sim5diskraytrace  # noqa: E402  (the reference's own modules)
    rng = np.random.default_rng(20260607)
    out = {}
    cases = []
    devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(2)
    t0 = time.time()
    os.dup2(devnull, 2)
    try:
        for ci in range(30):
            a = float([0.0, 0.3, 0.7, 0.9, 0.998][ci % 5]) if ci < 15 else float(rng.uniform(0.0, 0.998))
            inc = float(rng.uniform(5.0, 86.0))
            N = int(rng.integers(10, 23))
            disk = sim5diskmodel.DiskModel_ThinDisk(10.0, a, 0.1, 0.1)
            rt = sim5diskraytrace.DiskRaytrace(10.0, a, 10.0, disk, None)
            rmax = ref.r_ms(a) + float(rng.choice([8.0, 20.0, 50.0]))
            img = rt.image(inc, rmax, N)
            for k, v in img.items():
                out["img%d_%s" % (ci, k)] = np.array(v, dtype=np.float64)
            rr = np.full((N, N), np.nan); kk = np.full((N, N, 4), np.nan)
            for y in range(N):
                for x in range(N):
                    al = ((x + .5) / N - 0.5) * 2.0 * rmax; be = ((y + .5) / N - 0.5) * 2.0 * rmax
                    r, m, gd, k = rt.geodesic(math.radians(inc), al, be, flat=True)
                    if gd is not None:
                        rr[y, x] = r; kk[y, x] = [k[0], k[1], k[2], k[3]]
            out["geo%d_r" % ci] = rr; out["geo%d_k" % ci] = kk
            out["rmax%d" % ci] = np.array([rmax])
            cases.append((a, inc))
            os.write(saved, ("case %2d a=%.3f inc=%.1f N=%d rmax=%.1f: %d pixels lit, %d geodesics (%.0f s)\n" % (
                ci, a, inc, N, rmax, int(np.isfinite(out["img%d_flux" % ci]).sum()), int(np.isfinite(rr).sum()), time.time() - t0)).encode())
    finally:
        os.dup2(saved, 2)
    out["cases"] = np.array(cases)
    path = os.path.join(ROOT, "tests", "golden", "py_thin_more.npz")
    np.savez_compressed(path, **out)
    print(path, "%.1f KiB" % (os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
