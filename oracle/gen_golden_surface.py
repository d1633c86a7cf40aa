#!/usr/bin/env python3
"""More golden vectors of the reference's PYTHON surface search (python/sim5diskraytrace.py:228-335, DiskRaytrace.geodesic with a
disk of finite thickness), over the things oracle/gen_golden_py.py holds fixed: the TABLE of the surface (equal, logarithmic and
growing steps; 2, 5, 64, 256 nodes; flat, thin, steep, with a bump), spins from 0 to 0.998, inclinations from 8 to 85 degrees, narrow
and wide fields of view (wide ones have rays that miss the disk and go through the reference's three retries).  Captured like the
other Python goldens: the reference's own class, imported in the build container, over a throw-away `sim5lib` made of ctypes calls
into the unmodified reference build (gen_golden_py.make_shim).  Output: tests/golden/py_surface_more.npz (inputs + outputs).

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


def tables():
    t = {}
    R = np.linspace(2.0, 60.0, 256); t["equal256_slope0.25"] = (R, 0.25 * (R - 2.0))
    R = np.linspace(1.5, 120.0, 64); t["equal64_slope0.5"] = (R, 0.5 * (R - 1.5))
    R = 10.0 ** np.linspace(0.2, 2.5, 200); t["log200_flaring"] = (R, 0.05 * R ** 1.2)
    R = 3.0 + 100.0 * np.linspace(0, 1, 120) ** 1.8; t["power120_bump"] = (R, 0.15 * (R - 3.0) + 2.0 * np.exp(-((R - 20.0) / 5.0) ** 2))
    t["two_nodes"] = (np.array([2.0, 80.0]), np.array([0.0, 20.0]))
    t["five_nodes"] = (np.array([2.0, 6.0, 15.0, 40.0, 100.0]), np.array([0.0, 0.5, 3.0, 6.0, 30.0]))
    R = np.linspace(4.0, 50.0, 33); t["thin33"] = (R, 0.02 * (R - 4.0))
    return t


def main():
    ref = ol.Reference()
    sys.modules["sim5lib"] = G.make_shim(ref)
    sys.path.insert(0, G.REFPY)
    logging.disable(logging.CRITICAL)
    import sim5diskmodel, sim5diskraytrace  # noqa: E402  (the reference's own modules)
    rng = np.random.default_rng(20260604)
    out = {}
    cases = []
    T = tables()
    names = sorted(T)
    devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(2)
    t0 = time.time()
    os.dup2(devnull, 2)
    try:
        for ci in range(60):
            name = names[ci % len(names)]
            tR, tH = T[name]
            a = float([0.0, 0.3, 0.7, 0.9, 0.998][ci % 5]) if ci < 15 else float(rng.uniform(0.05, 0.99))
            inc = float(rng.uniform(8.0, 85.0))
            rmax = float(rng.choice([12.0, 25.0, 60.0, 150.0]))
            Ns = 14
            c = ((np.arange(Ns) + .5) / Ns - 0.5) * 2.0 * rmax
            al, be = np.tile(c, Ns), np.repeat(c, Ns)

            class Disk(sim5diskmodel.DiskModel):
                def h(self, R, tR=tR, tH=tH):
                    if not (R > tR[0]): return float(tH[0])
                    if R >= tR[-1]: return float(tH[-1] * (R / tR[-1]))
                    hi = int(np.searchsorted(tR, R, side="left")); lo = hi - 1
                    w = (R - tR[lo]) / (tR[hi] - tR[lo])
                    return float(tH[lo] + w * (tH[hi] - tH[lo]))

            rt = sim5diskraytrace.DiskRaytrace(10.0, a, 10.0, Disk(), None)
            n = Ns * Ns
            rr = np.zeros(n); mm = np.zeros(n); ok = np.zeros(n, np.int32); kk = np.full((n, 4), np.nan)
            for j in range(n):
                r, m, gd, k = rt.geodesic(math.radians(inc), float(al[j]), float(be[j]), flat=False)
                if gd is not None:
                    rr[j], mm[j], ok[j] = r, m, 1
                    kk[j] = [k[0], k[1], k[2], k[3]]
            out["c%d_R" % ci] = tR; out["c%d_H" % ci] = tH; out["c%d_alpha" % ci] = al; out["c%d_beta" % ci] = be
            out["c%d_r" % ci] = rr; out["c%d_m" % ci] = mm; out["c%d_ok" % ci] = ok; out["c%d_k" % ci] = kk
            cases.append((a, inc, rmax, names.index(name)))
            os.write(saved, ("case %2d %-20s a=%.3f inc=%.1f rmax=%g: %d of %d rays on the surface (%.0f s)\n" % (
                ci, name, a, inc, rmax, int(ok.sum()), n, time.time() - t0)).encode())
    finally:
        os.dup2(saved, 2)
    out["cases"] = np.array(cases); out["table_names"] = np.array(names)
    path = os.path.join(ROOT, "tests", "golden", "py_surface_more.npz")
    np.savez_compressed(path, **out)
    print(path, "%.1f KiB" % (os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
