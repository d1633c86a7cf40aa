#!/usr/bin/env python3
"""More golden vectors of the reference's PYTHON image of a disk of finite thickness (python/sim5diskraytrace.py:138-210 with the
non-flat branch :176, the surface search :228-335, the local frame / g-factor / emission angle :340-413), over what
oracle/gen_golden_py.py holds fixed: the surface TABLE (equal, logarithmic and growing steps; thin, steep, flaring, with a bump),
the radial-velocity profile (none, slow inflow, fast inflow), spins 0 .. 0.998, inclinations 8 .. 80 degrees, masses and accretion
rates, fields of view.  Captured like the other Python goldens (gen_golden_py.make_shim).
Output: tests/golden/py_thick_more.npz (inputs + the reference's image planes).

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
import gen_golden_surface as GS  # noqa: E402


def main():
    ref = ol.Reference()
    sys.modules["sim5lib"] = G.make_shim(ref)
    np.float = float                       # the reference predates numpy 1.24 (python/sim5diskraytrace.py:154)
    sys.path.insert(0, G.REFPY)
    logging.disable(logging.CRITICAL)
    import sim5diskmodel, sim5diskraytrace  # noqa: E402  (the reference's own modules)
    rng = np.random.default_rng(20260606)
    T = GS.tables()
    names = [n for n in sorted(T) if n not in ("two_nodes",)]
    out = {}
    cases = []
    devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(2)
    t0 = time.time()
    os.dup2(devnull, 2)
    try:
        for ci in range(30):
            name = names[ci % len(names)]
            tR, tH = T[name]
            a = float([0.0, 0.3, 0.7, 0.9, 0.998][ci % 5]) if ci < 15 else float(rng.uniform(0.0, 0.99))
            inc = float(rng.uniform(8.0, 80.0))
            mass = float(rng.choice([10.0, 1e6])); mdot = float(rng.choice([0.05, 0.3]))
            vkind = ci % 3
            tV = np.zeros_like(tR) if vkind == 0 else (-0.05 / np.sqrt(tR) if vkind == 1 else -0.3 / np.sqrt(tR))
            rmax = float(rng.choice([15.0, 30.0, 80.0]))
            Ns = 12

            class ThickDisk(sim5diskmodel.DiskModel):
                def _seg(self, R):
                    hi = int(np.searchsorted(tR, R, side="left")); return hi - 1, hi
                def h(self, R):
                    if not (R > tR[0]): return float(tH[0])
                    if R >= tR[-1]: return float(tH[-1] * (R / tR[-1]))
                    lo, hi = self._seg(R); w = (R - tR[lo]) / (tR[hi] - tR[lo])
                    return float(tH[lo] + w * (tH[hi] - tH[lo]))
                def dhdr(self, R):
                    if not (R > tR[0]): return 0.0
                    if R >= tR[-1]: return float(tH[-1] / tR[-1])
                    lo, hi = self._seg(R)
                    return float((tH[hi] - tH[lo]) / (tR[hi] - tR[lo]))
                def flux(self, R): return ref.disk_nt_flux(R)
                def l(self, R): return ref.disk_nt_ell(R)
                def vr(self, R):
                    if not (R > tR[0]): return float(tV[0])
                    if R >= tR[-1]: return float(tV[-1])
                    lo, hi = self._seg(R); w = (R - tR[lo]) / (tR[hi] - tR[lo])
                    return float(tV[lo] + w * (tV[hi] - tV[lo]))

            rt = sim5diskraytrace.DiskRaytrace(mass, a, 10.0, ThickDisk(), None)
            ref.disk_nt_setup(mass, a, mdot, 0.1, 0)
            img = rt.image(inc, rmax, Ns)
            for kq, v in img.items():
                out["c%d_%s" % (ci, kq)] = np.array(v, dtype=np.float64)
            out["c%d_tR" % ci] = tR; out["c%d_tH" % ci] = tH; out["c%d_tV" % ci] = tV
            cases.append((a, inc, rmax, Ns, mass, mdot, names.index(name), vkind))
            os.write(saved, ("case %2d %-20s a=%.3f inc=%.1f rmax=%g M=%g mdot=%g v%d: %d of %d pixels lit (%.0f s)\n" % (
                ci, name, a, inc, rmax, mass, mdot, vkind, int(np.isfinite(out["c%d_flux" % ci]).sum()), Ns * Ns, time.time() - t0)).encode())
    finally:
        os.dup2(saved, 2)
    out["cases"] = np.array(cases); out["table_names"] = np.array(names)
    path = os.path.join(ROOT, "tests", "golden", "py_thick_more.npz")
    np.savez_compressed(path, **out)
    print(path, "%.1f KiB" % (os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
