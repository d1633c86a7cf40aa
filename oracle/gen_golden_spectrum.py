#!/usr/bin/env python3
"""More golden vectors of the reference's PYTHON spectrum (python/sim5diskraytrace.py:122-123 over DiskRaytrace.image quantities,
python/sim5diskspectrum.py:54-88), over what oracle/gen_golden_py.py holds fixed: image sizes 12 .. 28 (odd ones included), spins
0 .. 0.998, inclinations 5 .. 86 degrees, black-hole masses 5 .. 1e8 and accretion rates 0.01 .. 1 (the temperature scale: spectra
that peak from the optical to hard X-rays), energy grids in logarithmic AND equal steps (the kernel's recurrence), every combination
of limb darkening and hardening.  Captured like the other Python goldens (gen_golden_py.make_shim).
Output: tests/golden/py_spectrum_more.npz (inputs + the reference's spectra).

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
    import sim5diskmodel, sim5diskraytrace, sim5diskspectrum  # noqa: E402  (the reference's own modules)
    rng = np.random.default_rng(20260605)
    out = {}
    cases = []
    devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(2)
    t0 = time.time()
    os.dup2(devnull, 2)
    try:
        for ci in range(36):
            a = float([0.0, 0.3, 0.7, 0.9, 0.998][ci % 5]) if ci < 20 else float(rng.uniform(0.0, 0.998))
            inc = float(rng.uniform(5.0, 86.0))
            N = int(rng.integers(12, 29))
            mass = float(rng.choice([5.0, 10.0, 30.0, 1e6, 1e8]))
            mdot = float(rng.choice([0.01, 0.1, 0.5, 1.0]))
            disk = sim5diskmodel.DiskModel_ThinDisk(mass, a, mdot, 0.1)
            rt = sim5diskraytrace.DiskRaytrace(mass, a, 10.0, disk, None)
            rmax = ref.r_ms(a) + float(rng.choice([8.0, 20.0, 60.0]))
            img = rt.image(inc, rmax, N)
            g_ = np.array(img["gfactor"], dtype=np.float64); T_ = np.array(img["T"], dtype=np.float64)
            mu_ = np.cos(np.radians(np.array(img["mue"], dtype=np.float64)))
            Tmax = float(np.nanmax(T_)) if np.isfinite(T_).any() else 1e7
            kT = 8.617333e-8 * Tmax                                   # keV: where this disk's spectrum peaks
            nE = int(rng.choice([24, 48, 64, 100]))
            if ci % 2:
                E = np.linspace(0.02 * kT, 30.0 * kT, nE)             # equal steps
            else:
                E = kT * 10.0 ** np.linspace(-2.0, 1.6, nE)           # logarithmic
            limb = int(rng.integers(0, 2)); hard = float(rng.choice([1.0, 1.5, 1.7, 2.4]))
            bb = sim5diskspectrum.DiskSpectrum_BlackBody()
            spec = np.zeros(len(E))
            for y in range(N):
                for x in range(N):
                    if not np.isfinite(g_[y, x]):
                        continue
                    e = mu_[y, x] if limb > 0 else -1.0
                    spec += bb.spectrum(T_[y, x], e, hard, E / g_[y, x]) * g_[y, x] ** 3
            out["c%d_E" % ci] = E; out["c%d_spec" % ci] = spec
            cases.append((a, inc, N, mass, mdot, rmax, limb, hard, ci % 2))
            os.write(saved, ("case %2d a=%.3f inc=%.1f N=%d M=%g mdot=%g rmax=%.1f limb=%d hard=%g %s grid of %d: lit pixels %d, peak %.3e (%.0f s)\n" % (
                ci, a, inc, N, mass, mdot, rmax, limb, hard, "equal" if ci % 2 else "log", nE, int(np.isfinite(g_).sum()), spec.max(), time.time() - t0)).encode())
    finally:
        os.dup2(saved, 2)
    out["cases"] = np.array(cases)
    path = os.path.join(ROOT, "tests", "golden", "py_spectrum_more.npz")
    np.savez_compressed(path, **out)
    print(path, "%.1f KiB" % (os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
