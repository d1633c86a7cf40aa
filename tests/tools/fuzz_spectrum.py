"""Randomised parity campaign for the spectrum job (run on the GPU box; not part of the suite): the fast kernel (mirrored pairs,
closed-form frame, 16-slot Planck factor: k_spectrum.hip) against the strict kernel (the reference's tetrad chain and a
full-precision Planck function, ref python/sim5diskraytrace.py:340-390, python/sim5diskspectrum.py:54-88) on random jobs: spin,
inclination, image size (odd and even), field of view, hardening factor, limb darkening on/off, disk spin equal or not, 1..400
energies on log grids of random span (up to 1e13 keV: the bounded form of the loop), and random row sub-sets (the unpaired kernel).
Every bin above 1e-250 of the peak within 1e-6; bins the strict kernel has at zero at zero; two runs identical.
usage: python tests/tools/fuzz_spectrum.py [n_cases] [seed] [uniform]   (uniform: grids in equal steps, 64 .. 400 energies)"""
import sys, math, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
from gpuutil import deg2rad

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
worst = 0.0
t0 = time.time()
for case in range(ncases):
    a = float(rng.choice([1e-4, 0.3, 0.7, 0.9, 0.998, 0.9999, rng.uniform(1e-4, 0.999)]))
    inc = float(rng.uniform(3.0, 87.0))
    nx, ny = int(rng.integers(17, 260)), int(rng.integers(2, 260))
    rmax = float(rng.choice([0.0, rng.uniform(3.0, 200.0)]))
    hard = float(rng.choice([1.0, 1.7, rng.uniform(1.0, 3.0)]))
    limb = int(rng.integers(0, 2))
    dspin = float(rng.choice([-1.0, -1.0, 0.0, rng.uniform(0, 0.99)]))
    ne = int(rng.choice([1, 7, 64, 128, 200, 256, 400, rng.integers(1, 400)]))
    lo = float(rng.uniform(-4, 0)); hi = lo + float(rng.choice([rng.uniform(0.5, 4), rng.uniform(4, 16)]))
    E = 10.0 ** np.linspace(lo, hi, ne)
    if len(sys.argv) > 3 and sys.argv[3] == "uniform":
        # round 6: EQUAL steps (>= 64 energies take the recurrence along the energies, k_spectrum.hip planck_runs_uniform) -- from
        # 10^lo up to a random multiple of it, so that runs start in the Rayleigh-Jeans range and end in the Wien tail
        ne = int(rng.choice([64, 71, 128, 200, 256, 400, rng.integers(64, 400)]))
        E = np.linspace(10.0 ** lo, 10.0 ** lo * float(rng.choice([3.0, 30.0, 1e3, 1e5])), ne)
    if rng.random() < 0.3:
        y0 = int(rng.integers(0, ny)); y1 = int(rng.integers(y0 + 1, ny + 1))
    else:
        y0, y1 = 0, ny
    kw = dict(y0=y0, y1=y1, rmax=rmax, disk_spin=dspin)
    f1 = capi.disk_spectrum(capi.image_desc(nx, ny, a, deg2rad(inc), **kw), E, hardening=hard, limb_darkening=limb)
    f2 = capi.disk_spectrum(capi.image_desc(nx, ny, a, deg2rad(inc), **kw), E, hardening=hard, limb_darkening=limb)
    s = capi.disk_spectrum(capi.image_desc(nx, ny, a, deg2rad(inc), strict=True, **kw), E, hardening=hard, limb_darkening=limb)
    msg = []
    if not np.array_equal(f1, f2, equal_nan=True):
        msg.append("two runs differ")
    if not (np.isfinite(f1).all() and np.isfinite(s).all()):
        msg.append("non-finite bins: fast %d strict %d" % (int((~np.isfinite(f1)).sum()), int((~np.isfinite(s)).sum())))
    elif s.max() > 0:
        live = s > 1e-250 * s.max()
        err = float(np.max(np.abs(f1[live] / s[live] - 1)))
        worst = max(worst, err)
        if err > 1e-6:
            j = int(np.argmax(np.abs(f1[live] / s[live] - 1)))
            msg.append("bin error %.2e at E = %.3e (bin / peak = %.1e)" % (err, E[live][j], s[live][j] / s.max()))
        if (f1[~live] > 1e-240 * s.max()).any():
            msg.append("%d bins the strict kernel has at (next to) zero are not" % int((f1[~live] > 1e-240 * s.max()).sum()))
    elif f1.max() > 0:
        msg.append("strict spectrum is zero, fast is not")
    if msg:
        bad += 1
        print("case %d: a=%.17g inc=%.17g %dx%d rows %d..%d rmax=%.17g hard=%.17g limb=%d disk_spin=%.17g ne=%d E 1e%.3f..1e%.3f: %s"
              % (case, a, inc, nx, ny, y0, y1, rmax, hard, limb, dspin, ne, lo, hi, "; ".join(msg)), flush=True)
    if case % 50 == 49:
        print("... %d cases, %d findings, worst bin error %.2e, %.0f s" % (case + 1, bad, worst, time.time() - t0), flush=True)
print("done: %d cases, %d findings, worst bin error %.2e" % (ncases, bad, worst))
