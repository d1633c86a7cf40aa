"""Randomised campaign for the surface search (run on the GPU box; not part of the suite): the fast variant (walk by
addition theorems between anchors, short ladders + side stream) against the strict one (full evaluation in the reference's
operation order at every sub-step) on random spins, inclinations, fields of view and surface tables.
usage: python tests/tools/fuzz_surface.py [n_cases] [seed]"""
import sys, math, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
from gpuutil import deg2rad

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
bad = 0
t0 = time.time()
for case in range(ncases):
    a = float(rng.choice([0.0, 0.3, 0.9, 0.998, rng.uniform(0, 0.998)]))
    inc = float(rng.uniform(5.0, 80.0))
    n = int(rng.integers(40, 160))
    rmax = float(rng.uniform(8.0, 60.0))
    nt = int(rng.choice([2, 16, 256, 1000]))
    r_in = float(rng.uniform(1.5, 6.0))
    tR = np.sort(np.unique(r_in + (rng.uniform(20, 200)) * np.linspace(0, 1, nt) ** rng.uniform(1.0, 2.0)))
    slope = float(rng.uniform(0.0, 0.6))
    tH = slope * (tR - tR[0]) * (1.0 + 0.2 * np.sin(tR / rng.uniform(3.0, 30.0)) * rng.integers(0, 2))
    tH = np.maximum(tH, 0.0)
    ax = ((np.arange(n) + .5) / n - .5) * 2 * rmax
    al, be = np.meshgrid(ax, ax)
    f = capi.disk_surface_rays(a, deg2rad(inc), tR, tH, al.ravel(), be.ravel(), strict=False)
    s = capi.disk_surface_rays(a, deg2rad(inc), tR, tH, al.ravel(), be.ravel(), strict=True)
    same = f["status"] == s["status"]
    ok = same & (s["status"] == 1)
    msg = []
    if same.mean() < 0.999:
        msg.append("status differs on %.3f %%" % (100 * (1 - same.mean())))
    if ok.any():
        dr = np.abs(f["r"][ok] / s["r"][ok] - 1); dm = np.abs(f["m"][ok] - s["m"][ok]); dP = np.abs(f["P"][ok] / s["P"][ok] - 1)
        near = (dr < 1e-8) & (dm < 1e-8) & (dP < 1e-8)
        if near.mean() < 0.998:
            msg.append("only %.3f %% of the hits within 1e-8 (worst r %.1e)" % (100 * near.mean(), dr.max()))
        R = f["r"][ok] * np.sqrt(1 - f["m"][ok] ** 2); H = f["r"][ok] * f["m"][ok]
        inside = R < tR[-1]
        if inside.any():
            off = np.abs(H - np.interp(R, tR, tH))[inside].max()
            if off > 3e-2:
                msg.append("hit %.2e off the surface" % off)
    print("case %3d a=%.4g inc=%.1f n=%d rmax=%.1f table %d pts [%.1f, %.0f] slope %.2f hits %d : %s" % (
        case, a, inc, n, rmax, tR.size, tR[0], tR[-1], slope, int(ok.sum()), "ok" if not msg else "; ".join(msg)), flush=True)
    bad += bool(msg)
print("%d cases, %d with findings, %.0f s" % (ncases, bad, time.time() - t0))
