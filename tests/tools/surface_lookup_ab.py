"""ON THE GPU BOX: the surface search of two builds of the library on the same random jobs, bit for bit (the look-up of the
surface height by a guessed segment, k_surface.hip surface_height_guess, against a build that bisects: the segment found is
the same, so every output word must be).  Each build runs in a child process (SIM5GPU_LIB selects the library).
usage: python tests/tools/surface_lookup_ab.py <lib A> <lib B> [n_cases] [seed]"""
import os, subprocess, sys, math
import numpy as np

def jobs(ncases, seed):
    rng = np.random.default_rng(seed)
    for case in range(ncases):
        a = float(rng.choice([0.0, 0.3, 0.9, 0.998, rng.uniform(0, 0.998)]))
        inc = float(rng.uniform(5.0, 80.0))
        n = int(rng.integers(40, 160))
        rmax = float(rng.uniform(8.0, 60.0))
        nt = int(rng.choice([2, 3, 4, 5, 16, 256, 1000, 4096]))
        r_in = float(rng.uniform(1.5, 6.0))
        power = float(rng.choice([1.0, 1.0, rng.uniform(1.0, 2.0)]))          # equal steps (the guess holds) or growing steps
        tR = np.sort(np.unique(r_in + rng.uniform(20, 200) * np.linspace(0, 1, nt) ** power))
        if rng.integers(0, 3) == 0 and nt > 8:                                 # nearly equal steps: the neighbours decide
            tR[1:-1] += 0.3 * (tR[1] - tR[0]) * np.sin(np.arange(1.0, tR.size - 1))
            tR = np.sort(np.unique(tR))
        slope = float(rng.uniform(0.0, 0.6))
        tH = np.maximum(slope * (tR - tR[0]) * (1.0 + 0.2 * np.sin(tR / rng.uniform(3.0, 30.0)) * rng.integers(0, 2)), 0.0)
        ax = ((np.arange(n) + .5) / n - .5) * 2 * rmax
        al, be = np.meshgrid(ax, ax)
        yield a, inc, tR, tH, al.ravel().copy(), be.ravel().copy()

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
    import sim5_amd.capi as capi
    out = {}
    for ci, (a, inc, tR, tH, al, be) in enumerate(jobs(int(sys.argv[3]), int(sys.argv[4]))):
        for strict in (False, True):
            s = capi.disk_surface_rays(a, inc / 180.0 * math.pi, tR, tH, al, be, strict=strict)
            for k in ("status", "P", "r", "m", "k"):
                out["%d_%d_%s" % (ci, strict, k)] = s[k]
    np.savez(sys.argv[2], **out)
    sys.exit(0)

libs = sys.argv[1:3]
ncases = sys.argv[3] if len(sys.argv) > 3 else "40"
seed = sys.argv[4] if len(sys.argv) > 4 else "11"
files = []
for i, lib in enumerate(libs):
    f = "/tmp/surface_lookup_%d.npz" % i
    subprocess.run([sys.executable, __file__, "--child", f, ncases, seed], check=True, env=dict(os.environ, SIM5GPU_LIB=os.path.abspath(lib)))
    files.append(np.load(f))
A, B = files
rays = hits = 0
diff = {"fast": 0, "strict": 0}
worst = {"fast": 0.0, "strict": 0.0}
for key in A.files:
    ci, strict, name = key.split("_")
    va, vb = A[key], B[key]
    same = (va.view(np.int64) == vb.view(np.int64)) if va.dtype == np.float64 else (va == vb)
    which = "strict" if strict == "1" else "fast"
    if name == "status":
        rays += va.size; hits += int((va == 1).sum())
    diff[which] += int((~same).sum())
    if name == "r" and (~same).any():
        ok = (A["%s_%s_status" % (ci, strict)] == 1) & (B["%s_%s_status" % (ci, strict)] == 1)
        worst[which] = max(worst[which], float(np.nanmax(np.abs(va[ok] / vb[ok] - 1))))
print("surface search, %s jobs, %d rays x 2 variants (%d hits), %s against %s: words that differ: strict %d, fast %d (worst |r/r' - 1| %.1e)"
      % (ncases, rays // 2, hits // 2, os.path.basename(libs[0]), os.path.basename(libs[1]), diff["strict"], diff["fast"], worst["fast"]))
