"""ON THE GPU BOX: where along ONE ray of a step-wise job does a variant leave the CPU loop?  The job is run with the step
cap at m = 1 .. the ray's count (both the GPU variant and the checker stop after m calls of raytrace()), and the state after
m calls compared: the first m at which the difference jumps names the call.
usage: python tests/tools/torus_ray_steps.py <seed> <case of fuzz_torus.py> <ray index> [fast|strict]"""
import sys, math, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
from gpuutil import deg2rad
import oraclelib as ol
import gen_golden_access as gga
import test_gpu_raytrace as T
if sys.argv[1] == "c4":            # the C4 job (a = 0.998, i = 70 deg, r0 = 100, torus (8, 2)):  c4 <n> <ray = iy * n + ix> [fast|strict] [m,m,...]
    sys.argv[1:3] = ["0", "-1"] + [sys.argv[3]] if False else sys.argv[1:3]
C4 = sys.argv[1] == "c4"
seed, want, ray = (0, -1, int(sys.argv[3])) if C4 else (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
strict = len(sys.argv) > 4 and sys.argv[4] == "strict"
rng = np.random.default_rng(seed)
if C4:
    a, inc, n, r0, prec, absorb0, tr, tw = 0.998, 70.0, int(sys.argv[2]), 100.0, 1.0, 0.0, 8.0, 2.0
for case in range(want + 1):
    a = float(rng.choice([0.1, 0.3, 0.9, 0.998, rng.uniform(0.01, 0.99)]))
    inc = float(rng.uniform(10.0, 85.0))
    n = int(rng.integers(6, 28))
    r0 = float(rng.uniform(40.0, 200.0))
    prec = float(rng.choice([1.0, 1.0, 0.3, 0.1, 0.03]))
    absorb0 = float(rng.choice([0.0, 0.3]))
    tr, tw = float(rng.uniform(5.0, 12.0)), float(rng.uniform(1.0, 3.0))
rmax = ol.Oracle().r_ms(a) + 8.0
c = ((np.arange(n) + .5) / n - 0.5) * 2.0 * rmax
al, be = np.tile(c, n), np.repeat(c, n)
job = dict(r0=r0, precision=prec, absorb0=absorb0, torus_r=tr, torus_w=tw)
full = gga.torus_rays(ol.ORACLE_SO, "orc_", a, deg2rad(inc), al[ray:ray + 1], be[ray:ray + 1], **job)
total = int(full["steps"][0])
print("ray %d: alpha %.6f beta %.6f, %d calls in the CPU loop" % (ray, al[ray], be[ray], total))
def state(m):
    ref = gga.torus_rays(ol.ORACLE_SO, "orc_", a, deg2rad(inc), al[ray:ray + 1], be[ray:ray + 1], max_steps=m, **job)
    d = T.torus_desc(capi, n, a, inc, max_steps=m, **job)
    if strict:
        d.img.flags = 1
    S, steps, xe, ce, me, ke = T.run_torus(capi, d, full=True)
    e = T.ray_errors({"x_end": xe[ray:ray + 1], "k_end": ke[ray:ray + 1], "I": S[ray:ray + 1, 0], "tau": S[ray:ray + 1, 4]}, ref)
    return max(float(v[0]) for v in e.values()), int(steps[ray]), int(ref["steps"][0]), xe[ray], ref["x_end"][0], {k: float(v[0]) for k, v in e.items()}
lo, hi = 1, total
grid = sorted(set([1, 2, 5, 10, 20, 50, 100, 200, 400] + list(range(100, total, max(total // 40, 1))) + [total]))
if len(sys.argv) > 5: grid = [int(x) for x in sys.argv[5].split(',')]
prev = 0.0
for m in grid:
    if m > total: break
    e, sg, sc, xg, xc, parts = state(m)
    flag = "  <-- jump" if e > 30 * max(prev, 1e-15) and e > 1e-12 else ""
    print("after %5d calls (GPU %d, CPU %d): worst %.2e  r %.10g  cos(theta) %+.6f%s" % (m, sg, sc, e, xc[1], xc[2], flag))
    if len(sys.argv) > 5 and e > 1e-9:
        dx = xg - xc
        print("      GPU - CPU: dt %.3e dr %.3e dcos %.3e dphi %.3e ; per unit of dt: dr/dt %.4f dcos/dt %.4f dphi/dt %.4f" % (
            dx[0], dx[1], dx[2], dx[3], dx[1] / dx[0] if dx[0] else float("nan"), dx[2] / dx[0] if dx[0] else float("nan"), dx[3] / dx[0] if dx[0] else float("nan")))
    if m == 1:
        print("      components after the first call:", {k: "%.1e" % v for k, v in parts.items()}, " GPU x", xg.tolist(), " CPU x", xc.tolist())
    prev = e
