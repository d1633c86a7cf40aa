"""One case of tests/tools/fuzz_torus.py in detail (run on the GPU box): the rays of either variant whose end state differs from
the CPU loop's by more than 1e-6, each beside the CHECKER's own sensitivity to +-1 ulp of its start state (the exemption rule
of tests/test_gpu_raytrace.py compare_rays: a ray passes only if the checker's own result moves by at least as much).
usage: python tests/tools/fuzz_torus_case.py <seed> <case>"""
import sys, math, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
from gpuutil import deg2rad
import oraclelib as ol
import gen_golden_access as gga
import test_gpu_raytrace as T
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(want + 1):
    a = float(rng.choice([0.1, 0.3, 0.9, 0.998, rng.uniform(0.01, 0.99)]))
    inc = float(rng.uniform(10.0, 85.0))
    n = int(rng.integers(6, 28))
    r0 = float(rng.uniform(40.0, 200.0))
    prec = float(rng.choice([1.0, 1.0, 0.3, 0.1, 0.03]))
    absorb0 = float(rng.choice([0.0, 0.3]))
    tr, tw = float(rng.uniform(5.0, 12.0)), float(rng.uniform(1.0, 3.0))
print("case %d: a=%r inc=%r n=%d r0=%r precision=%g absorb=%g torus(%r, %r)" % (want, a, inc, n, r0, prec, absorb0, tr, tw))
rmax = ol.Oracle().r_ms(a) + 8.0
c = ((np.arange(n) + .5) / n - 0.5) * 2.0 * rmax
al, be = np.tile(c, n), np.repeat(c, n)
job = dict(r0=r0, precision=prec, absorb0=absorb0, torus_r=tr, torus_w=tw)
ref = gga.torus_rays(ol.ORACLE_SO, "orc_", a, deg2rad(inc), al, be, **job)
for strict in (True, False):
    d = T.torus_desc(capi, n, a, inc, **job)
    if strict:
        d.img.flags = 1
    S, steps, xe, ce, me, ke = T.run_torus(capi, d, full=True)
    tag = "strict" if strict else "fast"
    same = (steps == ref["steps"])
    m = same & (steps > 0) & (al != 0.0)
    errs = T.ray_errors({"x_end": xe, "k_end": ke, "I": S[:, 0], "tau": S[:, 4]}, ref)
    tot = np.max(np.stack(list(errs.values())), axis=0)
    over = np.nonzero(m & (tot > 1e-6))[0]
    print("%s: %d of %d step counts equal; rays above 1e-6: %d" % (tag, int(same.sum()), same.size, over.size))
    if over.size:
        kap = T.oracle_sensitivity(ol.ORACLE_SO, "orc_", a, deg2rad(inc), al[over], be[over],
                                   {k: (v[over] if hasattr(v, "__len__") and len(v) == al.size else v) for k, v in ref.items()}, **job)
        for i, k in zip(over, kap):
            print("   ray alpha %.4f beta %.4f: %d steps (median %d), r_end %.9g; differs by %.2e (%s); the checker's own +-1 ulp sensitivity %.2e -> %s" % (
                al[i], be[i], steps[i], int(np.median(ref["steps"])), xe[i, 1], tot[i],
                ", ".join("%s %.1e" % (kk, vv[i]) for kk, vv in errs.items() if vv[i] > 1e-7), k, "excused" if tot[i] <= k else "NOT EXCUSED"))
