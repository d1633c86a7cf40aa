"""Randomised campaign for the batch geodesic entry points (run on the GPU box; not part of the suite): random spins,
inclinations and impact parameters; sim5gpu_geodesic_init_inf / _find_midplane_crossing / _position_rad / _position_pol /
_P_int against the CPU checker call by call: outcome, class and number of real roots equal, every field and value within 1e-9
(1e-6 for quantities that go through an inverse Jacobi function next to its singular end).
usage: python tests/tools/fuzz_geodesic.py [n_rays] [seed]"""
import sys, math, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
import oracle_capi as oc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
a = rng.choice([0.0, 1e-6, 0.3, 0.9, 0.998, 0.999999], n) * (rng.random(n) < 0.7) + rng.uniform(0, 0.999999, n) * 0
a = np.where(rng.random(n) < 0.3, rng.uniform(0, 0.999999, n), a)
inc = np.radians(rng.uniform(1.0, 89.0, n))
al = rng.normal(0, 8, n) * (rng.random(n) < 0.9) + rng.normal(0, 200, n) * 0
be = rng.normal(0, 8, n)
big = rng.random(n) < 0.05
al = np.where(big, rng.normal(0, 300, n), al); be = np.where(big, rng.normal(0, 300, n), be)
t0 = time.time()
G, eg, okg = capi.geodesic_init_inf(inc, a, al, be)
O, eo, oko = oc.geodesic_init_inf(inc, a, al, be)
bad = 0
def report(name, cond, extra=""):
    global bad
    k = int(np.count_nonzero(cond))
    if k:
        bad += 1
        i = np.nonzero(cond)[0][0]
        print("%s: %d rays; first: a=%.8g inc=%.3f alpha=%.6g beta=%.6g %s" % (name, k, a[i], math.degrees(inc[i]), al[i], be[i], extra))
# alpha = 0 exactly (10 % of the sample on purpose): l = 0, the degenerate set of the algorithm -- counted apart
zero = (al == 0.0)
print("alpha = 0 rays: %d, outcome differs on %d of them" % (int(zero.sum()), int(((okg != oko) & zero).sum())))
report("outcome differs", (okg != oko) & ~zero)
report("error code differs", (eg != eo) & ~zero)
both = (okg != 0) & (oko != 0) & ~zero
report("class differs", both & (G["type"] != O["type"]))
report("nrr differs", both & (G["nrr"] != O["nrr"]))
ok = both & (G["type"] == O["type"])
def rel(x, y, floor): return np.abs(x - y) / np.maximum(np.abs(y), floor)
for f, tol, floor in (("l", 1e-12, 1e-3), ("q", 1e-12, 1e-3), ("m2p", 1e-9, 1e-6), ("m2m", 1e-9, 1e-6), ("mm", 1e-9, 1e-6), ("mK", 1e-9, 1e-6),
                      ("rp", 1e-9, 1e-3), ("Rpc", 1e-9, 1e-6), ("Tpp", 1e-9, 1e-6), ("Tip", 1e-6, 1e-6)):
    e = rel(G[f][ok], O[f][ok], floor)
    e = np.where(np.isnan(G[f][ok]) & np.isnan(O[f][ok]), 0.0, e)
    w = np.nanmax(e) if e.size else 0.0
    idx = np.nonzero(ok)[0]
    report("field %s off by up to %.1e" % (f, w), np.isin(np.arange(n), idx[(e > tol) | np.isnan(e)]))
# crossing, r(P), mu(P), P(r) on the valid geodesics of the CHECKER's records (same input to both sides)
v = np.nonzero(ok)[0][:6000]
rec = O[v]
for order in (0, 1):
    pg = capi.geodesic_find_midplane_crossing(rec, order); po = oc.geodesic_find_midplane_crossing(rec, order)
    same_nan = np.isnan(pg) == np.isnan(po)
    e = np.where(np.isnan(po), 0.0, rel(pg, po, 1e-6))
    print("midplane crossing order %d: NaN pattern equal %s, worst %.1e" % (order, bool(same_nan.all()), np.nanmax(e)))
    bad += (not same_nan.all()) or np.nanmax(e) > 1e-6
P = rec["Rpc"] * rng.uniform(0.05, 1.95, v.size)
for name, fg, fo in (("position_rad", capi.geodesic_position_rad, oc.geodesic_position_rad), ("position_pol", capi.geodesic_position_pol, oc.geodesic_position_pol)):
    xg, xo = fg(rec, P), fo(rec, P)
    same_nan = np.isnan(xg) == np.isnan(xo)
    e = np.where(np.isnan(xo), 0.0, rel(xg, xo, 1e-3))
    print("%s: NaN pattern equal %s, worst %.1e" % (name, bool(same_nan.all()), np.nanmax(e)))
    bad += (not same_nan.all()) or np.nanmax(e) > 1e-7
rr = rec["rp"] * rng.uniform(1.01, 50.0, v.size)
pg, po = capi.geodesic_P_int(rec, rr, 0), oc.geodesic_P_int(rec, rr, 0)
e = np.where(np.isnan(po), 0.0, rel(pg, po, 1e-9))
print("P_int: NaN pattern equal %s, worst %.1e" % (bool((np.isnan(pg) == np.isnan(po)).all()), np.nanmax(e)))
bad += np.nanmax(e) > 1e-7
print("%d rays (%d valid), %d findings, %.0f s" % (n, int(ok.sum()), bad, time.time() - t0))
