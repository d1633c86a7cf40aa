"""ON THE GPU BOX: the C4 job (BASELINE.json configs[3]: 1024 x 1024 rays through the torus, a = 0.998, i = 70 deg) with EVERY ray held
to the CPU checker's raytrace() loop -- the suite holds every 16th pixel (4 096 rays of tests/golden/torus_c4.npz) and step counts
on random sets.  The CPU loop runs in 16 processes over row blocks (oracle/cpu_driver.c:cpu_torus_rays, ~70 core-seconds).
usage: python tests/tools/c4_all_rays.py [n] [absorb0]"""
import sys, math, time, json, os, numpy as np
from multiprocessing import Pool
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
absorb0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
A, INC, R0 = 0.998, 70.0, 100.0

def cpu_block(rows):
    import oraclelib as ol, gen_golden_access as gga
    rmax = ol.Oracle().r_ms(A) + 8.0
    c = ((np.arange(n) + .5) / n - 0.5) * 2.0 * rmax
    al = np.tile(c, len(rows)); be = np.repeat(c[rows], n)
    o = gga.torus_rays(ol.ORACLE_SO, "orc_", A, INC / 180.0 * math.pi, al, be, r0=R0, absorb0=absorb0)
    return rows[0], {k: o[k] for k in ("steps", "x_end", "k_end", "I", "tau")}

if __name__ == "__main__":
    t0 = time.time()
    blocks = [np.arange(r, min(r + 8, n)) for r in range(0, n, 8)]
    with Pool(min(16, os.cpu_count() or 1)) as pool:
        parts = dict(pool.imap_unordered(cpu_block, blocks))
    ref = {k: np.concatenate([parts[b[0]][k] for b in blocks]) for k in ("steps", "x_end", "k_end", "I", "tau")}
    t_cpu = time.time() - t0
    import sim5_amd.capi as capi
    import test_gpu_raytrace as T
    out = {"job": "C4 %d x %d, absorb0 = %g" % (n, n, absorb0), "rays": n * n, "cpu_loop_wall_s": round(t_cpu, 1), "raytrace_calls_cpu": int(ref["steps"].sum())}
    for strict in (True, False):
        d = T.torus_desc(capi, n, A, INC, r0=R0, absorb0=absorb0)
        if strict:
            d.img.flags = 1
        S, steps, xe, ce, me, ke = T.run_torus(capi, d, full=True)
        same = steps == ref["steps"]
        m = same & (steps > 0)
        errs = T.ray_errors({"x_end": xe, "k_end": ke, "I": S[:, 0], "tau": S[:, 4]}, ref)
        tot = np.max(np.stack([np.nan_to_num(v, nan=0.0) for v in errs.values()]), axis=0)
        rec = {"rays_with_another_call_count": int((~same).sum()), "rays_started": int((steps > 0).sum()),
               "unstarted_equal_to_cpu": bool(np.array_equal(steps == 0, ref["steps"] == 0)),
               "worst_per_component": {k: float(np.nan_to_num(v[m], nan=0.0).max()) for k, v in errs.items()},
               "rays_above_1e-6": int((m & (tot > 1e-6)).sum()), "rays_above_1e-7": int((m & (tot > 1e-7)).sum()), "rays_above_1e-8": int((m & (tot > 1e-8)).sum())}
        over = np.nonzero(m & (tot > 1e-6))[0]
        # the suite's rule for such a ray (test_gpu_raytrace.py compare_rays): it passes only if the CHECKER's own end state moves by
        # at least as much when its start state is changed by +-1 ulp (oracle_sensitivity)
        if over.size:
            import oraclelib as ol
            rmax = ol.Oracle().r_ms(A) + 8.0
            c = ((np.arange(n) + .5) / n - 0.5) * 2.0 * rmax
            kap = T.oracle_sensitivity(ol.ORACLE_SO, "orc_", A, INC / 180.0 * math.pi, c[over % n], c[over // n],
                                       {k: v[over] for k, v in ref.items()}, r0=R0, absorb0=absorb0)
            kap = kap * (1.0 + 1e-6)          # (a ray whose perturbed checker run lands exactly on the variant's result is covered)
            rec["above_1e-6_covered_by_the_checkers_own_sensitivity"] = int((tot[over] <= kap).sum())
            rec["above_1e-6_not_covered"] = int((tot[over] > kap).sum())
            nc = over[tot[over] > kap]
            rec["worst_not_covered"] = float(tot[nc].max()) if nc.size else 0.0
            rec["largest_|alpha|_of_a_ray_above_1e-6"] = float(np.abs(c[over % n]).max()); rec["pixel_size"] = float(c[1] - c[0])
            far = over[np.abs(c[over % n]) > 0.1]
            rec["above_1e-6_with_|alpha|>0.1"] = int(far.size); rec["worst_with_|alpha|>0.1"] = float(tot[far].max()) if far.size else 0.0
            rec["columns_of_the_rays_above_1e-6 (ix: count)"] = {int(k): int(v) for k, v in zip(*np.unique(over % n, return_counts=True))}
            rec["not_covered"] = [{"ix": int(i % n), "iy": int(i // n), "calls": int(steps[i]), "difference": float(tot[i]), "checker_sensitivity": float(kk)}
                                  for i, kk in sorted(zip(over, kap), key=lambda t: -tot[t[0]]) if tot[i] > kk][:30]
        rec["the_rays_above_1e-6"] = [{"ix": int(i % n), "iy": int(i // n), "calls": int(steps[i]), "difference": float(tot[i])} for i in over[:10]]
        out["strict" if strict else "fast"] = rec
        print("strict" if strict else "fast", json.dumps({k: v for k, v in rec.items() if k not in ("the_rays_above_1e-6", "not_covered")})[:1500], flush=True)
    path = os.path.join("gpurun_out", "c4_all_rays_absorb%g.json" % absorb0)
    if os.path.isdir("gpurun_out"):
        json.dump(out, open(path, "w"), indent=1)
