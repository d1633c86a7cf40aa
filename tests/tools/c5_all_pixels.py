"""ON THE GPU BOX: BASELINE.json configs[4] (8192 x 8192, a = 0.998, inclinations 10 .. 80 deg) with EVERY pixel held to the unmodified
reference run live on the box's host cores (the suite holds every 64th pixel in x and y and the hit counts): classes, r and g of both
variants.  537 M rays on the CPU: ~30 s of 16 cores.
usage: python tests/tools/c5_all_pixels.py [n] [inclinations, comma separated]"""
import sys, os, json, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi, oraclelib as ol
from gpuutil import deg2rad
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
incs = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [10., 20., 30., 40., 50., 60., 70., 80.]
a = 0.998
out = {"job": "C5 %d x %d, a = %g" % (n, n, a), "per_inclination": {}}
tot_px = 0
for inc in incs:
    t0 = time.time()
    ref = ol.cpu_disk_image("reference", n, n, a, inc, nthreads=min(16, os.cpu_count() or 1), full=True)
    t_cpu = time.time() - t0
    rec = {"pixels": n * n, "reference_wall_s": round(t_cpu, 1), "reference_hits": int(np.isin(ref["cls"], (2, 4)).sum())}
    hit = np.isin(ref["cls"], (2, 4))
    for strict in (False, True):
        g = capi.disk_image(capi.image_desc(n, n, a, deg2rad(inc), strict=strict), full=True)
        d = {"class_differences": int((g["cls"] != ref["cls"]).sum())}
        same = hit & (g["cls"] == ref["cls"])
        er = np.abs(g["r"][same] / ref["r"][same] - 1); eg = np.abs(g["g"][same] / ref["g"][same] - 1)
        d.update({"worst_r": float(er.max()), "worst_g": float(eg.max()), "pixels_r_above_1e-6": int((er > 1e-6).sum()), "pixels_r_above_1e-9": int((er > 1e-9).sum()),
                  "pixels_g_above_1e-6": int((eg > 1e-6).sum())})
        rec["strict" if strict else "fast"] = d
        del g
    out["per_inclination"]["%g" % inc] = rec
    tot_px += n * n
    print("i = %g deg: %s" % (inc, json.dumps(rec)), flush=True)
    del ref
out["pixels_total"] = tot_px
if os.path.isdir("gpurun_out"):
    json.dump(out, open(os.path.join("gpurun_out", "c5_all_pixels.json"), "w"), indent=1)
