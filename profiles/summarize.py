#!/usr/bin/env python3
"""Turn the raw rocprofv3 CSVs that profiles/collect.sh left under gpurun_out/prof_<tag>/ into the
committed summaries profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json and profiles/traffic.json
(HBM bytes per launch of the image kernel, read by bench.py for roofline.traffic).

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB,
collected in separate passes; on gfx950 FETCH_SIZE under-reports wide streaming reads by 2x -- this
kernel reads no input at all (kernel arguments only), so the read side is negligible either way and
is reported both raw and doubled."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
# the headline image is traced by the mirror kernel since round 2 (a lane takes a ray and its mirror image in beta);
# `tag`s collected before that hold the grid kernel
# ... and since round 4 by the same routine reading its job from the argument segment (disk_image_jobs_kernel)
KERNEL = "disk_image_jobs_kernel"
for _d in glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")):
    _txt = open(_d).read()
    if "disk_image_jobs_kernel" not in _txt:
        KERNEL = "disk_image_mirror_kernel" if "disk_image_mirror_kernel" in _txt else "disk_image_grid_kernel"

def parent_file(pattern):
    """The program under rocprofv3 starts child processes (the C programs of the scalar-API legs) and every process leaves its
    own CSVs: the file of the PARENT is the one that holds the image kernel -- chosen by kernel name, not by position in a
    directory listing (round 5's default-command summary was a child's: VERDICT r5 weak 7)."""
    found = [f for f in sorted(glob.glob(pattern)) if KERNEL in open(f).read()]
    return found[0] if found else None


TIMED_STEPS = 10          # profiles/collect.sh: bench.py --steps 10 --warmup 2 (the LAST launches of the image kernel in that trace)
stats = parent_file(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
trace_file = parent_file(os.path.join(src, "stats", "*", "*kernel_trace.csv"))
if stats:
    rows = list(csv.reader(open(stats)))
    with open(os.path.join(root, "profiles", tag + "_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        for r in rows[:12]:
            w.writerow([c[:160] for c in r])
        if trace_file:
            # the same statistic over the launches of the TIMED region only (the all-launch row above averages ~700 spin-up
            # launches and the three cold-clock ones with them)
            d_all = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(trace_file)) if KERNEL in r["Kernel_Name"]]
            d_t = d_all[-TIMED_STEPS:]
            if d_t:
                mean = sum(d_t) / len(d_t)
                sd = (sum((x - mean) ** 2 for x in d_t) / max(len(d_t) - 1, 1)) ** 0.5
                w.writerow(["%s -- TIMED REGION ONLY (the last %d launches of this trace: bench.py --steps %d)" % (KERNEL, len(d_t), TIMED_STEPS),
                            len(d_t), sum(d_t), "%.6f" % mean, "", min(d_t), max(d_t), "%.6f" % sd])

pmc = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv"))):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(d)):
        if KERNEL in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        pmc[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
trace = [trace_file] if trace_file else []
dur = []
meta = {}
if trace:
    for row in csv.DictReader(open(trace[0])):
        if KERNEL in row["Kernel_Name"]:
            dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            meta = {k: row[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size",
                                        "Workgroup_Size_X", "Grid_Size_X", "Grid_Size_Y") if k in row}
out = {"kernel": KERNEL, "pmc": pmc, "dispatch": meta,
       "kernel_ns_avg": sum(dur) / len(dur) if dur else None, "kernel_launches": len(dur),
       "kernel_ns_avg_timed_region": (sum(dur[-TIMED_STEPS:]) / len(dur[-TIMED_STEPS:])) if dur else None,
       "kernel_launches_timed_region": len(dur[-TIMED_STEPS:])}
rays = 4096 * 4096
if "SQ_INSTS_VALU" in pmc:
    out["valu_wave_instructions_per_ray_lane"] = pmc["SQ_INSTS_VALU"]["mean_per_launch"] / (rays / 64)
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    fetch = pmc["FETCH_SIZE"]["mean_per_launch"] * 1024.0
    write = pmc["WRITE_SIZE"]["mean_per_launch"] * 1024.0
    out["hbm"] = {"fetch_bytes_raw": fetch, "fetch_bytes_x2_gfx950": 2 * fetch, "write_bytes": write,
                  "algorithmic_bytes": rays * 8}
    tj = {"hbm_bytes_per_launch": 2 * fetch + write, "source": "profiles/%s_pmc.json" % tag,
          "note": "2*FETCH_SIZE + WRITE_SIZE, KiB->B, per launch of " + KERNEL}
    # FP64 operations the kernel EXECUTES (wave instructions x 64 lanes; an FMA counts 2; the quarter-rate seeds rcp / rsq /
    # sqrt are listed separately): what roofline.executed_frac in bench.py is computed from
    if all(k in pmc for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64")):
        add, mul, fma = (pmc[k]["mean_per_launch"] for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
        tj["executed_fp64_flops_per_launch"] = 64.0 * (add + mul + 2.0 * fma)
        tj["executed_fp64_flops_per_ray"] = 64.0 * (add + mul + 2.0 * fma) / rays
        tj["executed_fp64_wave_instructions_per_64_rays"] = {"add": add / (rays / 64), "mul": mul / (rays / 64), "fma": fma / (rays / 64),
                                                            "trans": pmc.get("SQ_INSTS_VALU_TRANS_F64", {}).get("mean_per_launch", 0.0) / (rays / 64)}
        tj["valu_wave_instructions_per_64_rays"] = pmc["SQ_INSTS_VALU"]["mean_per_launch"] / (rays / 64) if "SQ_INSTS_VALU" in pmc else None
        tj["kernel"] = KERNEL
    json.dump(tj, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
# the default bench command (headline + the other configurations): per-kernel statistics as rocprofv3 prints them, and the
# headline launches picked out of its kernel trace by their grid (256 x 256 workgroups of 256 threads: X = 65536, Y = 256;
# the mirror kernel covers the upper half: Y = 128; the job-list kernel has a one-dimensional grid of 256 x 128 tiles of 256
# threads = 8388608, and the headline runs its single-job instantiation)
dstats = parent_file(os.path.join(src, "stats_default_cmd", "*", "*kernel_stats.csv"))
if dstats:
    rows = list(csv.reader(open(dstats)))
    with open(os.path.join(root, "profiles", tag + "_kernel_stats_default_cmd.csv"), "w") as f:
        w = csv.writer(f)
        for r in rows[:16]:
            w.writerow([c[:160] for c in r])
dtrace = parent_file(os.path.join(src, "stats_default_cmd", "*", "*kernel_trace.csv"))
if dtrace:
    hd = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(dtrace))
          if KERNEL in r["Kernel_Name"] and (
              (r.get("Grid_Size_X") == "8388608" and "ILb1E" in r["Kernel_Name"] + "ILb1E" * ("<true>" in r["Kernel_Name"])) if "jobs" in KERNEL else
              (r.get("Grid_Size_X") == "65536" and r.get("Grid_Size_Y") == ("128" if "mirror" in KERNEL else "256")))]
    out["default_cmd_headline_launches"] = {"launches": len(hd), "kernel_ns_avg": sum(hd) / len(hd) if hd else None}
    assert hd, "the default-command trace of the parent process holds no headline launch: wrong file?"
dl = os.path.join(src, "stats_default_cmd.log")
if os.path.exists(dl):
    for line in open(dl):
        if line.startswith("{") and '"metric"' in line:
            try:
                out["bench_default_cmd_under_rocprof"] = json.loads(line)
            except Exception:
                pass
bj = os.path.join(src, "bench_under_rocprof.json")
if os.path.exists(bj):
    try:
        out["bench_under_rocprof"] = json.load(open(bj))
    except Exception:
        pass
json.dump(out, open(os.path.join(root, "profiles", tag + "_pmc.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "bench_under_rocprof"}, indent=1)[:3000])
