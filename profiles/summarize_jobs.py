#!/usr/bin/env python3
"""Raw rocprofv3 CSVs of profiles/collect_jobs.sh (gpurun_out/prof_jobs_<tag>/) -> profiles/<tag>_jobs_summary.json:
per kernel the launch count, mean duration (kernel trace), registers / scratch / LDS of the dispatch, and the mean
of every collected counter per launch, plus a few derived figures (VALU instructions per lane-step or per ray where
the unit count is known).  FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md, HBM section); FETCH_SIZE is
reported raw and doubled (gfx950 under-reports wide streaming reads by 2x)."""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_jobs_" + tag)


def short(name):
    m = re.match(r"(?:void )?((?:s5f?|s5abi)::)?([A-Za-z0-9_]+)", name)
    ns = (m.group(1) or "") if m else ""
    return (ns + m.group(2)) if m else name[:60]


kern = collections.OrderedDict()
for f in glob.glob(os.path.join(src, "stats", "*", "*kernel_trace.csv")):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        e = kern.setdefault(k, {"launches": 0, "ns": [], "dispatch": {}})
        e["launches"] += 1
        e["ns"].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        e["dispatch"] = {c: row[c] for c in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size",
                                             "Workgroup_Size_X", "Grid_Size_X", "Grid_Size_Y") if c in row}
for d in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv"))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    per_dispatch = collections.defaultdict(lambda: collections.defaultdict(dict))
    for row in csv.DictReader(open(d)):
        acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        per_dispatch[short(row["Kernel_Name"])][row["Dispatch_Id"]][row["Counter_Name"]] = float(row["Counter_Value"])
    # VALUBusy / VALUUtilization are per-launch percentages: a kernel launched several times per job, most launches finding nothing
    # to do (the surface search's walk rounds 2-4), has a plain mean that says nothing -- weight them by the launch's active cycles
    for k, ds in per_dispatch.items():
        rows = [v for v in ds.values() if "GRBM_GUI_ACTIVE" in v and "VALUBusy" in v]
        tot = sum(v["GRBM_GUI_ACTIVE"] for v in rows)
        if rows and tot > 0:
            e = kern.setdefault(k, {"launches": 0, "ns": [], "dispatch": {}})
            e["pmc_weighted_by_active_cycles"] = {c: sum(v[c] * v["GRBM_GUI_ACTIVE"] for v in rows) / tot
                                                  for c in ("VALUBusy", "VALUUtilization") if all(c in v for v in rows)}
    for k, cs in acc.items():
        e = kern.setdefault(k, {"launches": 0, "ns": [], "dispatch": {}})
        for c, v in cs.items():
            e.setdefault("pmc_mean_per_launch", {})[c] = sum(v) / len(v)
            # tests/tools/bench_jobs.py runs the march twice: precision 1 (the C4 job) first, then 0.01 -- the first half of
            # the launches, in dispatch order, is the C4 job by itself
            if "torus_pool" in k and len(v) >= 2:
                h = v[:len(v) // 2]
                e.setdefault("pmc_mean_per_launch_c4_precision_1", {})[c] = sum(h) / len(h)
out = {}
for k, e in kern.items():
    if not any(s in k for s in ("disk_", "torus_", "spectrum", "map_rays", "surface_")):
        continue
    ns = e.pop("ns")
    e["kernel_ns_avg"] = sum(ns) / len(ns) if ns else None
    e["kernel_ns_min"] = min(ns) if ns else None
    p = e.get("pmc_mean_per_launch", {})
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        e["hbm_bytes_per_launch"] = {"fetch_raw": p["FETCH_SIZE"] * 1024, "fetch_x2_gfx950": 2 * p["FETCH_SIZE"] * 1024,
                                     "write": p["WRITE_SIZE"] * 1024}
    out[k] = e
txt = os.path.join(src, "bench_jobs_under_rocprof.txt")
if os.path.exists(txt):
    out["_program_output_under_rocprof"] = [l.rstrip() for l in open(txt) if ("ms" in l and ":" in l)][:20]
with open(os.path.join(root, "profiles", tag + "_jobs_summary.json"), "w") as fh:
    json.dump(out, fh, indent=1)
for k, e in out.items():
    if k.startswith("_"):
        continue
    p = e.get("pmc_weighted_by_active_cycles", e.get("pmc_mean_per_launch", {}))
    print("%-40s n=%-3d %10.3f ms  VGPR %s scratch %s LDS %s  VALUBusy %s VALUUtil %s (weighted by active cycles)" % (
        k, e["launches"], (e["kernel_ns_avg"] or 0) / 1e6, e["dispatch"].get("VGPR_Count"), e["dispatch"].get("Scratch_Size"),
        e["dispatch"].get("LDS_Block_Size"), "%.1f" % p["VALUBusy"] if "VALUBusy" in p else "-",
        "%.1f" % p["VALUUtilization"] if "VALUUtilization" in p else "-"))
