# ON THE GPU BOX: per-kernel time and counters of the surface search job (tests/tools/bench_surface.py).
#   gpurun -- bash tests/tools/prof_surface.sh [variant]
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
[ -n "$1" ] && export SIM5GPU_LIB=$GRAFT_REPO_ROOT/sim5_amd/lib/ab_$1.so      # the in-tree library stays as it is
OUT=gpurun_out/prof_surf; rm -rf $OUT; mkdir -p $OUT
B="python3 tests/tools/bench_surface.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B > $OUT/kt.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU -- $B > $OUT/pmc1.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc3 --pmc GRBM_GUI_ACTIVE VALUBusy VALUUtilization -- $B > $OUT/pmc3.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/prof_surf/kt/*/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        if "surface" in row["Name"]: print(row["Name"][:60], row["Calls"], "avg_us=%.1f" % (float(row["AverageNs"]) / 1e3), "total_ms=%.2f" % (float(row["TotalDurationNs"]) / 1e6))
for d in sorted(glob.glob("gpurun_out/prof_surf/pmc*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(d)):
        if "surface" in row["Kernel_Name"] and "s5f" in row["Kernel_Name"]:
            acc[(row["Kernel_Name"].split("(")[0][-24:], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()): print(k[0], k[1], "n=%d" % len(v), "max=%.6g" % max(v), "mean=%.6g" % (sum(v) / len(v)))
PY
