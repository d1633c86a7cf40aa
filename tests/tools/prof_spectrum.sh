#!/bin/bash
# ON THE GPU BOX: per-kernel times and counters of the spectrum job (tests/tools/bench_spectrum.py: three energy counts, the middle
# one 128).   prof_spectrum.sh [log | uniform]   -> gpurun_out/prof_spec (log grid) or gpurun_out/prof_spec_uniform
GRID=${1:-log}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_spec; [ "$GRID" = uniform ] && OUT=gpurun_out/prof_spec_uniform
rm -rf $OUT; mkdir -p $OUT; export S5_PROF_SPEC_OUT=$OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tests/tools/bench_spectrum.py 1024 $GRID > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS -- python3 tests/tools/bench_spectrum.py 1024 $GRID > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc2 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU -- python3 tests/tools/bench_spectrum.py 1024 $GRID > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc3 --pmc GRBM_GUI_ACTIVE VALUBusy VALUUtilization -- python3 tests/tools/bench_spectrum.py 1024 $GRID > $OUT/pmc3.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json, os
OUT = os.environ["S5_PROF_SPEC_OUT"]
out = {}
for d in sorted(glob.glob(OUT + "/pmc*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(d)):
        if "disk_spectrum" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        # three energy counts in the run, 350 launches each: the middle third is the 128-energy job
        n = len(v) // 3
        mid = v[n:2 * n]
        out[k] = {"launches": len(mid), "mean_128_energies": sum(mid) / max(len(mid), 1)}
        print(k, out[k])
for f in glob.glob(OUT + "/stats/*/*kernel_stats.csv"):
    for l in open(f):
        if "spectrum" in l: print(l[:220])
json.dump(out, open(OUT + "/spectrum_pmc.json", "w"), indent=1)
PY
