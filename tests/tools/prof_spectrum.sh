#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_spec; rm -rf $OUT; mkdir -p $OUT
B="python3 tests/tools/bench_jobs.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS -- $B > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc3 --pmc GRBM_GUI_ACTIVE VALUBusy VALUUtilization -- $B > $OUT/pmc3.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/prof_spec/pmc*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(d)):
        if "disk_spectrum_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in acc.items(): print(k, "n=%d"%len(v), "mean=%.6g"%(sum(v)/len(v)))
for f in glob.glob("gpurun_out/prof_spec/stats/*/*kernel_stats.csv"):
    for l in open(f):
        if "spectrum" in l: print(l[:200])
for f in glob.glob("gpurun_out/prof_spec/stats/*/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        if "disk_spectrum_kernel" in row["Kernel_Name"]:
            print({k: row[k] for k in ("VGPR_Count","SGPR_Count","Scratch_Size","LDS_Block_Size","Grid_Size_X","Grid_Size_Y")}); break
PY
