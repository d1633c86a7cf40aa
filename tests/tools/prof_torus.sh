#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_torus; rm -rf $OUT; mkdir -p $OUT
B="python3 tests/tools/bench_torus.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR -- $B > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc2 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INSTS_FLAT -- $B > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc3 --pmc GRBM_GUI_ACTIVE VALUBusy VALUUtilization -- $B > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc4 --pmc WRITE_SIZE -- $B > $OUT/pmc4.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc5 --pmc FETCH_SIZE -- $B > $OUT/pmc5.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/prof_torus/pmc*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(d)):
        if "torus_pool" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in acc.items():
        print(k, "n=%d"%len(v), "mean=%.6g"%(sum(v)/len(v)))
for f in glob.glob("gpurun_out/prof_torus/stats/*/*kernel_stats.csv"):
    print(open(f).read()[:600])
PY
