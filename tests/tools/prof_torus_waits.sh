#!/bin/bash
# ON THE GPU BOX: where the march kernel's waves wait (one counter group per run; --pmc never combined with other trace domains).
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_torus_waits; rm -rf $OUT; mkdir -p $OUT
B="python3 tests/tools/bench_torus.py"
rocprofv3 -L > $OUT/counters.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- $B > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_IFETCH -- $B > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_WAIT_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_WAVES -- $B > $OUT/p3.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for d in sorted(glob.glob("gpurun_out/prof_torus_waits/p*/*/*counter_collection.csv")):
    for row in csv.DictReader(open(d)):
        if "torus_pool" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, "n=%d" % len(v), "mean=%.6g" % (sum(v) / len(v)))
PY
grep -i "error\|invalid\|not found" $OUT/p*.log | head
