cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_surf2; rm -rf $OUT; mkdir -p $OUT
B="python3 tests/tools/bench_surface.py"
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS -- $B > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc3 --pmc GRBM_GUI_ACTIVE VALUBusy VALUUtilization -- $B > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc4 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 -- $B > $OUT/pmc4.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/prof_surf2/pmc*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(d)):
        if "disk_surface_kernel" in row["Kernel_Name"] and "s5f" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in acc.items(): print(k, "n=%d"%len(v), "mean=%.6g"%(sum(v)/len(v)))
PY
