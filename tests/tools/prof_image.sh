# ON THE GPU BOX: counters of the headline image kernel for one or more prebuilt variants (tests/tools/ab_build.sh).
#   gpurun -- bash tests/tools/prof_image.sh name1 [name2 ...]
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python3 tests/tools/bench_image.py"
for name in "$@"; do
  export SIM5GPU_LIB=$GRAFT_REPO_ROOT/sim5_amd/lib/ab_$name.so      # the in-tree library stays as it is
  OUT=gpurun_out/prof_img_$name; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVES -- $B > $OUT/p1.log 2>&1 &&
  rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc GRBM_GUI_ACTIVE VALUBusy VALUUtilization -- $B > $OUT/p2.log 2>&1
  python3 - $name <<'PY'
import csv, glob, collections, sys
name = sys.argv[1]
acc = collections.defaultdict(list)
for d in sorted(glob.glob("gpurun_out/prof_img_%s/p*/*/*counter_collection.csv" % name)):
    for row in csv.DictReader(open(d)):
        if "disk_image_" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== %s: " % name + "  ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(acc.items())))
PY
done
