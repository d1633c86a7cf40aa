# ON THE GPU BOX: lane utilisation of the plain image kernel on the upper and the lower half of the headline image
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for r in "0 2047" "2049 4096"; do
  OUT=gpurun_out/prof_rows; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES -- python3 tests/tools/bench_image_rows.py $r > $OUT/p1.log 2>&1
  tail -1 $OUT/p1.log
  python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for d in sorted(glob.glob("gpurun_out/prof_rows/p*/*/*counter_collection.csv")):
    for row in csv.DictReader(open(d)):
        if "disk_image_" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print("  ".join("%s %.4g" % kv for kv in sorted(m.items())), " lane use %.3f" % (m["SQ_THREAD_CYCLES_VALU"] / 64 / m["SQ_ACTIVE_INST_VALU"]))
PY
done
