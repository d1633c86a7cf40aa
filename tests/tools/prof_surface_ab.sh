#!/bin/bash
# ON THE GPU BOX: per-kernel times of the surface job (tests/tools/bench_surface.py) for each prebuilt variant library
#   gpurun -- bash tests/tools/prof_surface_ab.sh base cnt      -> gpurun_out/prof_surface_ab/<name>.txt
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_surface_ab; rm -rf $OUT; mkdir -p $OUT
for name in "$@"; do
  export SIM5GPU_LIB=$GRAFT_REPO_ROOT/sim5_amd/lib/ab_$name.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 tests/tools/bench_surface.py > $OUT/$name.log 2>&1
  python3 - "$OUT/$name" <<'PY' > $OUT/$name.txt
import csv, glob, sys, collections
acc = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "surface" not in n: continue
        k = ("fast " if "s5f::" in n else "strict ") + n.split("surface_")[1].split("(")[0]
        acc.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    v2 = sorted(v)
    print("%-40s launches %3d  mean %9.1f us  max %9.1f us  sum %10.1f us" % (k, len(v), sum(v) / len(v), v2[-1], sum(v)))
PY
  echo "== $name"; cat $OUT/$name.txt
done
