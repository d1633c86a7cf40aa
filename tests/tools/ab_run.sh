#!/bin/bash
# ON THE GPU BOX: run a command against each prebuilt variant library, twice, alternating (box-to-box and run-to-run
# noise is several per cent: only numbers from one call are comparable).
#   gpurun -- bash tests/tools/ab_run.sh "python tests/tools/bench_torus.py" base fma
cd $GRAFT_REPO_ROOT
cmd=$1; shift
cp sim5_amd/lib/libsim5gpu.so /tmp/keep.so
for rep in 1 2; do
  for name in "$@"; do
    cp sim5_amd/lib/ab_$name.so sim5_amd/lib/libsim5gpu.so
    echo "== $name: $($cmd 2>&1 | tail -1)"
  done
done
cp /tmp/keep.so sim5_amd/lib/libsim5gpu.so
