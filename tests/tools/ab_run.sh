#!/bin/bash
# ON THE GPU BOX: run a command against each prebuilt variant library, twice, alternating (box-to-box and run-to-run
# noise is several per cent: only numbers from one call are comparable).  The variant is selected through SIM5GPU_LIB
# (sim5_amd/capi.py and the C host shim read it): the in-tree libsim5gpu.so is never touched.
#   gpurun -- bash tests/tools/ab_run.sh "python tests/tools/bench_torus.py" base fma
cd $GRAFT_REPO_ROOT
cmd=$1; shift
for rep in 1 2; do
  for name in "$@"; do
    echo "== $name: $(SIM5GPU_LIB=$GRAFT_REPO_ROOT/sim5_amd/lib/ab_$name.so $cmd 2>&1 | tail -${AB_TAIL:-1})"
  done
done
