#!/bin/bash
# usage: surf_ab.sh "<flags A>" "<flags B>" ...   (fast variant of the surface-search kernel)
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  rm -f sim5_amd/csrc/_build/k_surface_fast.o
  S5_FAST_EXTRA="$cfg" python sim5_amd/build.py > /dev/null 2>&1
  echo "=== surface kernel with [$cfg]"
  timeout 200 python tests/tools/bench_jobs.py 2>&1 | grep "surface search 1024^2 fast"
done
rm -f sim5_amd/csrc/_build/k_surface_fast.o; python sim5_amd/build.py > /dev/null 2>&1
