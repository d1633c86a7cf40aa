#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  rm -f sim5_amd/csrc/_build/k_torus_fast.o
  S5_TORUS_EXTRA="$cfg" python sim5_amd/build.py > /dev/null 2>&1
  echo "=== torus with [$cfg]"
  timeout 200 python tests/tools/bench_jobs.py 2>&1 | grep "C4" | cut -c1-110
done
rm -f sim5_amd/csrc/_build/k_torus_fast.o; python sim5_amd/build.py > /dev/null 2>&1
