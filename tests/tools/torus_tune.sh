#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  rm -f sim5_amd/csrc/_build/k_torus_*.o
  S5_TORUS_EXTRA="$cfg" python sim5_amd/build.py > /dev/null 2>&1
  echo "=== torus with [$cfg]"
  python tests/tools/bench_torus.py
done
rm -f sim5_amd/csrc/_build/k_torus_*.o; python sim5_amd/build.py > /dev/null 2>&1
