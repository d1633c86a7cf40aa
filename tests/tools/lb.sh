#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "-DS5_LB_WAVES=1" "-DS5_LB_WAVES=2" "-DS5_LB_WAVES=3" "-DS5_LB_WAVES=4"; do
  rm -f sim5_amd/csrc/_build/k_disk_image_fast.o
  S5_FAST_EXTRA="$cfg" python sim5_amd/build.py > /dev/null 2>&1
  echo "=== fast with [$cfg]"
  python tests/tools/dbg3.py
done
rm -f sim5_amd/csrc/_build/*_fast.o; python sim5_amd/build.py > /dev/null 2>&1
