#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "-DS5_TILE_W=16" "-DS5_TILE_W=32" "-DS5_TILE_W=64" "-DS5_TILE_W=8"; do
  rm -f sim5_amd/csrc/_build/k_disk_image_fast.o
  S5_FAST_EXTRA="$cfg" python sim5_amd/build.py > /dev/null 2>&1
  echo "=== fast with [$cfg]"
  python tests/tools/image_check.py
done
rm -f sim5_amd/csrc/_build/*_fast.o; python sim5_amd/build.py > /dev/null 2>&1
