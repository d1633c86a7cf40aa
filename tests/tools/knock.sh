#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "" "-DS5_KO_FLUX" "-DS5_KO_FLUX -DS5_KO_G" "-DS5_KO_FLUX -DS5_KO_G -DS5_KO_RAD" "-DS5_KO_FLUX -DS5_KO_G -DS5_KO_RAD -DS5_KO_RF"; do
  rm -f sim5_amd/csrc/_build/*_fast.o
  S5_FAST_EXTRA="$cfg" python sim5_amd/build.py > /dev/null 2>&1
  echo "=== fast with [$cfg]"
  python tests/tools/image_check.py
done
rm -f sim5_amd/csrc/_build/*_fast.o; python sim5_amd/build.py > /dev/null 2>&1
