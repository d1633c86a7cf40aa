#!/bin/bash
# builds the fast variant with one feature off at a time and reports worst error vs the CPU oracle + speed
cd $GRAFT_REPO_ROOT
for cfg in "" "-DS5_F_SQRTDIV=0" "-DS5_F_RF7=0" "-DS5_F_AGMK=0" "-DS5_F_LIBM=0" "-ffp-contract=off"; do
  rm -f sim5_amd/csrc/_build/*_fast.o
  S5_FAST_EXTRA="$cfg" python sim5_amd/build.py > /dev/null 2>&1
  echo "=== fast with [$cfg]"
  python tests/tools/image_check.py
done
