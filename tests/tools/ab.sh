#!/bin/bash
# usage: ab.sh "<flags A>" "<flags B>" ...   (fast variant of the image kernel)
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  rm -f sim5_amd/csrc/_build/*_fast.o
  S5_FAST_EXTRA="$cfg" python sim5_amd/build.py > /dev/null 2>&1
  echo "=== fast with [$cfg]"
  python tests/tools/image_check.py | grep -o "flips [0-9]*\|r max [0-9.e+-]*\|flux max [0-9.e+-]*\|[0-9.]* ms.*" | tr '\n' ' '; echo
done
rm -f sim5_amd/csrc/_build/*_fast.o; python sim5_amd/build.py > /dev/null 2>&1
