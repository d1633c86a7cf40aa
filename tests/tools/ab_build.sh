#!/bin/bash
# Build a variant of the library HERE as sim5_amd/lib/ab_<name>.so (git-ignored, travels with gpurun), with its own
# objects (sim5_amd/csrc/_build/ab_<name>/): the in-tree libsim5gpu.so is not touched.
#   tests/tools/ab_build.sh <name> [ENV=VALUE ...]     e.g.  ab_build.sh fma S5_TORUS_FAST_EXTRA="-DPOOL_RUN=8"
cd /root/repo
name=$1; shift
env S5_VARIANT=$name "$@" python sim5_amd/build.py > /tmp/ab_build_$name.log 2>&1 || { grep -i "error" -A3 /tmp/ab_build_$name.log | head -30; exit 1; }
echo "built ab_$name.so"
