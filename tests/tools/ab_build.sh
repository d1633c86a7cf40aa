#!/bin/bash
# Build a variant of the library HERE and keep it as sim5_amd/lib/ab_<name>.so (git-ignored, travels with gpurun).
#   tests/tools/ab_build.sh <name> [ENV=VALUE ...]     e.g.  ab_build.sh fma S5_TORUS_FAST_EXTRA="-DPOOL_RUN=8"
cd /root/repo
name=$1; shift
env "$@" python sim5_amd/build.py > /tmp/ab_build_$name.log 2>&1 || { grep -i "error" -A3 /tmp/ab_build_$name.log | head -30; exit 1; }
cp sim5_amd/lib/libsim5gpu.so sim5_amd/lib/ab_$name.so
echo "built ab_$name.so"
