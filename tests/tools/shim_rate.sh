#!/bin/bash
# ON THE GPU BOX: rays per second of a caller written against the SIM5 SCALAR API (tests/c/shim_probe.c: the call
# sequence of the reference's example 04 -- init_inf, midplane crossing, position_rad, gfactorK, disk_nt_flux per
# pixel -- every call one n = 1 launch through sim5_amd/host/sim5lib.c), next to the reference CPU library and the
# whole-image entry point on the same image.
cd $GRAFT_REPO_ROOT
N=${1:-160}
gcc tests/c/shim_probe.c src/sim5lib.c -Isrc -o /tmp/probe -lm -O3 -w -fgnu89-inline || exit 1
export SIM5GPU_LIB=$GRAFT_REPO_ROOT/sim5_amd/lib/libsim5gpu.so
t0=$(date +%s.%N); /tmp/probe 0.998 70 $N > /tmp/probe.out; t1=$(date +%s.%N)
python3 - <<PY
n=$N*$N; dt=$t1-$t0
print("scalar SIM5 API over the GPU library: %d rays in %.2f s = %.3e rays/s (%.1f us per ray, ~5 calls per ray)" % (n, dt, n/dt, dt/n*1e6))
PY
