#!/bin/bash
# ON THE GPU BOX: rays per second of a caller written against the SIM5 SCALAR API (tests/c/shim_probe.c: the call
# sequence of the reference's example 04 -- init_inf, midplane crossing, position_rad, gfactorK, disk_nt_flux per
# pixel -- through sim5_amd/host/sim5lib.c).  Two image sizes, so that process start-up (library load, GPU context:
# ~0.3 s) drops out of the marginal rate; with the row look-ahead (the default: one launch per image row), with the per-ray
# record alone (SIM5_SHIM_NO_LOOKAHEAD=1: one round trip per ray) and call by call (SIM5_SHIM_NO_CHAIN=1: five per ray).
cd $GRAFT_REPO_ROOT
gcc tests/c/shim_probe.c src/sim5lib.c -Isrc -o /tmp/probe -lm -O3 -w -fgnu89-inline || exit 1
export SIM5GPU_LIB=${SIM5GPU_LIB:-$GRAFT_REPO_ROOT/sim5_amd/lib/libsim5gpu.so}
run() { local t0=$(date +%s%N); "$@" /tmp/probe 0.998 70 $N > /tmp/probe.$N.$1$2.out; local t1=$(date +%s%N); echo "$(( t1 - t0 ))e-9"; }
N=64;  a1=$(run env); s1=$(run env SIM5_SHIM_NO_LOOKAHEAD=1); b1=$(run env SIM5_SHIM_NO_CHAIN=1)
N=256; a2=$(run env); s2=$(run env SIM5_SHIM_NO_LOOKAHEAD=1); b2=$(run env SIM5_SHIM_NO_CHAIN=1)
N=1024; a3=$(run env)
cmp /tmp/probe.256.env.out /tmp/probe.256.envSIM5_SHIM_NO_LOOKAHEAD=1.out && cmp /tmp/probe.256.env.out /tmp/probe.256.envSIM5_SHIM_NO_CHAIN=1.out && echo "outputs of the three modes: identical text (256^2)"
python3 - <<PY
n1, n2, n3 = 64 * 64, 256 * 256, 1024 * 1024
for name, t1, t2, m1, m2 in (("row look-ahead (default: one launch per image row)", $a1, $a2, n1, n2),
                     ("row look-ahead, 256^2 -> 1024^2", $a2, $a3, n2, n3),
                     ("one round trip per ray (SIM5_SHIM_NO_LOOKAHEAD=1)", $s1, $s2, n1, n2),
                     ("call by call (SIM5_SHIM_NO_CHAIN=1)", $b1, $b2, n1, n2)):
    per = (t2 - t1) / (m2 - m1)
    print("scalar SIM5 API over the GPU library, %s: %d rays in %.2f s, %d in %.2f s -> %.2f us per ray = %.3e rays/s (start-up %.2f s)" % (
        name, m1, t1, m2, t2, per * 1e6, 1.0 / per, t1 - m1 * per))
PY
# the loop by itself (no per-pixel printing, timed inside the program: tests/c/shim_probe.c quiet)
for N in 64 256 1024 2048; do /tmp/probe 0.998 70 $N quiet; done
SIM5_SHIM_NO_LOOKAHEAD=1 /tmp/probe 0.998 70 256 quiet
