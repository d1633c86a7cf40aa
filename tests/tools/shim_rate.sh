#!/bin/bash
# ON THE GPU BOX: rays per second of a caller written against the SIM5 SCALAR API (tests/c/shim_probe.c: the call
# sequence of the reference's example 04 -- init_inf, midplane crossing, position_rad, gfactorK, disk_nt_flux per
# pixel -- through sim5_amd/host/sim5lib.c).  Two image sizes, so that process start-up (library load, GPU context:
# ~0.3 s) drops out of the marginal rate; with the shim's per-ray record (one round trip per ray) and without it
# (SIM5_SHIM_NO_CHAIN=1: five round trips per ray).
cd $GRAFT_REPO_ROOT
gcc tests/c/shim_probe.c src/sim5lib.c -Isrc -o /tmp/probe -lm -O3 -w -fgnu89-inline || exit 1
export SIM5GPU_LIB=$GRAFT_REPO_ROOT/sim5_amd/lib/libsim5gpu.so
run() { local t0=$(date +%s%N); "$@" /tmp/probe 0.998 70 $N > /tmp/probe.out; local t1=$(date +%s%N); echo "$(( t1 - t0 ))e-9"; }
N=64;  a1=$(run env); s1=$(run env SIM5_SHIM_STRICT=1); b1=$(run env SIM5_SHIM_NO_CHAIN=1)
N=256; a2=$(run env); s2=$(run env SIM5_SHIM_STRICT=1); b2=$(run env SIM5_SHIM_NO_CHAIN=1)
python3 - <<PY
n1, n2 = 64 * 64, 256 * 256
for name, t1, t2 in (("one round trip per ray (record, fast arithmetic: the default)", $a1, $a2),
                     ("one round trip per ray (record, strict arithmetic: SIM5_SHIM_STRICT=1)", $s1, $s2),
                     ("call by call (SIM5_SHIM_NO_CHAIN=1)", $b1, $b2)):
    per = (t2 - t1) / (n2 - n1)
    print("scalar SIM5 API over the GPU library, %s: %d rays in %.2f s, %d in %.2f s -> %.1f us per ray = %.3e rays/s (start-up %.2f s)" % (
        name, n1, t1, n2, t2, per * 1e6, 1.0 / per, t1 - n1 * per))
PY
