#!/bin/bash
# A/B of the torus march kernel: rebuild the fast variant of k_torus.hip with extra flags HERE (no GPU needed),
# then run the timing + step-count check on the GPU box.   usage: tests/tools/torus_try.sh "<extra hipcc flags>"
cd /root/repo
rm -f sim5_amd/csrc/_build/k_torus_fast.o
S5_TORUS_FAST_EXTRA="$1" python sim5_amd/build.py > /tmp/torus_try_build.log 2>&1 || { tail -20 /tmp/torus_try_build.log; exit 1; }
grep -A12 "torus_pool_kernel" /tmp/torus_try_build.log | head -0
/usr/local/graft/bin/gpurun --timeout 600 -- 'python tests/tools/bench_torus.py && python tests/tools/diag_c4.py' 2>&1 | grep -v "^\[gpurun\] sending\|merged"
