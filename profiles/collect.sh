#!/bin/bash
# Runs ON THE GPU BOX (gpurun): rocprofv3 kernel statistics and PMC passes of the default bench
# command, one counter group per run (the TCC counters do not fit together; --pmc is never combined
# with other trace domains).  Writes raw CSVs under gpurun_out/prof_$TAG; profiles/summarize.py turns
# them into the committed summaries.
TAG=${1:-r01}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
# the headline launches only (the default bench line also times C2..C5 with the same kernel at other image sizes,
# which would mix into a per-kernel average); the default command itself is profiled as well, further down
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default_cmd -- python3 bench.py --no-cpu-baseline > $OUT/stats_default_cmd.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_sq1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $B > $OUT/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_sq2 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_WAIT_ANY -- $B > $OUT/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_fetch --pmc FETCH_SIZE -- $B > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_write --pmc WRITE_SIZE -- $B > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_busy --pmc GRBM_GUI_ACTIVE VALUBusy VALUUtilization -- $B > $OUT/pmc_busy.log 2>&1
grep '"metric"' $OUT/stats.log | tail -1 > $OUT/bench_under_rocprof.json
find $OUT -name "*.csv" | wc -l
