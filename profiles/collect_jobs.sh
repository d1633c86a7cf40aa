#!/bin/bash
# Runs ON THE GPU BOX (gpurun): rocprofv3 kernel statistics and PMC passes of the whole-job kernels other than the
# headline image kernel (polarized image, torus start + march, spectrum, surface search; tests/tools/bench_jobs.py
# launches each a few times).  One counter group per run, --pmc never combined with other trace domains, the
# program directly after `--`.  Raw CSVs go to gpurun_out/prof_jobs_$TAG; profiles/summarize_jobs.py turns them
# into the committed profiles/${TAG}_jobs_summary.json.
TAG=${1:-r02}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_jobs_$TAG
rm -rf $OUT; mkdir -p $OUT
B="python3 tests/tools/bench_jobs.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_sq1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS -- $B > $OUT/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_sq2 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY -- $B > $OUT/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_busy --pmc GRBM_GUI_ACTIVE VALUBusy VALUUtilization -- $B > $OUT/pmc_busy.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_fetch --pmc FETCH_SIZE -- $B > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_write --pmc WRITE_SIZE -- $B > $OUT/pmc_write.log 2>&1
cp $OUT/stats.log $OUT/bench_jobs_under_rocprof.txt
python3 profiles/summarize_jobs.py $TAG
