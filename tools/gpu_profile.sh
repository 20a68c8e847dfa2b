#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel stats + PMC passes.
# Outputs under gpurun_out/<tag>/ ; tools/collect_profiles.py copies the summaries
# that should be judged into profiles/.
TAG=${1:-r02}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
# kernel durations are only meaningful with one registration in flight: --batch 1
B="python $ROOTDIR/bench.py --batch 1 --steps 5 --warmup 1 --no-cpu --no-side-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_batch -o stats -- python $ROOTDIR/bench.py --steps 5 --warmup 1 --no-cpu --no-side-legs > $OUT/stats_batch.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $B > $OUT/stats.log 2>&1
# PMC passes: counters only (gpurun refuses --pmc together with the trace domains)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $B > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $B > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o sq -- $B > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_lds -o lds -- $B > $OUT/pmc_lds.log 2>&1
# HBM traffic of the batched run's kernels (default batch: the launches of the timed region)
BB="python $ROOTDIR/bench.py --steps 3 --warmup 1 --no-cpu --no-side-legs"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_b -o fetch -- $BB > $OUT/pmc_fetch_b.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_b -o write -- $BB > $OUT/pmc_write_b.log 2>&1
# the front end: kernel trace of 100 synthetic VGA frames
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fe -o stats -- python $ROOTDIR/tools/gpu_frontend.py 100 1.0 > $OUT/stats_fe.log 2>&1
tail -2 $OUT/stats_fe.log
find $OUT -name "*.csv" | head -30
head -12 $OUT/stats/*kernel_stats.csv
