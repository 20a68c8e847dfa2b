#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel stats + PMC traffic.
# Outputs under gpurun_out/<tag>/ ; copy what should be judged into profiles/.
TAG=${1:-r01}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $ROOTDIR/bench.py --steps 5 --warmup 1 --no-cpu > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python $ROOTDIR/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python $ROOTDIR/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o sq -- python $ROOTDIR/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -20
find $OUT -name "*kernel_stats*" -exec head -12 {} \;
