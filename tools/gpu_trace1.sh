#!/bin/bash
# kernel trace of one-registration-at-a-time runs: per-kernel durations and the gaps between launches
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/${TAG:-trace1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $ROOTDIR/tools/gpu_batch.py ${N:-10000} ${REPS:-10} ${B:-1} $MODE > $OUT/log.txt 2>&1
grep "^B" $OUT/log.txt
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python $ROOTDIR/tools/trace_timeline.py $f ${SKIP_MS:-0} | cut -c1-250 | head -${LINES_OUT:-30}
rm -f $f
