#!/bin/bash
# kernel trace of one-registration-at-a-time runs (cvo_hip_align): per-kernel durations and the gaps
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/${TAG:-trace_single}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CVO_HIP_GRAPH=1
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $ROOTDIR/tools/gpu_single.py ${N:-10000} ${REPS:-10} ${MODE:-cvo} > $OUT/log.txt 2>&1
grep "^single" $OUT/log.txt
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python $ROOTDIR/tools/trace_timeline.py $f ${SKIP_MS:-0} | cut -c1-200 | head -${LINES_OUT:-24}
rm -f $f
