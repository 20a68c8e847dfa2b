#!/bin/bash
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/phases
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
DISTINCT=1 CVO_HIP_GRAPH=1 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $ROOTDIR/tools/gpu_batch.py 10000 3 64 > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt
python $ROOTDIR/tools/trace_engine_phases.py $(ls $OUT/*kernel_trace.csv | head -1) 3
