#!/bin/bash
# What the waves of the engines' list kernels wait for ("or counters"): SQ busy / wait / per-unit active cycles and the
# texture path's busy and stall cycles, per kernel, on the roofline object's own command (one engine of 22 distinct 10k pairs).
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/r6_counters
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R="python $ROOTDIR/bench.py --roofline-only"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/a -o a -- $R > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAVES --output-format csv -d $OUT/b -o b -- $R > $OUT/b.log 2>&1
# (passes c, d -- TA_TA_BUSY, TA_*_STALLED_*, TCP_PENDING_STALL_CYCLES, TCP_TCC_READ_REQ ... -- abort inside rocprofv3 on this image with signal 6 and then
# sit until killed: one cost 25 GPU-minutes.  Not run.)
cd $ROOTDIR && python tools/r6_counters_sum.py $OUT > $OUT/summary.json; cat $OUT/summary.json | head -80
find $OUT -name "*.csv" -size +8M -delete
