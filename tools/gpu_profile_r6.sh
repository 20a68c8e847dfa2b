#!/bin/bash
# Round-6 profile run (GPU box, via gpurun): round 3's commands + the compact bench line beside the full object, the acvo
# single-stream counter pass (kt_run_acvo) and the run clocks of a -DCVO_RUN_CLOCKS build.  Outputs under gpurun_out/<tag>/;
# tools/collect_profiles_r3.py copies what is judged into profiles/.
TAG=${1:-r06}
ROOTDIR=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOTDIR
python bench.py --steps 20 --warmup 3 > $OUT/bench_line.json 2> $OUT/bench.err
cp bench_detail.json $OUT/bench.json      # (the full object; bench_line.json is the line the driver reads)
tail -c 900 $OUT/bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
# ---- the roofline object's own command: one engine of 22 pairs, flow-pass dispatches bracketed by HIP events
R="python $ROOTDIR/bench.py --roofline-only"
$R > $OUT/roofline_plain.json 2> $OUT/roofline_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_roofline -o stats -- $R > $OUT/stats_roofline.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_r -o fetch -- $R > $OUT/pmc_fetch_r.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_r -o write -- $R > $OUT/pmc_write_r.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq_r -o sq -- $R > $OUT/pmc_sq_r.log 2>&1
# ---- the timed region (three engines, 64 distinct pairs): kernel time shares
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_batch -o stats -- python $ROOTDIR/bench.py --steps 5 --warmup 1 --no-cpu --no-side-legs > $OUT/stats_batch.log 2>&1
# ---- VALU issue of the batched run at saturation (256 distinct pairs per call), and its wall time without counters
export CVO_HIP_GRAPH=1
DISTINCT=1 python $ROOTDIR/tools/gpu_batch.py 10000 4 256 > $OUT/valu_wall.log 2>&1
DISTINCT=1 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_valu -o p -- python $ROOTDIR/tools/gpu_batch.py 10000 1 256 > $OUT/pmc_valu.log 2>&1
unset CVO_HIP_GRAPH
# ---- one registration at a time (head mode): kernel trace + the wait counters of the post-step part
S="python $ROOTDIR/bench.py --batch 1 --steps 5 --warmup 1 --no-cpu --no-side-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $S > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o sq -- $S > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $S > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $S > $OUT/pmc_write.log 2>&1
# ---- acvo one at a time (kt_run_acvo): kernel trace + SQ counters
A="python $ROOTDIR/bench.py --mode acvo --batch 1 --steps 5 --warmup 1 --no-cpu --no-side-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_acvo -o stats -- $A > $OUT/stats_acvo.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq_acvo -o sq -- $A > $OUT/pmc_sq_acvo.log 2>&1
# ---- where a run's iteration goes (ticks of the first solver block by phase; a -DCVO_RUN_CLOCKS build, tools/build_variant.sh clk)
if [ -f $ROOTDIR/cvo-rgbd_amd/csrc/libcvo_hip_clk.so ]; then
  (cd $ROOTDIR && CVO_LIB=libcvo_hip_clk.so python tools/gpu_run_clocks.py 3000 10000 > $OUT/run_clocks_cvo.txt 2>&1; ACVO=1 CVO_LIB=libcvo_hip_clk.so python tools/gpu_run_clocks.py 3000 10000 > $OUT/run_clocks_acvo.txt 2>&1)
fi
# ---- the front end
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fe -o stats -- python $ROOTDIR/tools/gpu_frontend.py 100 1.0 > $OUT/stats_fe.log 2>&1
cd $ROOTDIR && python tools/collect_profiles_r3.py $TAG --keep-in-out
find $OUT -name "*.csv" -size +20M -delete   # (gpurun_out/ comes back at <= 64 MiB)
du -sh $OUT
