cd $GRAFT_REPO_ROOT
echo "#### head 10k"; TAG=ts_h10 N=10000 bash tools/gpu_trace_single.sh
echo "#### classic 10k"; CVO_HIP_NO_HEAD=1 TAG=ts_c10 N=10000 bash tools/gpu_trace_single.sh
echo "#### head 3k"; TAG=ts_h3 N=3000 bash tools/gpu_trace_single.sh
echo "#### acvo head 256 10k"; CVO_HIP_HEAD_ACVO=1 CVO_HIP_PROC_BLOCKS=256 MODE=acvo TAG=ts_ha10 N=10000 bash tools/gpu_trace_single.sh
echo "#### acvo classic 256 10k"; CVO_HIP_PROC_BLOCKS=256 MODE=acvo TAG=ts_ca10 N=10000 bash tools/gpu_trace_single.sh
export CVO_HIP_GRAPH=1
echo "#### acvo classic 256/512"; for pb in 256 512; do for n in 10000 3000; do CVO_HIP_PROC_BLOCKS=$pb python tools/gpu_single.py $n 30 acvo 2>&1 | grep "^single"; done; done
echo "#### acvo head 128/512"; for pb in 128 512; do for n in 10000 3000; do CVO_HIP_HEAD_ACVO=1 CVO_HIP_PROC_BLOCKS=$pb python tools/gpu_single.py $n 30 acvo 2>&1 | grep "^single"; done; done
