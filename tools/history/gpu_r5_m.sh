#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
S=$(date +%s.%N)
python bench.py > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err
E=$(date +%s.%N); echo "bench wall $(echo "$E - $S" | bc) s"
tail -2 gpurun_out/bench_now.err | cut -c1-200
