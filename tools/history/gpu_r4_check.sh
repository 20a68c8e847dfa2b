#!/bin/bash
# round-4 working check: GPU suite, then the default bench line
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 900 python bench.py > gpurun_out/r4_bench_base.json 2> gpurun_out/r4_bench_base.err
tail -c 600 gpurun_out/r4_bench_base.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4_bench_base.json') if l.startswith('{')][-1])
print('value',d['value'],'ms/step',d['ms_per_step'])
for k in ('value_including_set_pcd','saturation','single_stream','config4','acvo'):
    print(k, json.dumps(d.get(k))[:300])
PY
