#!/bin/bash
# GPU suite + the driver's bench command
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
( time timeout 900 python bench.py > gpurun_out/bench_r3.json 2> gpurun_out/bench_r3.err ) 2>&1 | grep real
tail -c 3000 gpurun_out/bench_r3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r3.json').read().strip().splitlines()[-1])
def show(k):
    v=d.get(k); print(k, '=>', json.dumps(v)[:1500])
for k in ('value','ms_per_step','single_stream','parity_vs_oracle','value_including_set_pcd','acvo','config3_single_gpu','config4','saturation','identical_pairs'):
    show(k)
print('roofline =>', json.dumps({k:v for k,v in d['roofline'].items() if k not in ('reading','kernel','bytes_definition')})[:3000])
print('frontend =>', json.dumps(d.get('frontend'))[:1500])
print('cpu =>', json.dumps(d.get('cpu_baseline'))[:800])
PY
