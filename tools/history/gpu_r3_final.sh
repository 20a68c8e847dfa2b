#!/bin/bash
# what the driver runs at round end: GPU suite, smoke, bench
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -4
( time timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err ) 2>&1 | grep real
tail -c 1500 gpurun_out/bench_final.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_final.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value', d['value'], 'single', d['single_stream']['registrations_per_s'], 'incl', d['value_including_set_pcd'].get('registrations_per_s'))
print('roofline', {k: r.get(k) for k in ('achieved','frac','avg_launch_us','launches','registrations_per_launch','frac_at_rocprofv3_duration','valu_issue_frac','traffic')}, r.get('rocprofv3'))
print('acvo', d['acvo'].get('registrations_per_s'), d['acvo'].get('single_stream'))
print('cfg3', d['config3_single_gpu'], 'cfg4', d['config4'].get('registrations_per_s'), 'sat', d['saturation'].get('registrations_per_s'))
print('frontend', {k:v for k,v in d['frontend'].items() if k in ('frames_per_s','stream','chain','matlab_prep')})
print('parity', d['parity_vs_oracle']['R_T_bit_identical'], d['parity_vs_oracle']['batched']['bit_identical_to_lone_cvo_hip_align'], d['parity_vs_oracle']['batched'].get('vs_oracle_bit_identical'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
