#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1200 python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err
tail -c 400 gpurun_out/r4_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4_bench.json') if l.startswith('{')][-1])
print('value',d['value'],'ms/step',d['ms_per_step'], 'incl', d.get('value_including_set_pcd'), d.get('value_including_set_pcd_over_value'))
r=d['roofline']
print('roofline frac', r.get('frac'), 'bound', r.get('bound'), 'avg_launch_us', r.get('avg_launch_us'))
print('by_phase', json.dumps(r.get('by_phase'), indent=1)[:3000])
print('config3', json.dumps(d.get('config3_single_gpu'), indent=1)[:2500])
print('shard', json.dumps(d.get('config3_shard_of_8'), indent=1))
for k in ('saturation','single_stream','config4','hand_over','frontend'):
    print(k, json.dumps(d.get(k))[:600])
PY
