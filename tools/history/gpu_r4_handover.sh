#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_paths.py -x -q -k "one_launch_hand_over or batched_hand_over or cloud_in_device_memory or stream_of_varying" 2>&1 | tail -8
timeout 600 python bench.py --steps 40 --no-cpu --no-frontend --saturation-batch 0 2> gpurun_out/r4_ho.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('value', d['value'], 'incl', d.get('value_including_set_pcd'), d.get('value_including_set_pcd_over_value'))
print(json.dumps(d.get('hand_over'), indent=1))
print('single', d.get('single_stream'))
"
tail -5 gpurun_out/r4_ho.err
