#!/bin/bash
# round 6: the changed paths under stress -- engines + the call's tail against registrations on their own (flaky hunt, cvo and acvo sizes),
# host threads, colliding runs
cd ${GRAFT_REPO_ROOT:-$(pwd)}
F='^RCCL\|^HIP\|^ROCm\|^/opt\|^Hostname\|^Librccl'
echo "== flaky hunt 10k x 64 x 6"; timeout 900 python tools/gpu_flaky_hunt.py 10000 64 6 2>&1 | grep -v "$F" | tail -3
echo "== flaky hunt 3k x 48 x 10"; timeout 900 python tools/gpu_flaky_hunt.py 3000 48 10 2>&1 | grep -v "$F" | tail -3
echo "== threads 8 x 30"; timeout 900 python tools/gpu_threads.py 8 30 2>&1 | grep -v "$F" | tail -3
echo "== colliding runs"; timeout 900 python tools/gpu_run_collide.py 10000 20 3 2>&1 | grep -v "$F" | tail -3
echo "== fresh contexts"; timeout 600 python tools/gpu_fresh_hunt.py 3000 32 30 2>&1 | grep -v "$F" | tail -2
