#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
tools/gpu_timeline.sh 3000 | head -40
tools/gpu_timeline.sh 6000 | head -30
