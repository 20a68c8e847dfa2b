#!/bin/bash
# the round's closing check: the whole GPU suite and smoke() on the final binary, results into gpurun_out/
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > /tmp/suite.log 2>&1
grep -E "passed|failed|error" /tmp/suite.log | tail -3 > gpurun_out/r4b_final_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke >> gpurun_out/r4b_final_suite.txt
cat gpurun_out/r4b_final_suite.txt
