#!/bin/bash
# Full gpu test suite on the GPU box: /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools_dev/gpu_tests.sh [pytest args]'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q "$@" > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log
grep -E "passed|failed|FAILED|Error|rc=" gpurun_out/gpu_tests.log | tail -15
