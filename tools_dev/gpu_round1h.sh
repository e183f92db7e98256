#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -s > gpurun_out/test8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/test8.log
grep -E "queue|hires|passed|failed|FAILED|Error|error|rc=" gpurun_out/test8.log | tail -30
