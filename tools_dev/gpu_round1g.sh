#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_training.py -m gpu -q -s > gpurun_out/test7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/test7.log
grep -E "loss|prev=|angle|rotated|passed|failed|FAILED|Error|error|rc=" gpurun_out/test7.log | tail -30
