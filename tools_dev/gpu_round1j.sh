#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dvr.py -m gpu -q -s > gpurun_out/test10.log 2>&1; echo "pytest rc=$?" >> gpurun_out/test10.log
grep -E "differ|RayIoU|passed|failed|FAILED|Error|error|rc=" gpurun_out/test10.log | tail -40
