#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools_dev/linear_probe.py > gpurun_out/r02_linear_probe2.log 2>&1; cat gpurun_out/r02_linear_probe2.log | tail -7
( timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_linear.py tests/test_gpu_decoder.py -q ) > gpurun_out/r02_tests6.log 2>&1; tail -12 gpurun_out/r02_tests6.log | cut -c1-300
grep -E "fp16" gpurun_out/r02_tests6.log | head
