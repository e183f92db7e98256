#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_linear.py -q -k "ffn" -s ) > gpurun_out/r02_tests8.log 2>&1; grep -E "ffn_fused|passed|failed|Error" gpurun_out/r02_tests8.log | cut -c1-200
timeout 300 python tools_dev/linear_probe.py > gpurun_out/r02_linear_probe3.log 2>&1; tail -4 gpurun_out/r02_linear_probe3.log
( timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py -q -x ) > gpurun_out/r02_tests8b.log 2>&1; tail -3 gpurun_out/r02_tests8b.log
timeout 300 python bench.py --scope hotpath --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench8_hot.log 2>&1; tail -1 gpurun_out/r02_bench8_hot.log | cut -c1-200
