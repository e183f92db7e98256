#!/bin/bash
# round 3, call 3: Linear / FFN kernels after the register-pressure rework, history-chain debug
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_linear.py -m gpu -q -x ) > gpurun_out/r03c3_tests_linear.log 2>&1; tail -4 gpurun_out/r03c3_tests_linear.log | cut -c1-200
timeout 300 python tools_dev/linear_probe.py > gpurun_out/r03c3_linear_probe.log 2>&1; grep -v amdgpu.ids gpurun_out/r03c3_linear_probe.log | cut -c1-250
timeout 600 python tools_dev/debug_history.py > gpurun_out/r03c3_debug_history.log 2>&1; grep -v "amdgpu.ids\|Warn\|warn" gpurun_out/r03c3_debug_history.log | tail -40 | cut -c1-200
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c3_bench_hot.log 2>&1; grep '^{' gpurun_out/r03c3_bench_hot.log | cut -c1-150
OCC_LINEAR_KERNEL=x3 timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c3_bench_hot_x3.log 2>&1; grep '^{' gpurun_out/r03c3_bench_hot_x3.log | cut -c1-150
