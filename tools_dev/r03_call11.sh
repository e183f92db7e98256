#!/bin/bash
# round 3, call 11: activation-resident Linear: tests, A/B probe, hot path
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_linear.py -m gpu -q -x -s ) > gpurun_out/r03c11_tests.log 2>&1; tail -5 gpurun_out/r03c11_tests.log | cut -c1-200; grep -h "^xr_\|xr_.*max" gpurun_out/r03c11_tests.log | cut -c1-120
timeout 200 python tools_dev/linear_probe.py > gpurun_out/r03c11_linear_probe.txt 2>&1
OCC_LINEAR_RESIDENT=0 timeout 200 python tools_dev/linear_probe.py >> gpurun_out/r03c11_linear_probe.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03c11_linear_probe.txt
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c11_bench_hot.log 2>&1; grep '^{' gpurun_out/r03c11_bench_hot.log | cut -c1-260
OCC_LINEAR_RESIDENT=0 timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c11_bench_hot_tiled.log 2>&1; grep '^{' gpurun_out/r03c11_bench_hot_tiled.log | cut -c1-260
