#!/bin/bash
# round 2, call 18: generic forward op and fused TSA gather on buffer loads (no dummy loads for out-of-map corners)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout -k 5 500 python -m pytest tests/test_gpu_msda.py tests/test_gpu_modules.py tests/test_gpu_backward.py tests/test_gpu_training.py tests/test_gpu_configs.py -m gpu -q ) > gpurun_out/r02c18_tests.log 2>&1; tail -4 gpurun_out/r02c18_tests.log | cut -c1-200
timeout -k 5 200 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02c18_hot.log 2>&1; grep '^{' gpurun_out/r02c18_hot.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("hot", d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"].get("tsa_launch_ms"))'
