#!/bin/bash
# round 2, call 16: heads kernel with next-tile prefetch, TSA query Linears merged in training
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout -k 5 400 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_training.py tests/test_gpu_modules.py tests/test_gpu_configs.py -m gpu -q ) > gpurun_out/r02c16_tests.log 2>&1; tail -4 gpurun_out/r02c16_tests.log | cut -c1-200
timeout -k 5 200 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02c16_hot.log 2>&1; grep '^{' gpurun_out/r02c16_hot.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("hot", d["value"], d["ms_per_step"], d.get("kernel_ms_per_step"))'
timeout -k 5 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02c16_e2e.log 2>&1; grep '^{' gpurun_out/r02c16_e2e.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("e2e", d["value"], d["ms_per_step"]); print({k:v for k,v in d.items() if "kernel" in k or "mfma" in k})' | cut -c1-900
timeout -k 5 200 python bench.py --mode train --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r02c16_train.log 2>&1; grep '^{' gpurun_out/r02c16_train.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("train", d["value"], d["ms_per_step"])'
