#!/bin/bash
# round 2, call 17: MIOpen solver search (cudnn.benchmark) for the training step's stock convolutions
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 5 200 python bench.py --mode train --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r02c17_train_default.log 2>&1; grep '^{' gpurun_out/r02c17_train_default.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("default", d["value"], d["ms_per_step"])'
OCC_CUDNN_BENCHMARK=1 timeout -k 5 400 python bench.py --mode train --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r02c17_train_benchmark.log 2>&1; grep '^{' gpurun_out/r02c17_train_benchmark.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("cudnn.benchmark", d["value"], d["ms_per_step"])'; tail -3 gpurun_out/r02c17_train_benchmark.log | cut -c1-300
