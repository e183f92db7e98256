#!/bin/bash
# round 2, call 20: SCA training path's query-side preparation as one kernel forward + one backward
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout -k 5 300 python -m pytest tests/test_gpu_training.py tests/test_gpu_modules.py -m gpu -q ) > gpurun_out/r02c20_tests.log 2>&1; tail -4 gpurun_out/r02c20_tests.log | cut -c1-200
for v in kernel torch; do
  OCC_CUDNN_BENCHMARK=0 OCC_SCA_TRAIN_PREP=$v timeout -k 5 150 python bench.py --mode train --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r02c20_train_$v.log 2>&1; grep '^{' gpurun_out/r02c20_train_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('train prep=$v', d['value'], d['ms_per_step'])"
done
