#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_training.py -m gpu -q -x -s > gpurun_out/test7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/test7.log
grep -E "loss|prev=|angle|rotated|passed|failed|FAILED|Error|error|rc=" gpurun_out/test7.log | tail -30
timeout 900 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench7_train.log 2>&1; echo "rc=$?" >> gpurun_out/bench7_train.log
tail -4 gpurun_out/bench7_train.log | cut -c1-600
