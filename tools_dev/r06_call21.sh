#!/bin/bash
# round 6 call 21: fused training nodes — backbone + training tests, kernel trace of the step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r06_c21
timeout 900 python -m pytest tests/test_gpu_backbone.py tests/test_gpu_training.py -m gpu -q 2>&1 | tail -6
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o r -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 6 --warmup 3 --passes 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${T}_trace.log 2>&1)
DB=$(find /tmp/prof_tr -name "*.db" | head -1)
python tools_dev/train_step_dump.py $DB > gpurun_out/${T}_train_step_dump.txt 2>&1
head -1 gpurun_out/${T}_train_step_dump.txt; tail -1 gpurun_out/${T}_train_step_dump.txt
