#!/bin/bash
# one training step cut out of a rocprofv3 kernel trace (tools_dev/train_step_dump.py) -> gpurun_out/<tag>_train_step_dump.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-train}
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o r -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 6 --warmup 3 --passes 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${T}_trace.log 2>&1)
DB=$(find /tmp/prof_tr -name "*.db" | head -1)
python tools_dev/train_step_dump.py $DB > gpurun_out/${T}_train_step_dump.txt 2>&1
head -1 gpurun_out/${T}_train_step_dump.txt; tail -1 gpurun_out/${T}_train_step_dump.txt
