#!/bin/bash
# round 6 call 20: training step kernel trace with the fused backbone nodes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r06_c20
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o r -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 6 --warmup 3 --passes 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${T}_trace.log 2>&1)
grep '^{' gpurun_out/${T}_trace.log | cut -c1-200
DB=$(find /tmp/prof_tr -name "*.db" | head -1)
python tools_dev/train_step_dump.py $DB > gpurun_out/${T}_train_step_dump.txt 2>&1
head -3 gpurun_out/${T}_train_step_dump.txt; tail -2 gpurun_out/${T}_train_step_dump.txt
