#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof17 -o r17 -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof17.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find /tmp/prof17 -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 45 > gpurun_out/prof17_train_summary.txt 2>&1; head -48 gpurun_out/prof17_train_summary.txt | cut -c1-170
grep -E '^\{' gpurun_out/prof17.log | cut -c1-200
