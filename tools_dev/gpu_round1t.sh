#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python bench.py --mode train --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/bench18_train.log 2>&1; echo "rc=$?" >> gpurun_out/bench18_train.log
tail -2 gpurun_out/bench18_train.log | cut -c1-300
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof18 -o r18 -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 4 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof18.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find /tmp/prof18 -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 60 | grep -v naive_conv > gpurun_out/prof18_train_summary.txt 2>&1; head -34 gpurun_out/prof18_train_summary.txt | cut -c1-150
