#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python bench.py --steps 30 --warmup 5 > gpurun_out/bench9_e2e.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench9_e2e.log
tail -2 gpurun_out/bench9_e2e.log | cut -c1-1500
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench9_hot.log 2>&1
tail -1 gpurun_out/bench9_hot.log | cut -c1-200
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof9 -o r9 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof9.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find /tmp/prof9 -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 70 > gpurun_out/prof9_e2e_summary.txt 2>&1; head -50 gpurun_out/prof9_e2e_summary.txt | cut -c1-170
