#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof15 -o r15 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof15.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find /tmp/prof15 -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 40 > gpurun_out/prof15_e2e_summary.txt 2>&1; head -36 gpurun_out/prof15_e2e_summary.txt | cut -c1-150
python tools_dev/rocpd_summary.py $DB --dump conv1x1 40 | cut -c1-90
