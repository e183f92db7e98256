#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof16 -o r16 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof16.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find /tmp/prof16 -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB --dump conv1x1 40 | cut -c1-60 | tail -40 | awk '{s+=$1} END {print "conv1x1 per step us:", s}'
python tools_dev/rocpd_summary.py $DB --dump linear_bf16x3 32 | cut -c1-60 | tail -32 | awk '{s+=$1} END {print "linear per step us:", s}'
python tools_dev/rocpd_summary.py $DB --dump conv1x1 40 | cut -c1-64 | tail -40
