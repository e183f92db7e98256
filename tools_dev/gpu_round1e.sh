#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench5_e2e.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench5_e2e.log
tail -2 gpurun_out/bench5_e2e.log | cut -c1-400
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --backbone-plan folded > gpurun_out/bench5_e2e_folded.log 2>&1
tail -1 gpurun_out/bench5_e2e_folded.log | cut -c1-300
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof5 -o r5 -- python $GRAFT_REPO_ROOT/bench.py --scope hotpath --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/prof5.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find /tmp/prof5 -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 40 > gpurun_out/prof5_hot_summary.txt 2>&1; head -30 gpurun_out/prof5_hot_summary.txt | cut -c1-180
python tools_dev/rocpd_summary.py $DB --dump linear_mfma 40 > gpurun_out/prof5_linear_dump.txt 2>&1; cat gpurun_out/prof5_linear_dump.txt
