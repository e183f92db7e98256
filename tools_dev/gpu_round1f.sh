#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_linear.py tests/test_gpu_modules.py -m gpu -q -x > gpurun_out/test6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/test6.log
tail -3 gpurun_out/test6.log
timeout 300 python bench.py --scope hotpath --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench6_hot.log 2>&1
tail -1 gpurun_out/bench6_hot.log | cut -c1-200
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof6 -o r6 -- python $GRAFT_REPO_ROOT/bench.py --scope hotpath --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/prof6.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find /tmp/prof6 -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 40 > gpurun_out/prof6_hot_summary.txt 2>&1; head -16 gpurun_out/prof6_hot_summary.txt | cut -c1-160
python tools_dev/rocpd_summary.py $DB --dump linear_mfma 8 > gpurun_out/prof6_linear_dump.txt 2>&1; cat gpurun_out/prof6_linear_dump.txt | cut -c1-60
