#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_backbone.py tests/test_gpu_linear.py tests/test_gpu_modules.py -m gpu -q > gpurun_out/test16.log 2>&1; echo "pytest rc=$?" >> gpurun_out/test16.log
tail -3 gpurun_out/test16.log
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench16_e2e.log 2>&1
tail -1 gpurun_out/bench16_e2e.log | cut -c1-250
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench16_hot.log 2>&1
tail -1 gpurun_out/bench16_hot.log | cut -c1-200
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof16 -o r16 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof16.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find /tmp/prof16 -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB --dump conv1x1 40 | cut -c1-60 | tail -40 | awk '{s+=$1} END {print "conv1x1 per step us:", s}'
python tools_dev/rocpd_summary.py $DB --dump linear_bf16x3 32 | cut -c1-60 | tail -32 | awk '{s+=$1} END {print "linear per step us:", s}'
