#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_linear.py tests/test_gpu_modules.py tests/test_gpu_configs.py -m gpu -q -s > gpurun_out/test13.log 2>&1; echo "pytest rc=$?" >> gpurun_out/test13.log
grep -E "bf16x3|fused|golden|hires|bs=|tiny|prev|passed|failed|FAILED|Error|error|rc=" gpurun_out/test13.log | tail -60
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench13_hot.log 2>&1
tail -1 gpurun_out/bench13_hot.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('hot', d['value'], d['ms_per_step'], d['mfma_kernels'])"
OCC_LINEAR_PRECISION=f32 timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench13_hot_f32.log 2>&1
tail -1 gpurun_out/bench13_hot_f32.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('hot f32', d['value'], d['ms_per_step'], d['mfma_kernels'])"
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof13 -o r13 -- python $GRAFT_REPO_ROOT/bench.py --scope hotpath --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/prof13.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find /tmp/prof13 -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB --dump linear_bf16x3 8 | cut -c1-70
