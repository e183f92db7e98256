#!/bin/bash
# round 3, call 17: Linear epilogue with pipelined residual requests + launch bounds (256, 3): tests, probe (events + kernel trace), hot path
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_linear.py tests/test_gpu_modules.py -m gpu -q -x ) > gpurun_out/r03c17_tests.log 2>&1; tail -2 gpurun_out/r03c17_tests.log | cut -c1-200
timeout 200 python tools_dev/linear_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-110 > gpurun_out/r03c17_linear_probe.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_l -o r -- python $GRAFT_REPO_ROOT/tools_dev/linear_probe.py > /dev/null 2>&1)
DB=$(find /tmp/prof_l -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 6 2>&1 | grep -i "linear_bf16x3\|total" | cut -c1-150 >> gpurun_out/r03c17_linear_probe.txt
cat gpurun_out/r03c17_linear_probe.txt
for i in 1 2; do
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c17_bench_hot_$i.log 2>&1; grep '^{' gpurun_out/r03c17_bench_hot_$i.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['value_no_instrumentation'], d['mfma_kernels']['linear_ms_per_step'])"
done
