#!/bin/bash
# round 3, call 19: SCA gather with the head-local softmax prologue: tests, probe, hot path
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -m gpu -q -x ) > gpurun_out/r03c19_tests.log 2>&1; tail -2 gpurun_out/r03c19_tests.log | cut -c1-200
timeout 200 python tools_dev/sca_probe.py 60 2>&1 | grep "^{" | cut -c1-200 > gpurun_out/r03c19_sca_probe.txt; cat gpurun_out/r03c19_sca_probe.txt
for i in 1 2; do
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c19_bench_hot_$i.log 2>&1; grep '^{' gpurun_out/r03c19_bench_hot_$i.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['value_no_instrumentation'], d['roofline']['launch_ms'])"
done
