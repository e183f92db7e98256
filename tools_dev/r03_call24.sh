#!/bin/bash
# round 3, call 24: pixel-pair layout of the fp16 value maps end to end: tests, probe, hot path
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_linear.py tests/test_gpu_modules.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -m gpu -q -x ) > gpurun_out/r03c24_tests.log 2>&1; tail -8 gpurun_out/r03c24_tests.log | cut -c1-200
timeout 200 python tools_dev/sca_probe.py 60 2>&1 | grep "^{" | cut -c1-210 > gpurun_out/r03c24_sca_probe.txt; cat gpurun_out/r03c24_sca_probe.txt
timeout 200 python tools_dev/vproj_probe.py 40 2>&1 | grep "^{" | cut -c1-200 > gpurun_out/r03c24_vproj_probe.txt; cat gpurun_out/r03c24_vproj_probe.txt
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c24_bench_hot.log 2>&1; grep '^{' gpurun_out/r03c24_bench_hot.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['value_no_instrumentation'], d['roofline']['launch_ms'])"
