#!/bin/bash
# round 3, call 9: activation-resident value projection: tests, A/B probe
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_linear.py tests/test_gpu_modules.py -m gpu -q -x -k "value_proj or direct_value or fp16" -s ) > gpurun_out/r03c9_tests.log 2>&1; tail -5 gpurun_out/r03c9_tests.log | cut -c1-200; grep -h "resident value" gpurun_out/r03c9_tests.log
timeout 200 python tools_dev/vproj_probe.py > gpurun_out/r03c9_vproj_probe.txt 2>&1
OCC_VPROJ_RESIDENT=0 timeout 200 python tools_dev/vproj_probe.py >> gpurun_out/r03c9_vproj_probe.txt 2>&1
grep '^{' gpurun_out/r03c9_vproj_probe.txt
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c9_bench_hot.log 2>&1; grep '^{' gpurun_out/r03c9_bench_hot.log | cut -c1-260
