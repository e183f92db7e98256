#!/bin/bash
# round 3, call 25: 32-k activation chunks (whole-line activation loads) in the tiled Linear: tests, ABAB probe, hot path
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_gpu_linear.py -m gpu -q -x ) > gpurun_out/r03c25_tests.log 2>&1; tail -2 gpurun_out/r03c25_tests.log | cut -c1-200
for d in 2 1 2 1; do
  echo "chunk k-steps $d" >> gpurun_out/r03c25_linear_ck.txt
  OCC_LINEAR_CK=$d timeout 200 python tools_dev/linear_probe.py 2>&1 | grep -v amdgpu.ids | cut -c10-90 >> gpurun_out/r03c25_linear_ck.txt
done
cat gpurun_out/r03c25_linear_ck.txt
for d in 2 1; do
OCC_LINEAR_CK=$d timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c25_bench_hot_ck$d.log 2>&1; grep '^{' gpurun_out/r03c25_bench_hot_ck$d.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ck$d', d['value'], d['ms_per_step'], d['value_no_instrumentation'], d['mfma_kernels']['linear_ms_per_step'])"
done
