#!/bin/bash
# round 3, call 13: activation prefetch depth of the tiled Linear (OCC_LINEAR_ADEPTH 2 vs 4)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_linear.py -m gpu -q -x ) > gpurun_out/r03c13_tests.log 2>&1; tail -2 gpurun_out/r03c13_tests.log | cut -c1-200
for d in 4 2 4 2; do
  echo "adepth $d" >> gpurun_out/r03c13_linear_adepth.txt
  OCC_LINEAR_ADEPTH=$d timeout 200 python tools_dev/linear_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03c13_linear_adepth.txt
done
cat gpurun_out/r03c13_linear_adepth.txt | cut -c1-150
for d in 4 2; do
OCC_LINEAR_ADEPTH=$d timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c13_bench_hot_ad$d.log 2>&1; grep '^{' gpurun_out/r03c13_bench_hot_ad$d.log | cut -c1-230
done
