#!/bin/bash
# round 4, call 10: full GPU suite on the chain path + hot-path / e2e bench lines (chain vs OCC_LINEAR_CHAIN=0)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1700 python -m pytest tests -m gpu -q ) > gpurun_out/r04_c10_tests.log 2>&1; tail -6 gpurun_out/r04_c10_tests.log
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r04_c10_bench_hot.log 2>&1; grep '^{' gpurun_out/r04_c10_bench_hot.log | cut -c1-330
OCC_LINEAR_CHAIN=0 timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r04_c10_bench_hot_nochain.log 2>&1; grep '^{' gpurun_out/r04_c10_bench_hot_nochain.log | cut -c1-330
timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r04_c10_bench_e2e.log 2>&1; grep '^{' gpurun_out/r04_c10_bench_e2e.log | cut -c1-330
