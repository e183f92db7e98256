#!/bin/bash
# round 2, last call: the committed default configuration end to end — full GPU test suite, smoke(), one e2e and one
# training bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout -k 5 400 python -m pytest tests -m gpu -q ) > gpurun_out/r02f3_tests.log 2>&1; tail -3 gpurun_out/r02f3_tests.log | cut -c1-200
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02f3_smoke.log 2>&1; tail -2 gpurun_out/r02f3_smoke.log | cut -c1-200
timeout -k 5 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02f3_bench_e2e.log 2>&1; grep '^{' gpurun_out/r02f3_bench_e2e.log | cut -c1-180
