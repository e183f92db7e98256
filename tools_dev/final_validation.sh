#!/bin/bash
# last call of a round: full GPU suite, smoke(), and the bench lines whose numbers the docs quote (no profiler passes)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-final}
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/${T}_tests.log 2>&1; grep "passed\|failed" gpurun_out/${T}_tests.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 900 python bench.py --steps 30 --warmup 5 ) > gpurun_out/${T}_bench_e2e.log 2>&1; grep '^{' gpurun_out/${T}_bench_e2e.log | cut -c1-170
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/${T}_bench_hot.log 2>&1; grep '^{' gpurun_out/${T}_bench_hot.log | cut -c1-150
timeout 600 python bench.py --config configs/occ_hires_400x400x32.py --scope hotpath --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/${T}_bench_hires_hot.log 2>&1; grep '^{' gpurun_out/${T}_bench_hires_hot.log | cut -c1-150
