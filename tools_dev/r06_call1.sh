#!/bin/bash
# round 6 call 1: mask-free set-up shipped -> full gpu suite (with the standing hazard tests) + hot-path / e2e bench lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r06_c1
( time timeout 600 python -m pytest tests/test_gpu_hazard_repro.py -m gpu -q -s ) > gpurun_out/${T}_hazard.log 2>&1; grep -E "HAZARD|library TSA|training backward|passed|failed|skipped" gpurun_out/${T}_hazard.log | cut -c1-250
( time timeout 1700 python -m pytest tests -m gpu -q --deselect tests/test_gpu_hazard_repro.py ) > gpurun_out/${T}_tests.log 2>&1; tail -4 gpurun_out/${T}_tests.log
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_hot.log 2>&1; grep '^{' gpurun_out/${T}_bench_hot.log | cut -c1-300
