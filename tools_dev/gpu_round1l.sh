#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/test12.log 2>&1; echo "pytest rc=$?" >> gpurun_out/test12.log
tail -4 gpurun_out/test12.log
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench12_e2e.log 2>&1
tail -1 gpurun_out/bench12_e2e.log | cut -c1-250
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench12_hot.log 2>&1
tail -1 gpurun_out/bench12_hot.log | cut -c1-200
