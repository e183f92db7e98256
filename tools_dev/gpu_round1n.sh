#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_configs.py -k bench_multi -m gpu -q -s > gpurun_out/test14.log 2>&1; echo "pytest rc=$?" >> gpurun_out/test14.log
tail -15 gpurun_out/test14.log | cut -c1-300
