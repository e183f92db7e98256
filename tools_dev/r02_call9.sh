#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_linear.py -q -k "ffn" ) > gpurun_out/r02_tests9.log 2>&1; tail -2 gpurun_out/r02_tests9.log | cut -c1-200
timeout 300 python tools_dev/linear_probe.py > gpurun_out/r02_linear_probe4.log 2>&1; tail -3 gpurun_out/r02_linear_probe4.log
( timeout 900 python -m pytest tests/test_gpu_backbone.py -q -k "u8" ) > gpurun_out/r02_tests9b.log 2>&1; tail -5 gpurun_out/r02_tests9b.log | cut -c1-300
timeout 300 python bench.py --scope hotpath --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench9_hot.log 2>&1; tail -1 gpurun_out/r02_bench9_hot.log | cut -c1-200
