#!/bin/bash
# round 4, call 11: new tests (fp16 saturation, content fingerprint, deterministic backward) + the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_linear.py tests/test_gpu_training.py tests/test_gpu_backward.py tests/test_gpu_modules.py -m gpu -q -k "saturat or fingerprint or deterministic or history_bev_inside or gather_stats or rotat" > gpurun_out/r04_c11_tests.log 2>&1; tail -12 gpurun_out/r04_c11_tests.log
( time timeout 900 python bench.py ) > gpurun_out/r04_c11_bench_default.log 2>&1; grep '^{' gpurun_out/r04_c11_bench_default.log | cut -c1-300; grep real gpurun_out/r04_c11_bench_default.log
