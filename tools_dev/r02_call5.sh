#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_linear.py tests/test_gpu_modules.py -x -q ) > gpurun_out/r02_tests5.log 2>&1; tail -3 gpurun_out/r02_tests5.log
grep -E "fp16 values" gpurun_out/r02_tests5.log
timeout 300 python tools_dev/linear_probe.py > gpurun_out/r02_linear_probe.log 2>&1; cat gpurun_out/r02_linear_probe.log | tail -8
timeout 300 python tools_dev/sca_probe.py 40 > gpurun_out/r02_sca_probe4.log 2>&1; grep '^[0-9]' gpurun_out/r02_sca_probe4.log | cut -c1-200
OCC_SCA_VALUES=f16 OCC_BEV_SWEEP=polar timeout 300 python bench.py --scope hotpath --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench5_hot_f16.log 2>&1; tail -1 gpurun_out/r02_bench5_hot_f16.log | cut -c1-1800
timeout 300 python bench.py --scope hotpath --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench5_hot.log 2>&1; tail -1 gpurun_out/r02_bench5_hot.log | cut -c1-1800
OCC_LINEAR_KERNEL=x3 timeout 300 python bench.py --scope hotpath --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench5_hot_x3.log 2>&1; tail -1 gpurun_out/r02_bench5_hot_x3.log | cut -c1-400
