#!/bin/bash
# round 4, call 1: chain kernel parity + probe + module / full-size parity on the chain path + bench hot/e2e
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_linear.py -m gpu -q -k "chain" -x > gpurun_out/r04_c1_chain_tests.log 2>&1; tail -15 gpurun_out/r04_c1_chain_tests.log
timeout 300 python tools_dev/chain_probe.py > gpurun_out/r04_c1_chain_probe.txt 2>&1; cat gpurun_out/r04_c1_chain_probe.txt
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -q > gpurun_out/r04_c1_module_tests.log 2>&1; tail -8 gpurun_out/r04_c1_module_tests.log
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r04_c1_bench_hot.log 2>&1; grep '^{' gpurun_out/r04_c1_bench_hot.log | cut -c1-400
OCC_LINEAR_CHAIN=0 timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r04_c1_bench_hot_nochain.log 2>&1; grep '^{' gpurun_out/r04_c1_bench_hot_nochain.log | cut -c1-200
