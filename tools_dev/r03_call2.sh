#!/bin/bash
# round 3, call 2: full GPU suite on the new defaults (fp16 SCA value rows, weight-stationary Linear, one-launch FFN),
# Linear / SCA probes, bench lines.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -s ) > gpurun_out/r03c2_tests.log 2>&1; tail -15 gpurun_out/r03c2_tests.log | cut -c1-200
grep -h "max|hip - oracle|\|max|hip - reference|\|fp16 values" gpurun_out/r03c2_tests.log | grep -i "base \|hires\|golden base_full\|reference golden\|history\|fp16 values" | cut -c1-200 | head -40
timeout 300 python tools_dev/linear_probe.py > gpurun_out/r03c2_linear_probe.log 2>&1; grep -v amdgpu.ids gpurun_out/r03c2_linear_probe.log | cut -c1-250
timeout 300 python tools_dev/sca_probe.py 40 > gpurun_out/r03c2_sca_probe.log 2>&1; grep '^{' gpurun_out/r03c2_sca_probe.log | cut -c1-200
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c2_bench_hot.log 2>&1; grep '^{' gpurun_out/r03c2_bench_hot.log | cut -c1-150
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c2_bench_e2e.log 2>&1; grep '^{' gpurun_out/r03c2_bench_e2e.log | cut -c1-150
OCC_LINEAR_KERNEL=x3 OCC_FFN=two OCC_SCA_VALUES=f32 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c2_bench_e2e_r02kernels.log 2>&1; grep '^{' gpurun_out/r03c2_bench_e2e_r02kernels.log | cut -c1-150
