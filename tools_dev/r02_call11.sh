#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools_dev/linear_probe.py > gpurun_out/r02_linear_probe5.log 2>&1; head -8 gpurun_out/r02_linear_probe5.log | tail -6 | cut -c1-120
( timeout 900 python -m pytest tests/test_gpu_linear.py tests/test_gpu_decoder.py tests/test_gpu_modules.py -q ) > gpurun_out/r02_tests11.log 2>&1; tail -3 gpurun_out/r02_tests11.log | cut -c1-300
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench11_hot.log 2>&1; python - <<PY
import json
for l in open('gpurun_out/r02_bench11_hot.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; m=d['mfma_kernels']; print('hot', d['value'], d['ms_per_step'], r['launch_ms'], m['occ_heads_launch_ms'], m['linear_ms_per_step'], m['conv3d_lifter']['launch_ms'], m['conv3d_2']['launch_ms'])
PY
