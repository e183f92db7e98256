#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_backbone.py -q ) > gpurun_out/r02_tests10.log 2>&1; tail -4 gpurun_out/r02_tests10.log | cut -c1-300
for sw in raster polar; do
OCC_BEV_SWEEP=$sw timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench10_e2e_$sw.log 2>&1; python - <<PY
import json
for l in open('gpurun_out/r02_bench10_e2e_$sw.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$sw', d['value'], d['ms_per_step'], r['launch_ms'], d['mfma_kernels']['occ_heads_launch_ms'], d['mfma_kernels']['linear_ms_per_step'])
PY
done
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --input u8-h2d > gpurun_out/r02_bench10_e2e_u8.log 2>&1; tail -1 gpurun_out/r02_bench10_e2e_u8.log | cut -c1-200
timeout 600 python bench.py --mode train --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r02_bench10_train.log 2>&1; tail -1 gpurun_out/r02_bench10_train.log | cut -c1-250
