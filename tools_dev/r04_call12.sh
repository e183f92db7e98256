#!/bin/bash
# round 4, call 12: configs[4] kernels (Cin = 8 bf16x3 Conv3d, fused heads at Z = 32): parity + hi-res hot path
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_configs.py tests/test_gpu_linear.py -m gpu -q -k "conv3d or fused_conv or hires or saturat" > gpurun_out/r04_c12_tests.log 2>&1; tail -8 gpurun_out/r04_c12_tests.log
timeout 600 python bench.py --config configs/occ_hires_400x400x32.py --scope hotpath --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_c12_bench_hires_hot.log 2>&1; grep '^{' gpurun_out/r04_c12_bench_hires_hot.log | cut -c1-250
python - <<'PY'
import json
for l in open('gpurun_out/r04_c12_bench_hires_hot.log'):
    if l.startswith('{'):
        d = json.loads(l); print(json.dumps(d.get('mfma_kernels'))); print(d['roofline']['launch_ms'], d['roofline'].get('tsa_launch_ms'))
PY
