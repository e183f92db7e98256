#!/bin/bash
# fused dropout + residual + LayerNorm training nodes: tests, then training bench A/B (OCC_TRAIN_FUSED_LN)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r06_ln}
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_modules.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25
for v in 0 1 0 1; do
  OCC_TRAIN_FUSED_LN=$v timeout 600 python bench.py --mode train --steps 6 --warmup 3 --passes 3 --no-cpu-baseline > gpurun_out/${T}_train_ln$v.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/${T}_train_ln$v.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('fused_ln=$v', 'ms/step', round(d['ms_per_step'],3), 'samples/s', round(d['value'],3))
else:
    print('$v FAILED'); print(open('gpurun_out/${T}_train_ln$v.log').read()[-2500:])
PY
done
