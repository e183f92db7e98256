#!/bin/bash
# round 6 call 19: fused training nodes of the autocast backbone — tests, then training bench A/B on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r06_c19
timeout 900 python -m pytest tests/test_gpu_backbone.py -m gpu -q -x -k "bias_act or conv_bn_act or training_backbone" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_training.py -m gpu -q -x 2>&1 | tail -5
for v in 0 1 0 1; do
  OCC_TRAIN_FUSED_CONV=$v timeout 600 python bench.py --mode train --steps 6 --warmup 3 --passes 3 --no-cpu-baseline > gpurun_out/${T}_train_$v.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/${T}_train_$v.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('fused=$v', 'ms/step', round(d['ms_per_step'],3), 'samples/s', round(d['value'],3))
else:
    print('$v FAILED'); print(open('gpurun_out/${T}_train_$v.log').read()[-2500:])
PY
done
