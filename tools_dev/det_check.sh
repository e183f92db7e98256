#!/bin/bash
# deterministic msda backward after an edit: backward / training / hazard tests, then the training bench in both modes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r06_det}
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_training.py tests/test_gpu_hazard_repro.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8
for v in 0 1 0 1; do
  OCC_MSDA_BWD_DETERMINISTIC=$v timeout 600 python bench.py --mode train --steps 6 --warmup 3 --passes 3 --no-cpu-baseline > gpurun_out/${T}_train_det$v.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/${T}_train_det$v.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('deterministic=$v', 'ms/step', round(d['ms_per_step'],3), 'samples/s', round(d['value'],3))
else:
    print('$v FAILED'); print(open('gpurun_out/${T}_train_det$v.log').read()[-2500:])
PY
done
