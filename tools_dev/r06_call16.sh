#!/bin/bash
# round 6 call 16: persistent head-major SCA gather — parity tests, then A/B/A sweep on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r06_c16
timeout 900 python -m pytest tests/test_gpu_modules.py -m gpu -q -x -k "sca or gather" 2>&1 | tail -5
for cfg in "0 0" "1 0" "1 1" "0 0" "1 0" "1 1"; do
  set -- $cfg
  OCC_SCA_PERSIST=$1 OCC_SCA_HMP_VARIANT=$2 timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/${T}_hot_$1_$2.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/${T}_hot_$1_$2.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('persist=$1 variant=$2', 'ms/step', round(d['ms_per_step'],4), 'sca launch_ms', round(d['roofline']['launch_ms'],5))
else:
    print('$1 $2 FAILED'); print(open('gpurun_out/${T}_hot_$1_$2.log').read()[-1500:])
PY
done
