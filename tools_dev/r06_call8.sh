#!/bin/bash
# round 6 call 8: head-major SCA gather: parity + A/B timing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r06_c8
( time timeout 900 python -m pytest tests/test_gpu_modules.py -m gpu -q -k "sca_gather_kernels_match_oracle" -s ) > gpurun_out/${T}_tests.log 2>&1; grep -E "kernel |passed|failed|Error" gpurun_out/${T}_tests.log | cut -c1-200 | tail -30
for hm in 0 1 0 1; do
  OCC_SCA_HEAD_MAJOR=$hm timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/${T}_hot_hm${hm}.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/${T}_hot_hm${hm}.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('head-major=$hm', 'ms/step', round(d['ms_per_step'],4), 'sca launch_ms', round(d['roofline']['launch_ms'],5), d['roofline'].get('rows_R'), d['roofline'].get('n_in_corners'))
else:
    print('hm=$hm FAILED'); print(open('gpurun_out/${T}_hot_hm${hm}.log').read()[-1500:])
PY
done
