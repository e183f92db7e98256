#!/bin/bash
# round 6 call 6: full gpu suite under the q16 default; experiment: SCA gather with only the fine levels gathered (timing)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r06_c6
for xp in 0 1 0 1; do
  OCC_SCA_VALUES=f16 OCC_SCA_EXPERIMENT=$xp timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/${T}_hot_xp${xp}.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/${T}_hot_xp${xp}.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('fine-levels-only=$xp', 'ms/step', round(d['ms_per_step'],4), 'sca launch_ms', round(d['roofline']['launch_ms'],5))
else:
    print('xp=$xp FAILED'); print(open('gpurun_out/${T}_hot_xp${xp}.log').read()[-1500:])
PY
done
( time timeout 2000 python -m pytest tests -m gpu -q ) > gpurun_out/${T}_tests.log 2>&1; tail -4 gpurun_out/${T}_tests.log | cut -c1-300
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/${T}_bench_e2e.log 2>&1; grep '^{' gpurun_out/${T}_bench_e2e.log | cut -c1-300
