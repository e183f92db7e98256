#!/bin/bash
# round 6 call 4: q16 rows — encoders vs the numpy restatement, gather vs decoded rows, parity over scales, A/B timing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r06_c4
( time timeout 1500 python -m pytest tests/test_gpu_q16.py tests/test_gpu_value_range.py tests/test_gpu_modules.py -m gpu -q -s ) > gpurun_out/${T}_tests.log 2>&1; grep -E "q16|rows, amp|passed|failed|Error" gpurun_out/${T}_tests.log | cut -c1-260 | tail -60
for rows in f16 q16; do
  OCC_SCA_VALUES=$rows timeout 600 python bench.py --scope hotpath --steps 30 --warmup 5 --no-extras > gpurun_out/${T}_hot_${rows}.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/${T}_hot_${rows}.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('$rows', 'ms/step', round(d['ms_per_step'],4), 'sca', round(d['roofline']['launch_ms'],5), 'passes', [round(x,4) for x in d['passes']['ms_per_step']])
    print('   parity', d.get('headline_feature_parity',{}).get('max_abs_diff_by_value_rows'))
else:
    print('$rows FAILED'); print(open('gpurun_out/${T}_hot_${rows}.log').read()[-2500:])
PY
done
