#!/bin/bash
# round 6 call 5: q16 rows v2 — tests + A/B timing incl. kernel trace of the projection
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r06_c5
( time timeout 1500 python -m pytest tests/test_gpu_q16.py tests/test_gpu_value_range.py tests/test_gpu_modules.py -m gpu -q -s ) > gpurun_out/${T}_tests.log 2>&1; grep -E "q16|passed|failed|Error" gpurun_out/${T}_tests.log | cut -c1-260 | tail -40
for rows in f16 q16 f16 q16; do
  OCC_SCA_VALUES=$rows timeout 600 python bench.py --scope hotpath --steps 30 --warmup 5 --no-extras > gpurun_out/${T}_hot_${rows}.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/${T}_hot_${rows}.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('$rows', 'ms/step', round(d['ms_per_step'],4), 'sca', round(d['roofline']['launch_ms'],5), 'passes', [round(x,4) for x in d['passes']['ms_per_step']])
    hp = d.get('headline_feature_parity') or d.get('cpu_baseline',{}).get('headline_feature_parity')
    print('   parity', hp and hp.get('max_abs_diff_by_value_rows'))
else:
    print('$rows FAILED'); print(open('gpurun_out/${T}_hot_${rows}.log').read()[-2500:])
PY
done
for rows in f16 q16; do
(cd /tmp && OCC_SCA_VALUES=$rows timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$rows -o r -- python $GRAFT_REPO_ROOT/bench.py --scope hotpath --steps 10 --warmup 3 --passes 1 --no-cpu-baseline --no-kernel-timing --no-extras > $GRAFT_REPO_ROOT/gpurun_out/${T}_trace_$rows.log 2>&1)
DB=$(find /tmp/prof_$rows -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 60 --last-ms 25 > gpurun_out/${T}_hot_kernel_trace_${rows}.txt 2>&1; head -14 gpurun_out/${T}_hot_kernel_trace_${rows}.txt | cut -c1-140
done
