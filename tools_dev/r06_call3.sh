#!/bin/bash
# round 6 call 3: what does the q16 decode (17 VALU per 16-byte load against 9) cost the SCA gather?  A/B on one box,
# the q16 kernel decoding the fp16 planes (timing only, results are garbage)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r06_c3
for rep in 1 2; do
for q in 0 1; do
  OCC_SCA_Q16_TIMING=$q timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/${T}_hot_q${q}_${rep}.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/${T}_hot_q${q}_${rep}.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('q16=$q rep=$rep', 'ms/step', round(d['ms_per_step'],4), 'sca launch_ms', round(d['roofline']['launch_ms'],5), 'passes', [round(x,4) for x in d['passes']['ms_per_step']])
else:
    print('q16=$q rep=$rep FAILED'); print(open('gpurun_out/${T}_hot_q${q}_${rep}.log').read()[-1500:])
PY
done; done
