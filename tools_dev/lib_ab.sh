#!/bin/bash
# hot-path bench with two builds of the library (tools_dev/bin/libocc_amd_{base,new}.so), alternating on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in base new base new; do
  cp tools_dev/bin/libocc_amd_$v.so occnet_amd/lib/libocc_amd.so
  timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/ab_$v.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/ab_$v.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('$v', 'ms/step', round(d['ms_per_step'],4), 'sca launch_ms', round(d['roofline']['launch_ms'],5), 'tsa', round(d['roofline'].get('tsa_launch_ms',0),5))
else:
    print('$v FAILED'); print(open('gpurun_out/ab_$v.log').read()[-1500:])
PY
done
