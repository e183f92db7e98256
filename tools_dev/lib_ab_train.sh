#!/bin/bash
# training bench with two builds of the library (tools_dev/bin/libocc_amd_{base,new}.so), alternating on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in base new base new; do
  cp tools_dev/bin/libocc_amd_$v.so occnet_amd/lib/libocc_amd.so
  timeout 600 python bench.py --mode train --steps 6 --warmup 3 --passes 3 --no-cpu-baseline > gpurun_out/abt_$v.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/abt_$v.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('$v', 'ms/step', round(d['ms_per_step'],3), 'samples/s', round(d['value'],3))
else:
    print('$v FAILED'); print(open('gpurun_out/abt_$v.log').read()[-2500:])
PY
done
