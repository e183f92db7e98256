#!/bin/bash
# round 6 call 23: msda backward replay with one record load per 16 items — backward parity tests, training bench A/B (library swap)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r06_c23
cp tools_dev/bin/libocc_amd_new.so occnet_amd/lib/libocc_amd.so
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_training.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
for v in base new base new; do
  cp tools_dev/bin/libocc_amd_$v.so occnet_amd/lib/libocc_amd.so
  timeout 600 python bench.py --mode train --steps 6 --warmup 3 --passes 3 --no-cpu-baseline > gpurun_out/${T}_train_$v.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/${T}_train_$v.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('$v', 'ms/step', round(d['ms_per_step'],3), 'samples/s', round(d['value'],3))
else:
    print('$v FAILED'); print(open('gpurun_out/${T}_train_$v.log').read()[-2500:])
PY
done
cp tools_dev/bin/libocc_amd_new.so occnet_amd/lib/libocc_amd.so
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o r -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 6 --warmup 3 --passes 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${T}_trace.log 2>&1)
DB=$(find /tmp/prof_tr -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 30 --last-ms 200 2>&1 | grep -E "msda|total" | cut -c1-150
