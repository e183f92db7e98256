#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_modules.py -q -k "fp16" -s ) > gpurun_out/r02_tests7.log 2>&1; grep -E "fp16|passed|failed" gpurun_out/r02_tests7.log | cut -c1-200
for sw in raster polar; do
OCC_SCA_VALUES=f16 OCC_BEV_SWEEP=$sw timeout 300 python bench.py --scope hotpath --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench7_hot_f16_$sw.log 2>&1; python - <<PY
import json
for l in open('gpurun_out/r02_bench7_hot_f16_$sw.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$sw f16', d['ms_per_step'], r['launch_ms'], r['frac_l1'])
PY
done
timeout 300 python bench.py --scope hotpath --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench7_hot.log 2>&1; tail -1 gpurun_out/r02_bench7_hot.log | cut -c1-300
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o r -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/r02_trace7_train.log 2>&1)
tail -2 gpurun_out/r02_trace7_train.log | cut -c1-300
DB=$(find /tmp/prof_tr -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 70 --last-ms 400 > gpurun_out/r02_trace7_train_summary.txt 2>&1; head -75 gpurun_out/r02_trace7_train_summary.txt | cut -c1-170
