#!/bin/bash
# SCA gather after an edit: module parity tests, probe (back to back), launch time in the step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04_sca}
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_msda.py -m gpu -q -x 2>&1 | tail -2
timeout 300 python tools_dev/sca_probe.py 40 2>&1 | grep "f16 values" | cut -c1-150
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/${T}_bench_hot.log 2>&1; grep '^{' gpurun_out/${T}_bench_hot.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('hot', round(d['value'],1), round(d['ms_per_step'],4), 'SCA launch', round(r['launch_ms'],4), 'TSA', round(r['tsa_launch_ms'],4))"
