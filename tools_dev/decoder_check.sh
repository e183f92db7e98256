#!/bin/bash
# decoder kernels after an edit: parity tests, hot-path bench lines of the base and the hi-res config (mfma_kernels)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04_dec}
timeout 900 python -m pytest tests/test_gpu_decoder.py -m gpu -q -x 2>&1 | tail -3
for cfg in occ_base_200x200x16 occ_hires_400x400x32; do
timeout 300 python bench.py --config configs/$cfg.py --scope hotpath --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/${T}_bench_$cfg.log 2>&1; grep '^{' gpurun_out/${T}_bench_$cfg.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); m=d['mfma_kernels']; print('$cfg hot', round(d['value'],1), round(d['ms_per_step'],4), {k:(round(v['launch_ms'],4), round(v['frac'],3)) for k,v in m.items() if isinstance(v,dict)})"
done
