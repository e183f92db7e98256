#!/bin/bash
# chain kernels after an edit: parity tests, probe against the separate launches (+ stamped timeline), hot-path bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04_chain}
timeout 600 python -m pytest tests/test_gpu_linear.py tests/test_gpu_modules.py -m gpu -q -x 2>&1 | tail -3
CHAIN_SWEEP=1 CHAIN_ROUNDS=1 CHAIN_TRACE=${CHAIN_TRACE:-0} CHAIN_TRACE_ROWS=40000 timeout 300 python tools_dev/chain_probe.py > gpurun_out/${T}_chain_probe.txt 2>&1; grep -v amdgpu.ids gpurun_out/${T}_chain_probe.txt | cut -c1-200
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/${T}_bench_hot.log 2>&1; grep '^{' gpurun_out/${T}_bench_hot.log | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('hot', d['value'], d['ms_per_step'])"
