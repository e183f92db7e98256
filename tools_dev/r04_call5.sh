#!/bin/bash
# round 4, call 5: chain kernel with row-major transfers through the per-wave LDS scratch: parity, sweep, ablation 3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_linear.py -m gpu -q -k "chain" -x > gpurun_out/r04_c5_tests.log 2>&1; tail -5 gpurun_out/r04_c5_tests.log
CHAIN_ROUNDS=1 timeout 300 python tools_dev/chain_probe.py > gpurun_out/r04_c5_chain_probe.txt 2>&1; cat gpurun_out/r04_c5_chain_probe.txt
for a in 2 3; do
  echo "ablate $a" >> gpurun_out/r04_c5_chain_ablate.txt
  OCC_CHAIN_ABLATE=$a CHAIN_ROUNDS=0 timeout 300 python tools_dev/chain_probe.py 2>&1 | grep rows >> gpurun_out/r04_c5_chain_ablate.txt
done
cat gpurun_out/r04_c5_chain_ablate.txt
