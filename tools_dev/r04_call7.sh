#!/bin/bash
# round 4, call 7: persistent wave-specialised chain kernel: parity + probe, A/B against the block-per-tile kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_linear.py -m gpu -q -k "chain" -x > gpurun_out/r04_c7_tests.log 2>&1; tail -5 gpurun_out/r04_c7_tests.log
CHAIN_ROUNDS=1 timeout 300 python tools_dev/chain_probe.py > gpurun_out/r04_c7_chain_probe.txt 2>&1; cat gpurun_out/r04_c7_chain_probe.txt
echo "--- OCC_CHAIN_KERNEL=block" >> gpurun_out/r04_c7_chain_probe.txt
OCC_CHAIN_KERNEL=block CHAIN_ROUNDS=0 timeout 300 python tools_dev/chain_probe.py 2>&1 | grep rows | tee -a gpurun_out/r04_c7_chain_probe.txt
