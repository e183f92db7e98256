#!/bin/bash
# round 4, call 8: memory floor of the chain's traffic with stock streaming kernels; both chain kernels beside it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
CHAIN_SWEEP=0 CHAIN_ROUNDS=1 timeout 300 python tools_dev/chain_probe.py > gpurun_out/r04_c8_floor.txt 2>&1
OCC_CHAIN_KERNEL=block CHAIN_SWEEP=0 CHAIN_ROUNDS=1 CHAIN_FLOOR=0 timeout 300 python tools_dev/chain_probe.py >> gpurun_out/r04_c8_floor.txt 2>&1
cat gpurun_out/r04_c8_floor.txt
