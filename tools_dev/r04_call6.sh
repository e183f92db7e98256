#!/bin/bash
# round 4, call 6: staggered block starts in the chain kernels
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "0 0" "2 0" "4 0" "8 0" "16 0" "0 2" "0 4" "4 2" "8 4" "12 0" "24 0"; do
  set -- $cfg
  echo "stagger_a $1 stagger_b $2" >> gpurun_out/r04_c6_stagger.txt
  OCC_CHAIN_STAGGER_A=$1 OCC_CHAIN_STAGGER_B=$2 CHAIN_ROUNDS=0 timeout 300 python tools_dev/chain_probe.py 2>&1 | grep "rows  \(32768\|40000\)" >> gpurun_out/r04_c6_stagger.txt
done
cat gpurun_out/r04_c6_stagger.txt
