#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for k in persistent block; do for d in 0 1 2; do
  echo "kernel $k dbg $d (1 = no row stores, 2 = non-temporal stores)" >> gpurun_out/r04_c9_stores.txt
  OCC_CHAIN_KERNEL=$k OCC_CHAIN_DBG=$d CHAIN_ROUNDS=0 CHAIN_FLOOR=0 timeout 300 python tools_dev/chain_probe.py 2>&1 | grep "rows" >> gpurun_out/r04_c9_stores.txt
done; done
cat gpurun_out/r04_c9_stores.txt
