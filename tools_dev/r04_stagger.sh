#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04_c16}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -o /tmp/lds_probe tools_dev/lab/lds_alloc_probe.hip 2>/dev/null && /tmp/lds_probe > gpurun_out/${T}_lds_alloc.txt 2>&1; cat gpurun_out/${T}_lds_alloc.txt | head -44
: > gpurun_out/${T}_stagger.txt
for st in 0 4 8 12 16 24 32 0; do
  echo "OCC_CHAIN_STAGGER=$st (x 1024 clocks)" >> gpurun_out/${T}_stagger.txt
  OCC_CHAIN_STAGGER=$st CHAIN_FLOOR=0 CHAIN_ROUNDS=0 timeout 200 python tools_dev/chain_probe.py 2>&1 | grep "^rows" >> gpurun_out/${T}_stagger.txt
done
cat gpurun_out/${T}_stagger.txt
