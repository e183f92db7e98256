#!/bin/bash
# round 3, call 10: stagger sweep of the activation-resident value projection
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for st in 0 64 128 192 256 0 128; do
  echo "stagger $st" >> gpurun_out/r03c10_vproj_stagger.txt
  OCC_VPROJ_STAGGER=$st timeout 200 python tools_dev/vproj_probe.py 60 2>&1 | grep '^{' >> gpurun_out/r03c10_vproj_stagger.txt
done
cat gpurun_out/r03c10_vproj_stagger.txt | cut -c1-200
