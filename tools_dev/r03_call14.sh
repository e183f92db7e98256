#!/bin/bash
# round 3, call 14: 128-column blocks for the non-LayerNorm Linears (OCC_LINEAR_NARROW=1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for d in 1 0 1 0; do
  echo "narrow $d" >> gpurun_out/r03c14_linear_narrow.txt
  OCC_LINEAR_NARROW=$d timeout 200 python tools_dev/linear_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03c14_linear_narrow.txt
done
cat gpurun_out/r03c14_linear_narrow.txt | cut -c1-150
