#!/bin/bash
# round 3, call 15: ablation of the tiled Linear kernel (temporary instrumentation; wrong results by construction, timing only)
# bits: 1 no K-loop barriers, 2 no weight loads, 4 no activation loads, 8 no MFMAs, 16 no output stores
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for a in 0 1 2 4 8 16 9 6 31 0; do
  echo "ablate $a" >> gpurun_out/r03c15_linear_ablation.txt
  OCC_LINEAR_ABLATE=$a timeout 200 python tools_dev/linear_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-75 >> gpurun_out/r03c15_linear_ablation.txt
done
cat gpurun_out/r03c15_linear_ablation.txt
