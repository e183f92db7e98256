#!/bin/bash
# round 3, call 16: floor of the tiled Linear kernel (temporary instrumentation): 31 = no barriers / loads / MFMAs / stores,
# +32 = no K loop at all, +64 = no epilogue; event timing and rocprofv3 kernel-trace durations
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for a in 31 63 95 127 0; do
  echo "ablate $a" >> gpurun_out/r03c16_linear_floor.txt
  OCC_LINEAR_ABLATE=$a timeout 200 python tools_dev/linear_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-75 >> gpurun_out/r03c16_linear_floor.txt
  (cd /tmp && OCC_LINEAR_ABLATE=$a timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_l$a -o r -- python $GRAFT_REPO_ROOT/tools_dev/linear_probe.py > /dev/null 2>&1)
  DB=$(find /tmp/prof_l$a -name "*.db" | head -1)
  python tools_dev/rocpd_summary.py $DB 6 2>&1 | grep -i "linear_bf16x3\|total" | cut -c1-150 >> gpurun_out/r03c16_linear_floor.txt
done
cat gpurun_out/r03c16_linear_floor.txt
