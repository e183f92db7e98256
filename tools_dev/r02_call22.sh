#!/bin/bash
# round 2, call 22: final training number with everything on (solver search + SCA prep kernels)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 5 110 python bench.py --mode train --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r02c22_train.log 2>&1; grep '^{' gpurun_out/r02c22_train.log | cut -c1-260
