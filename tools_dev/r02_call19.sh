#!/bin/bash
# round 2, call 19: the four value projections prefetched on a side stream (OCC_VPROJ_OVERLAP=1) vs serial
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time OCC_VPROJ_OVERLAP=1 timeout -k 5 400 python -m pytest tests/test_gpu_modules.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -m gpu -q -k "not (hires or images_to_voxels or backward)" ) > gpurun_out/r02c19_tests.log 2>&1; tail -4 gpurun_out/r02c19_tests.log | cut -c1-200
for v in 0 1 0 1; do
  OCC_VPROJ_OVERLAP=$v timeout -k 5 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/r02c19_e2e_$v.log 2>&1; grep '^{' gpurun_out/r02c19_e2e_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('e2e overlap=$v', d['value'], d['ms_per_step'])"
done
