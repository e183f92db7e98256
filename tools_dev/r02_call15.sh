#!/bin/bash
# round 2, call 15: rows gather-sum kernel in the SCA training path, 4 samples per step in the per-sample backward kernel, software-pipelined replay
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout -k 5 400 python -m pytest tests/test_gpu_linear.py tests/test_gpu_backward.py tests/test_gpu_training.py tests/test_gpu_fullsize.py -m gpu -q -k "not (base_geometry or hires or images_to_voxels)" ) > gpurun_out/r02c15_tests.log 2>&1; tail -5 gpurun_out/r02c15_tests.log | cut -c1-200
timeout -k 5 200 python bench.py --mode train --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r02c15_train.log 2>&1; grep '^{' gpurun_out/r02c15_train.log | cut -c1-260
(cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o r -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/r02c15_trace.log 2>&1)
DB=$(find /tmp/prof_tr -name "*.db" | head -1)
timeout -k 5 120 python tools_dev/rocpd_summary.py $DB 400 --last-ms 400 > gpurun_out/r02c15_train_trace_summary.txt 2>&1; head -14 gpurun_out/r02c15_train_trace_summary.txt | cut -c1-170
