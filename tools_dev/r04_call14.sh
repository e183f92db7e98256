#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_training.py -m gpu -q -s -k "autograd or gradient or rebatch or train_step or ddp" > gpurun_out/r04_c14_tests.log 2>&1; grep -E "conv3d autograd|passed|failed|FAILED|decoder" gpurun_out/r04_c14_tests.log | tail -20
timeout 600 python bench.py --mode train --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r04_c14_bench_train.log 2>&1; grep '^{' gpurun_out/r04_c14_bench_train.log | cut -c1-200
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o r -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/r04_c14_trace.log 2>&1)
DB=$(find /tmp/prof_train -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 40 --last-ms 300 > gpurun_out/r04_c14_train_kernel_trace_stats.txt 2>&1; head -16 gpurun_out/r04_c14_train_kernel_trace_stats.txt | cut -c1-150
