#!/bin/bash
# round 4, call 13: training decoder on own kernels: gradient parity + train bench (own vs OCC_TRAIN_DECODER=torch) + kernel trace
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_training.py -m gpu -q -k "autograd or gradient or training or train or fingerprint" > gpurun_out/r04_c13_tests.log 2>&1; tail -8 gpurun_out/r04_c13_tests.log
timeout 600 python bench.py --mode train --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r04_c13_bench_train.log 2>&1; grep '^{' gpurun_out/r04_c13_bench_train.log | cut -c1-260
OCC_TRAIN_DECODER=torch timeout 600 python bench.py --mode train --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r04_c13_bench_train_torchdec.log 2>&1; grep '^{' gpurun_out/r04_c13_bench_train_torchdec.log | cut -c1-260
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o r -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/r04_c13_trace.log 2>&1)
DB=$(find /tmp/prof_train -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 70 --last-ms 300 > gpurun_out/r04_c13_train_kernel_trace_stats.txt 2>&1; head -60 gpurun_out/r04_c13_train_kernel_trace_stats.txt | cut -c1-160
