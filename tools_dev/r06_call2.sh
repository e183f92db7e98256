#!/bin/bash
# round 6 call 2: producer-side max|x| (no range pass), device-side weight terms, fdiv guard -> affected tests + bench lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r06_c2
( time timeout 1500 python -m pytest tests/test_gpu_value_range.py tests/test_gpu_backbone.py tests/test_gpu_modules.py tests/test_gpu_concurrency.py tests/test_gpu_configs.py tests/test_gpu_msda.py -m gpu -q -x ) > gpurun_out/${T}_tests.log 2>&1; tail -4 gpurun_out/${T}_tests.log | cut -c1-300
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_hot.log 2>&1; grep '^{' gpurun_out/${T}_bench_hot.log | cut -c1-400
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/${T}_bench_e2e.log 2>&1; grep '^{' gpurun_out/${T}_bench_e2e.log | cut -c1-400
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_e2e -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --passes 1 --no-cpu-baseline --no-kernel-timing --no-extras > $GRAFT_REPO_ROOT/gpurun_out/${T}_trace.log 2>&1)
DB=$(find /tmp/prof_e2e -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 60 --last-ms 60 > gpurun_out/${T}_e2e_kernel_trace_stats.txt 2>&1; head -30 gpurun_out/${T}_e2e_kernel_trace_stats.txt | cut -c1-150
