#!/bin/bash
# round 3, call 5: TSA fp16 rows + stacked value projection: tests, probes, bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > gpurun_out/r03c5_tests.log 2>&1; tail -6 gpurun_out/r03c5_tests.log | cut -c1-200
grep -h "fp16-row\|base 4 layers\|base 1 layer\|reference golden" gpurun_out/r03c5_tests.log | cut -c1-160 | head -20
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c5_bench_e2e.log 2>&1; grep '^{' gpurun_out/r03c5_bench_e2e.log | cut -c1-220
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c5_bench_hot.log 2>&1; grep '^{' gpurun_out/r03c5_bench_hot.log | cut -c1-220
OCC_VPROJ_OVERLAP=0 timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c5_bench_hot_nooverlap.log 2>&1; grep '^{' gpurun_out/r03c5_bench_hot_nooverlap.log | cut -c1-220
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_e2e -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/r03c5_trace.log 2>&1)
DB=$(find /tmp/prof_e2e -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 60 --last-ms 60 > gpurun_out/r03c5_trace_summary.txt 2>&1; head -40 gpurun_out/r03c5_trace_summary.txt | cut -c1-150
