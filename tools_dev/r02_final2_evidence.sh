#!/bin/bash
# round-2 final evidence (second pass, after the training-step work): full GPU test suite, e2e / hot-path / u8 /
# training bench lines, steady-state kernel traces (inference + training), PMC of the reworked backward kernels.
# Every profiler / long command under `timeout -k 5`.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout -k 5 900 python -m pytest tests -m gpu -q ) > gpurun_out/r02f2_tests.log 2>&1; tail -3 gpurun_out/r02f2_tests.log | cut -c1-200
( time timeout -k 5 400 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r02f2_bench_e2e.log 2>&1; grep '^{' gpurun_out/r02f2_bench_e2e.log | cut -c1-200
timeout -k 5 200 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02f2_bench_hot.log 2>&1; grep '^{' gpurun_out/r02f2_bench_hot.log | cut -c1-160
timeout -k 5 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --input u8-h2d > gpurun_out/r02f2_bench_e2e_u8.log 2>&1; grep '^{' gpurun_out/r02f2_bench_e2e_u8.log | cut -c1-160
timeout -k 5 300 python bench.py --mode train --steps 10 --warmup 4 --no-cpu-baseline > gpurun_out/r02f2_bench_train.log 2>&1; grep '^{' gpurun_out/r02f2_bench_train.log | cut -c1-240
(cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_e2e -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/r02f2_trace.log 2>&1)
DB=$(find /tmp/prof_e2e -name "*.db" | head -1)
timeout -k 5 120 python tools_dev/rocpd_summary.py $DB 60 --last-ms 60 > gpurun_out/r02f2_trace_summary.txt 2>&1; head -12 gpurun_out/r02f2_trace_summary.txt | cut -c1-150
(cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o r -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/r02f2_train_trace.log 2>&1)
DB=$(find /tmp/prof_tr -name "*.db" | head -1)
timeout -k 5 120 python tools_dev/rocpd_summary.py $DB 400 --last-ms 400 > gpurun_out/r02f2_train_trace_summary.txt 2>&1; head -10 gpurun_out/r02f2_train_trace_summary.txt | cut -c1-150
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCC_ATOMIC_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && OCC_CUDNN_BENCHMARK=0 timeout -k 5 200 rocprofv3 --pmc $set --kernel-trace --output-format csv --kernel-include-regex "msda_bwd" -d /tmp/pmcb_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/r02f2_pmc_$i.log 2>&1)
  f=$(find /tmp/pmcb_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r02f2_pmc_${i}_counters.csv
done
ls gpurun_out | grep r02f2 | wc -l
