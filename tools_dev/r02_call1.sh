#!/bin/bash
# round-2 first contact: GPU test suite (with the new full-size cases), e2e bench with the new cpu_baseline /
# roofline fields, steady-state kernel trace with per-dispatch dumps of the GEMM-shaped kernels.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
nproc > gpurun_out/r02_nproc.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q -s ) > gpurun_out/r02_tests1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_tests1.log
tail -5 gpurun_out/r02_tests1.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r02_bench1_e2e.log 2>&1
tail -4 gpurun_out/r02_bench1_e2e.log | cut -c1-1500
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_e2e -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/r02_trace1.log 2>&1)
DB=$(find /tmp/prof_e2e -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 60 --last-ms 60 > gpurun_out/r02_trace1_summary.txt 2>&1
for k in linear_bf16x3 value_proj sca_fused tsa_fused conv3d occ_heads; do python tools_dev/rocpd_summary.py $DB --dump $k 40; done > gpurun_out/r02_trace1_dispatch.txt 2>&1
head -30 gpurun_out/r02_trace1_summary.txt | cut -c1-150
