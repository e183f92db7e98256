#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/gpurun_out/r05_c12
( timeout 300 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_value_range.py tests/test_gpu_concurrency.py -m gpu -q -s 2>&1 | grep -E "reference golden|passed|failed|Error|assert" | cut -c1-200 ) > ${T}_tests.log 2>&1; tail -22 ${T}_tests.log
H="timeout 150 python tools_dev/hazard_matrix.py 150"
hz() { name=$1; shift; ( env "$@" $H $name ) > ${T}_hz_$name.log 2>&1; grep -E "HAZARD|   rep" ${T}_hz_$name.log | cut -c1-200 | head -4; }
hz default_copy_gemm A=1
hz default_ownvproj_load HZ_LOAD=vproj
B="timeout 100 python bench.py --scope hotpath --steps 40 --warmup 6 --no-cpu-baseline --no-extras"
run() { name=$1; dir=$2; shift; shift; ( cd $dir && env "$@" $B ) > ${T}_hot_$name.log 2>&1; echo "$name: $(grep '^{' ${T}_hot_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); m=d["mfma_kernels"]; r=d["roofline"]; print(round(d["ms_per_step"],3), "ms/step; enq", d.get("host_enqueue_ms_per_step"), "lin", round(m["linear_ms_per_step"],3), "sca", round(r["launch_ms"],4), "tsa", round(r["tsa_launch_ms"],4))' 2>/dev/null || tail -2 ${T}_hot_$name.log)"; }
run r4_a _r4 A=1
run head_a . A=1
run r4_b _r4 A=1
run head_b . A=1
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_hot -- python $GRAFT_REPO_ROOT/bench.py --scope hotpath --steps 20 --warmup 4 --no-cpu-baseline --no-extras --no-kernel-timing > ${T}_prof.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find /tmp/prof_hot -name "*.db" | head -1); python tools_dev/rocpd_summary.py $DB 40 --last-ms 40 > ${T}_hot_kernel_trace_stats.txt 2>&1; head -16 ${T}_hot_kernel_trace_stats.txt | cut -c1-150
