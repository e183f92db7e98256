#!/bin/bash
# Round 5, GPU call 3: new kernels (range kernel v2, TSA tile kernel) + the two open hazards.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/gpurun_out/r05_c3
( timeout 300 python -m pytest tests/test_gpu_value_range.py tests/test_gpu_msda.py "tests/test_gpu_modules.py" -m gpu -q -s 2>&1 | grep -E "tsa tile|rows, amp|passed|failed|Error|assert" | cut -c1-220 ) > ${T}_tests.log 2>&1; tail -40 ${T}_tests.log
for i in 1 2 3; do timeout 120 python -m pytest tests/test_gpu_concurrency.py -m gpu -q -x 2>&1 | grep -E "passed|failed|outputs changed" | cut -c1-400; done > ${T}_conc3.log 2>&1; cat ${T}_conc3.log
( timeout 200 python tools_dev/concurrency_probe.py fresh,copy,gemm 16 ) > ${T}_probe_fresh_head.log 2>&1; grep -E "^rep .*first|^   |solo twice" ${T}_probe_fresh_head.log | grep -v "= None" | cut -c1-260 | head -20; grep -c "first differing op = None" ${T}_probe_fresh_head.log
( cd _r4 && timeout 200 python tools_dev/concurrency_probe.py fresh,copy,gemm 16 ) > ${T}_probe_fresh_r4.log 2>&1; grep -E "^rep .*first|^   |solo twice" ${T}_probe_fresh_r4.log | grep -v "= None" | cut -c1-260 | head -20; grep -c "first differing op = None" ${T}_probe_fresh_r4.log
( timeout 200 python tools_dev/row_pipeline_bisect.py 2 3 serial,serial+dummy,serial ) > ${T}_bisect_dummy.log 2>&1; grep "mode=" ${T}_bisect_dummy.log | cut -c1-220
B="timeout 100 python bench.py --scope hotpath --steps 40 --warmup 6 --no-cpu-baseline --no-extras"
run() { name=$1; dir=$2; shift; shift; ( cd $dir && env "$@" $B ) > ${T}_hot_$name.log 2>&1; echo "$name: $(grep '^{' ${T}_hot_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); m=d["mfma_kernels"]; r=d["roofline"]; print(round(d["ms_per_step"],3), "ms/step; enq", d.get("host_enqueue_ms_per_step"), "lin", round(m["linear_ms_per_step"],3), "sca", round(r["launch_ms"],4), "tsa", round(r["tsa_launch_ms"],4))' 2>/dev/null || tail -2 ${T}_hot_$name.log)"; }
run r4_a _r4 A=1
run head_tile . A=1
run head_wave . OCC_TSA_TILE=0
run r4_b _r4 A=1
run head_tile_b . A=1
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_hot -- python $GRAFT_REPO_ROOT/bench.py --scope hotpath --steps 20 --warmup 4 --no-cpu-baseline --no-extras --no-kernel-timing > ${T}_prof.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find /tmp/prof_hot -name "*.db" | head -1); python tools_dev/rocpd_summary.py $DB 40 --last-ms 40 > ${T}_hot_kernel_trace_stats.txt 2>&1; head -16 ${T}_hot_kernel_trace_stats.txt | cut -c1-150
