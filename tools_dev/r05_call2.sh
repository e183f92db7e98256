#!/bin/bash
# Round 5, GPU call 2: (1) does the default path's concurrency hazard pre-date this round (r4 worktree)?  (2) op-level
# bisect of the default path under load, per load kind / side stream / runtime env; (3) same-box A/B r4 vs HEAD hot path +
# kernel trace of HEAD; (4) layered value-projection schedules; (5) row-pipeline serial bisect under two runtime switches.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/gpurun_out/r05_c2
( cd _r4 && timeout 120 python -m pytest tests/test_gpu_concurrency.py -m gpu -q -x 2>&1 | tail -4 | cut -c1-300 ) > ${T}_r4_conc.log 2>&1; cat ${T}_r4_conc.log
P="timeout 150 python tools_dev/concurrency_probe.py"
probe() { name=$1; shift; ( env "$@" $P $KINDS 4 ) > ${T}_probe_$name.log 2>&1; echo "== $name"; grep -E "^rep|^   |solo twice" ${T}_probe_$name.log | cut -c1-260 | head -14; }
KINDS="copy,gemm,gather,sort" probe all A=1
KINDS="copy,gemm,gather,sort" probe all_nooverlap OCC_VPROJ_OVERLAP=0
KINDS="copy" probe copy A=1
KINDS="gemm" probe gemm A=1
KINDS="sort" probe sort A=1
KINDS="copy,gemm,gather,sort" probe all_hostkernarg HIP_FORCE_DEV_KERNARG=0
KINDS="copy,gemm,gather,sort" probe all_1queue GPU_MAX_HW_QUEUES=1
B="timeout 100 python bench.py --scope hotpath --steps 40 --warmup 6 --no-cpu-baseline --no-extras"
run() { name=$1; dir=$2; shift; shift; ( cd $dir && env "$@" $B ) > ${T}_hot_$name.log 2>&1; echo "$name: $(grep '^{' ${T}_hot_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); m=d["mfma_kernels"]; r=d["roofline"]; print(round(d["ms_per_step"],3), "ms/step; enq", d.get("host_enqueue_ms_per_step"), "lin", round(m["linear_ms_per_step"],3), "sca", round(r["launch_ms"],4), "tsa", round(r["tsa_launch_ms"],4), d["config"].get("vproj_schedule"))' 2>/dev/null || tail -2 ${T}_hot_$name.log)"; }
run r4_a _r4 A=1
run head_a . A=1
run r4_b _r4 A=1
run head_b . A=1
run layered . OCC_VPROJ_SCHEDULE=layered
run layered_early . OCC_VPROJ_SCHEDULE=layered_early
run layered_gather . OCC_VPROJ_SCHEDULE=layered_gather
run head_c . A=1
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_hot -- python $GRAFT_REPO_ROOT/bench.py --scope hotpath --steps 20 --warmup 4 --no-cpu-baseline --no-extras --no-kernel-timing > ${T}_prof.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find /tmp/prof_hot -name "*.db" | head -1); python tools_dev/rocpd_summary.py $DB 40 --last-ms 40 > ${T}_hot_kernel_trace_stats.txt 2>&1; head -22 ${T}_hot_kernel_trace_stats.txt | cut -c1-170
( HIP_FORCE_DEV_KERNARG=0 timeout 150 python tools_dev/row_pipeline_bisect.py 2 2 ) > ${T}_bisect_hostkernarg.log 2>&1; grep "mode=serial" ${T}_bisect_hostkernarg.log | cut -c1-200
( GPU_MAX_HW_QUEUES=1 timeout 150 python tools_dev/row_pipeline_bisect.py 2 2 ) > ${T}_bisect_1queue.log 2>&1; grep "mode=serial" ${T}_bisect_1queue.log | cut -c1-200
