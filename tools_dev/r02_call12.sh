#!/bin/bash
# round 2, call 12: training step on own Linear forward/backward kernels + folded eval-BN + frozen prefix on the plan
# kernels: parity tests, ablation of the three switches on one box, kernel trace of the new step, PMC of the
# deformable-attention backward passes on the real training pattern, TA row-gather probe.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_linear.py tests/test_gpu_backbone.py tests/test_gpu_training.py tests/test_gpu_backward.py -m gpu -q -k "wgrad or autograd or x3linear or training or backward or train" ) > gpurun_out/r02c12_tests.log 2>&1; tail -4 gpurun_out/r02c12_tests.log
run_train() { # name, env...
  name=$1; shift
  env "$@" timeout 400 python bench.py --mode train --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r02c12_train_$name.log 2>&1
  echo "$name: $(grep '^{' gpurun_out/r02c12_train_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"
}
run_train all_on A=1
run_train all_off OCC_TRAIN_LINEAR=torch OCC_TRAIN_FOLD_BN=0 OCC_TRAIN_FROZEN_PREFIX=0
run_train no_linear OCC_TRAIN_LINEAR=torch
run_train no_foldbn OCC_TRAIN_FOLD_BN=0
run_train no_prefix OCC_TRAIN_FROZEN_PREFIX=0
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o r -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/r02c12_trace.log 2>&1)
DB=$(find /tmp/prof_tr -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 400 --last-ms 400 > gpurun_out/r02c12_train_trace_summary.txt 2>&1; head -30 gpurun_out/r02c12_train_trace_summary.txt | cut -c1-170
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "TCC_ATOMIC_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv --kernel-include-regex "msda_bwd" -d /tmp/pmcb_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/r02c12_pmc_$i.log 2>&1)
  f=$(find /tmp/pmcb_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r02c12_pmc_${i}_counters.csv
done
python - > gpurun_out/r02c12_pmc_bwd.txt <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
for path in sorted(glob.glob('gpurun_out/r02c12_pmc_[0-9]*_counters.csv')):
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        a = acc[k][row['Counter_Name']]; v = float(row['Counter_Value']); a[0] += 1; a[1] += v; a[2] = max(a[2], v)
print("# rocprofv3 --pmc on bench.py --mode train (1 warm-up + 1 step), ms_deform_attn_backward kernels: mean and max per launch")
for k, c in sorted(acc.items()):
    print(k)
    for n, (cnt, tot, mx) in sorted(c.items()):
        print(f"    {n:36s} n={cnt:3d} mean={tot/cnt:14.6g} max={mx:14.6g}")
PY
head -60 gpurun_out/r02c12_pmc_bwd.txt
timeout 120 tools_dev/bin/ta_probe 3 > gpurun_out/r02c12_ta_probe_3waves.txt 2>&1; cat gpurun_out/r02c12_ta_probe_3waves.txt | cut -c1-230
timeout 120 tools_dev/bin/ta_probe 8 > gpurun_out/r02c12_ta_probe_8waves.txt 2>&1; tail -12 gpurun_out/r02c12_ta_probe_8waves.txt | cut -c1-230
