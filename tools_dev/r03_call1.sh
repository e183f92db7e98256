#!/bin/bash
# round 3, call 1: new parity tests (reference-run goldens at BASE geometry, full-size history chain, N4 goldens),
# the fp16-value SCA gather vs the fp32 one on the base config's real inputs, wave-state PMC of both.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_io_golden.py -m gpu -q -x -s -k "golden or history or submission or u8_stem" ) > gpurun_out/r03c1_tests_new.log 2>&1; tail -5 gpurun_out/r03c1_tests_new.log
( time timeout 600 python -m pytest tests/test_gpu_modules.py tests/test_gpu_backbone.py -m gpu -q -x ) > gpurun_out/r03c1_tests_mod.log 2>&1; tail -3 gpurun_out/r03c1_tests_mod.log
timeout 300 python tools_dev/sca_probe.py 40 > gpurun_out/r03c1_sca_probe.log 2>&1; grep '^{\|^value' gpurun_out/r03c1_sca_probe.log | cut -c1-260
for mode in f32 f16; do
  OCC_SCA_VALUES=$mode timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c1_bench_hot_$mode.log 2>&1; grep '^{' gpurun_out/r03c1_bench_hot_$mode.log | cut -c1-150
done
KR="sca_fused"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  for mode in f32 f16; do
    if [ $mode = f32 ] && [ $i -ge 2 ] && [ $i -le 5 ]; then continue; fi      # f32: only the wave-state sets are new
    (cd /tmp && OCC_SCA_VALUES=$mode timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv --kernel-include-regex "$KR" -d /tmp/pmc_${mode}_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --scope hotpath --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/r03c1_pmc_${mode}_$i.log 2>&1)
    f=$(find /tmp/pmc_${mode}_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r03c1_pmc_${mode}_${i}_counters.csv
  done
done
python - <<'PY' > gpurun_out/r03c1_sca_pmc_summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sorted(glob.glob('gpurun_out/r03c1_pmc_f*_counters.csv')):
    mode = path.split('_pmc_')[1].split('_')[0]
    for row in csv.DictReader(open(path)):
        k = mode + ' ' + row['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
        a = acc[k][row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
print("# SCA gather kernels, per-launch means (rocprofv3 --pmc, one set per pass, --kernel-trace only; bench.py --scope hotpath)")
for k, c in sorted(acc.items()):
    print(k)
    for n, (cnt, tot) in sorted(c.items()):
        print(f"    {n:36s} {tot / cnt:16.1f}   (n={cnt})")
PY
head -80 gpurun_out/r03c1_sca_pmc_summary.txt
