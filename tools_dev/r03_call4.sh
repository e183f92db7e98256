#!/bin/bash
# round 3, call 4: full GPU suite, Linear ws-vs-x3 with wave-state PMC, bench lines (default, --history 3, hi-res)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r03c4_tests.log 2>&1; tail -6 gpurun_out/r03c4_tests.log | cut -c1-200
timeout 300 python tools_dev/linear_probe.py > gpurun_out/r03c4_linear_probe.log 2>&1; grep -v amdgpu.ids gpurun_out/r03c4_linear_probe.log | cut -c1-250
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS" "TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv --kernel-include-regex "linear_ws_kernel|linear_bf16x3_kernel" -d /tmp/pmc_lin_$i -o p -- python $GRAFT_REPO_ROOT/tools_dev/linear_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r03c4_pmc_lin_$i.log 2>&1)
  f=$(find /tmp/pmc_lin_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r03c4_pmc_lin_${i}_counters.csv
done
python - <<'PY' > gpurun_out/r03c4_linear_pmc_summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sorted(glob.glob('gpurun_out/r03c4_pmc_lin_*_counters.csv')):
    for row in csv.DictReader(open(path)):
        # key: kernel + grid size (distinguishes the shapes)
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')[:44] + ' grid=' + row.get('Grid_Size', '?') + ' wg=' + row.get('Workgroup_Size', '?')
        a = acc[k][row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
print("# Linear kernels under tools_dev/linear_probe.py (M = 40 000), per-launch means; rocprofv3 --pmc one set per pass")
for k, c in sorted(acc.items()):
    m = lambda n: c[n][1] / c[n][0] if n in c and c[n][0] else float('nan')
    wc = m('SQ_WAVE_CYCLES')
    print(k)
    print(f"    launches {c['SQ_WAVE_CYCLES'][0]}  busy_cyc/32 {m('SQ_BUSY_CYCLES')/32:10.0f}  wave_cycles {wc:12.0f}  WAIT_ANY {m('SQ_WAIT_ANY')/wc:5.2f}  WAIT_INST_ANY {m('SQ_WAIT_INST_ANY')/wc:5.2f}  "
          f"ACTIVE {m('SQ_ACTIVE_INST_ANY')/wc:5.2f}  MfmaUtil {m('SQ_VALU_MFMA_BUSY_CYCLES')/(m('SQ_BUSY_CYCLES')/32*1024):5.2f}  waves {m('SQ_WAVES'):7.0f}")
    print(f"    LDS active {m('SQ_ACTIVE_INST_LDS'):12.0f} bank-conflict/idx-active {m('SQ_LDS_BANK_CONFLICT')/m('SQ_LDS_IDX_ACTIVE'):5.2f}  WAIT_INST_LDS {m('SQ_WAIT_INST_LDS'):12.0f}  VMEM active {m('SQ_ACTIVE_INST_VMEM'):12.0f}  "
          f"TA_BUSY_avr {m('TA_BUSY_avr'):10.0f}  L1 acc {m('TCP_TOTAL_CACHE_ACCESSES_sum'):12.0f}  L1->L2 rd {m('TCP_TCC_READ_REQ_sum'):12.0f}  L2 hit {m('TCC_HIT_sum')/(m('TCC_HIT_sum')+m('TCC_MISS_sum')):5.2f}")
PY
cat gpurun_out/r03c4_linear_pmc_summary.txt | cut -c1-260
( time timeout 600 python bench.py --steps 30 --warmup 5 ) > gpurun_out/r03c4_bench_e2e.log 2>&1; grep '^{' gpurun_out/r03c4_bench_e2e.log | cut -c1-220
timeout 600 python bench.py --steps 10 --warmup 3 --history 3 --no-cpu-baseline > gpurun_out/r03c4_bench_e2e_hist3.log 2>&1; grep '^{' gpurun_out/r03c4_bench_e2e_hist3.log | cut -c1-220
timeout 600 python bench.py --config configs/occ_hires_400x400x32.py --scope hotpath --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03c4_bench_hires_hot.log 2>&1; grep '^{' gpurun_out/r03c4_bench_hires_hot.log | cut -c1-220
