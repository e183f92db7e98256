cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "TA_TA_BUSY_sum TA_BUSY_avr" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv --kernel-include-regex "conv1x1|conv3x3|bottleneck64|stem_conv" -d /tmp/pmcb_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/pmcb_$i.log 2>&1)
  f=$(find /tmp/pmcb_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/pmcb_${i}_counters.csv
done
python - > gpurun_out/pmc_backbone_derived.txt <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sorted(glob.glob('gpurun_out/pmcb_*_counters.csv')):
    for row in csv.DictReader(open(path)):
        k = (row['Kernel_Name'].split('(')[0].replace('void ', ''), row['Grid_Size'])
        a = acc[k][row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
print(f"{'kernel':40s} {'grid':>9s} {'n':>3s} {'dur_kcyc':>8s} {'Mfma':>6s} {'TA':>6s} {'L1hit':>6s} {'L2hit':>6s} {'L2missMB':>8s} {'LDScf':>6s} {'waitAny':>7s} {'waitInst':>8s}")
for k, c in sorted(acc.items()):
    m = lambda n: c[n][1] / c[n][0] if n in c and c[n][0] else float('nan')
    if 'SQ_BUSY_CYCLES' not in c or not m('SQ_BUSY_CYCLES'):
        continue
    dur = m('SQ_BUSY_CYCLES') / 32
    print(f"{k[0][:40]:40s} {k[1]:>9s} {c['SQ_BUSY_CYCLES'][0]:3d} {dur/1e3:8.1f} {m('SQ_VALU_MFMA_BUSY_CYCLES')/(dur*1024):6.3f} {m('TA_BUSY_avr')/dur:6.3f} "
          f"{1-m('TCP_TCC_READ_REQ_sum')/m('TCP_TOTAL_CACHE_ACCESSES_sum'):6.3f} {m('TCC_HIT_sum')/(m('TCC_HIT_sum')+m('TCC_MISS_sum')):6.3f} {m('TCC_MISS_sum')*128/1e6:8.1f} "
          f"{m('SQ_LDS_BANK_CONFLICT')/m('SQ_LDS_IDX_ACTIVE'):6.3f} {m('SQ_WAIT_ANY')/m('SQ_WAVE_CYCLES'):7.3f} {m('SQ_WAIT_INST_ANY')/m('SQ_WAVE_CYCLES'):8.3f}")
PY

cat gpurun_out/pmc_backbone_derived.txt
