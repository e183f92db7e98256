#!/bin/bash
# round 3, call 8: PMC of the activation-resident value projection (vproj_probe, fp16 + fp32 out)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_linear.py -m gpu -q -x -k "value_proj" ) > gpurun_out/r03c8_tests.log 2>&1; tail -3 gpurun_out/r03c8_tests.log | cut -c1-200
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES" "TA_TA_BUSY_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE" "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv --kernel-include-regex "value_proj" -d /tmp/pmc_$i -o p -- python $GRAFT_REPO_ROOT/tools_dev/vproj_probe.py 10 > $GRAFT_REPO_ROOT/gpurun_out/r03c8_pmc_$i.log 2>&1)
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r03c8_pmc_${i}_counters.csv
done
python - > gpurun_out/r03c8_vproj_pmc.txt <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sorted(glob.glob('gpurun_out/r03c8_pmc_[0-9]*_counters.csv')):
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        a = acc[k][row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
for k, c in sorted(acc.items()):
    print(k)
    for n, (cnt, tot) in sorted(c.items()):
        print(f"    {n:32s} n={cnt:4d} mean={tot / cnt:16.1f}")
PY
cat gpurun_out/r03c8_vproj_pmc.txt
