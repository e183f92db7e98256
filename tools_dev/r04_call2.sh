#!/bin/bash
# round 4, call 2: chain kernel block-count sweep + PMC of the chain kernels (probe) + gather_stats test
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools_dev/chain_probe.py > gpurun_out/r04_c2_chain_probe.txt 2>&1; cat gpurun_out/r04_c2_chain_probe.txt
timeout 300 python -m pytest tests/test_gpu_modules.py tests/test_gpu_linear.py -m gpu -q -k "gather_stats or chain" > gpurun_out/r04_c2_tests.log 2>&1; tail -3 gpurun_out/r04_c2_tests.log
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && CHAIN_SWEEP=0 CHAIN_ROUNDS=1 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv --kernel-include-regex "linear_chain_x3|linear_bf16x3" -d /tmp/pmc_$i -o p -- python $GRAFT_REPO_ROOT/tools_dev/chain_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r04_c2_pmc_$i.log 2>&1)
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r04_c2_pmc_${i}_counters.csv
done
python - > gpurun_out/r04_c2_pmc_derived.txt <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sorted(glob.glob('gpurun_out/r04_c2_pmc_[0-9]*_counters.csv')):
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        a = acc[k][row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
for k, c in sorted(acc.items()):
    print(k)
    for n, (cnt, tot) in sorted(c.items()):
        print(f"   {n:36s} n={cnt:4d} mean={tot / cnt:16.1f}")
PY
cat gpurun_out/r04_c2_pmc_derived.txt | head -120
