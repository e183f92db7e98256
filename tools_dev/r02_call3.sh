#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools_dev/sca_probe.py 40 > gpurun_out/r02_sca_probe2.log 2>&1; grep '^[0-9]' gpurun_out/r02_sca_probe2.log | cut -c1-300
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv --kernel-include-regex "sca_" -d /tmp/pmcs_$i -o p -- python $GRAFT_REPO_ROOT/tools_dev/sca_probe.py 4 > $GRAFT_REPO_ROOT/gpurun_out/r02_pmcs2_$i.log 2>&1)
  f=$(find /tmp/pmcs_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r02_pmcs2_${i}_counters.csv
done
python - <<'PY' > gpurun_out/r02_sca_pmc_summary2.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sorted(glob.glob('gpurun_out/r02_pmcs2_*_counters.csv')):
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        a = acc[k][row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
for k, c in sorted(acc.items()):
    print(k)
    for n in sorted(c):
        print(f"    {n:32s} n={c[n][0]:3d} mean={c[n][1] / c[n][0]:.6g}")
PY
cat gpurun_out/r02_sca_pmc_summary2.txt
( timeout 900 python -m pytest tests/test_gpu_modules.py "tests/test_gpu_training.py::test_ddp_two_ranks_gradients_identical" -x -q ) > gpurun_out/r02_tests3.log 2>&1; tail -3 gpurun_out/r02_tests3.log
