#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for sw in raster polar; do
  echo "== sweep $sw"
  OCC_BEV_SWEEP=$sw timeout 600 python tools_dev/sca_probe.py 40 > gpurun_out/r02_sca_probe3_$sw.log 2>&1; grep '^[0-9]' gpurun_out/r02_sca_probe3_$sw.log | cut -c1-260
done
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && OCC_BEV_SWEEP=polar SCA_PROBE_VARIANTS=0,1 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv --kernel-include-regex "sca_" -d /tmp/pmcs_$i -o p -- python $GRAFT_REPO_ROOT/tools_dev/sca_probe.py 4 > $GRAFT_REPO_ROOT/gpurun_out/r02_pmcs3_$i.log 2>&1)
  f=$(find /tmp/pmcs_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r02_pmcs3_${i}_counters.csv
done
python - <<'PY' > gpurun_out/r02_sca_pmc_summary3_polar.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sorted(glob.glob('gpurun_out/r02_pmcs3_*_counters.csv')):
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        a = acc[k][row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
for k, c in sorted(acc.items()):
    print(k)
    for n in sorted(c):
        print(f"    {n:32s} n={c[n][0]:3d} mean={c[n][1] / c[n][0]:.6g}")
PY
cat gpurun_out/r02_sca_pmc_summary3_polar.txt
