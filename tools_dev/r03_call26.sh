#!/bin/bash
# round 3, call 26: stall counters of the SCA gather AFTER the pixel-pair layout (same counters as call 22), probe inputs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
i=0
for grp in "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TD_TC_STALL_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum" "TCC_BUSY_avr TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv --kernel-include-regex "sca_fused_h" -d /tmp/v_$i -o p -- python $GRAFT_REPO_ROOT/tools_dev/sca_probe.py 8 > /tmp/v_$i.log 2>&1)
  f=$(find /tmp/v_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r03c26_v_$i.csv
done
python - > gpurun_out/r03c26_sca_stalls_pair.txt <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for path in sorted(glob.glob('gpurun_out/r03c26_v_*.csv')):
    for row in csv.DictReader(open(path)):
        a = acc[row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
for n, (c, t) in sorted(acc.items()): print(f"    {n:40s} n={c:4d} mean={t / c:18.1f}")
PY
cat gpurun_out/r03c26_sca_stalls_pair.txt
