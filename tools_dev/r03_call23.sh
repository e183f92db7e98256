#!/bin/bash
# round 3, call 23: EXPERIMENT pixel-pair value layout for the fp16 SCA gather (kernel-side support + probe-side permute)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for pr in 0 1 0 1; do
  echo "pair $pr" >> gpurun_out/r03c23_sca_pair.txt
  OCC_SCA_PAIR=$pr timeout 200 python tools_dev/sca_probe.py 60 2>&1 | grep "f16 values" | cut -c1-230 >> gpurun_out/r03c23_sca_pair.txt
done
for pr in 0 1; do
  (cd /tmp && OCC_SCA_PAIR=$pr timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv --kernel-include-regex "sca_fused_h" -d /tmp/pp_$pr -o p -- python $GRAFT_REPO_ROOT/tools_dev/sca_probe.py 8 > /tmp/pp_$pr.log 2>&1)
  f=$(find /tmp/pp_$pr -name "*counter_collection.csv" | head -1)
  echo "pair $pr PMC:" >> gpurun_out/r03c23_sca_pair.txt
  python - $f >> gpurun_out/r03c23_sca_pair.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for row in csv.DictReader(open(sys.argv[1])):
    a = acc[row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
for n, (c, t) in sorted(acc.items()): print(f"    {n:36s} n={c:3d} mean={t / c:16.1f}")
PY
done
cat gpurun_out/r03c23_sca_pair.txt
