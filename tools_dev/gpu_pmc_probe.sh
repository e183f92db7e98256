#!/bin/bash
# PMC counters for the kernels matching $1 while running the python command in the remaining args.
# usage: gpurun -- 'bash tools_dev/gpu_pmc_probe.sh "bottleneck64" tools_dev/bottleneck_probe.py'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
KR="$1"; shift
rm -f gpurun_out/pmcp_*_counters.csv
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "TA_TA_BUSY_sum TA_BUSY_avr" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_WAVES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv --kernel-include-regex "$KR" -d /tmp/pmcp_$i -o p -- python $GRAFT_REPO_ROOT/"$@" > $GRAFT_REPO_ROOT/gpurun_out/pmcp_$i.log 2>&1)
  f=$(find /tmp/pmcp_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/pmcp_${i}_counters.csv
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sorted(glob.glob('gpurun_out/pmcp_*_counters.csv')):
    for row in csv.DictReader(open(path)):
        k = (row['Kernel_Name'].split('(')[0].replace('void ', ''), row['Grid_Size'])
        a = acc[k][row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
for k, c in sorted(acc.items()):
    print(k)
    for n in sorted(c):
        print(f"    {n:32s} n={c[n][0]:3d} mean={c[n][1] / c[n][0]:.4g}")
PY
