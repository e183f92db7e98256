#!/bin/bash
# round 3, call 22: wide PMC sweep (TA / TCP / TCC / SQ stall counters) of the SCA gather and the Linear kernel on their probes,
# to name the stall reasons for the next round.  One small counter group per pass, --kernel-trace only.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rocprofv3 --list-avail > gpurun_out/r03c22_list_avail.txt 2>&1
WISH="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_COALESCED_READ_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_COALESCABLE_WAVEFRONT_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_avr TCC_TAG_STALL_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_BUBBLE_sum SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
AVAIL=""
for c in $WISH; do if grep -qw "$c" gpurun_out/r03c22_list_avail.txt; then AVAIL="$AVAIL $c"; fi; done
echo "available: $AVAIL" > gpurun_out/r03c22_pmc_wide.txt
set -- $AVAIL
i=0
while [ $# -gt 0 ]; do
  grp="$1 $2 $3"; shift; shift 2>/dev/null; shift 2>/dev/null
  i=$((i+1))
  for probe in sca linear; do
    (cd /tmp && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv --kernel-include-regex "sca_fused_h|linear_bf16x3" -d /tmp/w_${probe}_$i -o p -- python $GRAFT_REPO_ROOT/tools_dev/${probe}_probe.py 8 > /tmp/w_${probe}_$i.log 2>&1)
    f=$(find /tmp/w_${probe}_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r03c22_w_${probe}_${i}.csv
  done
done
python - >> gpurun_out/r03c22_pmc_wide.txt <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sorted(glob.glob('gpurun_out/r03c22_w_*.csv')):
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        a = acc[k][row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
for k, c in sorted(acc.items()):
    print(k)
    for n, (cnt, tot) in sorted(c.items()):
        print(f"    {n:40s} n={cnt:4d} mean={tot / cnt:18.1f}")
PY
cat gpurun_out/r03c22_pmc_wide.txt | cut -c1-200
