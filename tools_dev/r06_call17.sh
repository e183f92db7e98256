#!/bin/bash
# round 6 call 17: SCA gather with fewer non-gather wave loads (scalar map shapes, shuffled slot query / divisor, 16-byte anchor
# loads) against the previous library on one box; texture-path stall counters of the gather
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r06_c17
cp tools_dev/bin/libocc_amd_new.so occnet_amd/lib/libocc_amd.so
timeout 900 python -m pytest tests/test_gpu_modules.py -m gpu -q -x -k "sca or gather or golden" 2>&1 | tail -3
for v in base new base new; do
  cp tools_dev/bin/libocc_amd_$v.so occnet_amd/lib/libocc_amd.so
  OCC_SCA_PERSIST=0 timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/${T}_hot_$v.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/${T}_hot_$v.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('$v', 'ms/step', round(d['ms_per_step'],4), 'sca launch_ms', round(d['roofline']['launch_ms'],5), 'tsa', round(d['roofline'].get('tsa_launch_ms',0),5))
else:
    print('$v FAILED'); print(open('gpurun_out/${T}_hot_$v.log').read()[-1500:])
PY
done
cp tools_dev/bin/libocc_amd_new.so occnet_amd/lib/libocc_amd.so
i=0
for set in "GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_TC_STALL_sum TD_SPI_STALL_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" "TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TD_LOAD_WAVEFRONT_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv --kernel-include-regex "sca_fused|tsa_fused" -d /tmp/pmcq_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --scope hotpath --steps 2 --warmup 1 --passes 1 --no-cpu-baseline --no-kernel-timing --no-extras > $GRAFT_REPO_ROOT/gpurun_out/${T}_pmc_$i.log 2>&1)
  f=$(find /tmp/pmcq_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${T}_pmc_${i}_counters.csv
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sorted(glob.glob('gpurun_out/r06_c17_pmc_*_counters.csv')):
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        a = acc[k][row['Counter_Name']]; a[0] += 1; a[1] += float(row['Counter_Value'])
for k, c in sorted(acc.items()):
    print(k)
    for n in sorted(c):
        print(f"    {n:40s} n={c[n][0]:3d} mean={c[n][1] / c[n][0]:.5g}")
PY
