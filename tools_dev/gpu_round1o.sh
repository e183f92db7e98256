#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCC|TCP|SQ|GRBM|TA|TD)_[A-Za-z0-9_]+" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt
grep -E "MFMA|TCC_HIT|TCC_MISS|TCC_REQ|TCP_TOTAL|TCP_TCC_READ|TA_BUSY|TA_TA_BUSY|GRBM_GUI|SQ_BUSY_CY|SQ_WAVE_CYCLES|SQ_INSTS_VALU_MFMA|LDS_BANK|SQ_LDS" $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt | tr '\n' ' '
echo
KR="sca_fused|tsa_fused|conv3d_mfma|occ_heads|linear_bf16x3"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "TA_TA_BUSY_sum TA_BUSY_avr"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv --kernel-include-regex "$KR" -d /tmp/pmcx_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --scope hotpath --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/pmcx_$i.log 2>&1
  f=$(find /tmp/pmcx_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp $f $GRAFT_REPO_ROOT/gpurun_out/pmcx_${i}_counters.csv; else echo "set $i ($set): no csv"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/pmcx_$i.log; fi
done
cd $GRAFT_REPO_ROOT; python tools_dev/pmc_summary.py gpurun_out/pmcx_*_counters.csv
