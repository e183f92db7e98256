#!/bin/bash
# round 3, call 21: after the dead-code cleanup of sca_fused.hip / common.h (no behaviour change): SCA + TSA + MSDA tests, PMC for the traffic json
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r03_final4
( timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_msda.py tests/test_gpu_fullsize.py -m gpu -q ) > gpurun_out/${T}_tests_sca.log 2>&1; tail -2 gpurun_out/${T}_tests_sca.log | cut -c1-200
KR="sca_fused"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TA_TA_BUSY_sum TA_BUSY_avr" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv --kernel-include-regex "$KR" -d /tmp/pmc_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/${T}_pmc_$i.log 2>&1)
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${T}_pmc_${i}_counters.csv
done
python tools_dev/make_traffic_json.py "sca_fused" gpurun_out/${T}_sca_gather_traffic.json "profiles/r03_final3_pmc_sca.txt (rocprofv3 --pmc, one counter set per pass with --kernel-trace only, bench.py e2e scope, base config; gfx950 correction per MI355X_MICROARCH.md: HBM read bytes = 2 x FETCH_SIZE x 1024)" gpurun_out/${T}_pmc_*_counters.csv > /dev/null
cat gpurun_out/${T}_sca_gather_traffic.json | head -12
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_e2e.log 2>&1; grep '^{' gpurun_out/${T}_bench_e2e.log | cut -c1-160
