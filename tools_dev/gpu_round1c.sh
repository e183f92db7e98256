#!/bin/bash
# third GPU contact: full gpu test suite, e2e bench (default MIOpen find), kernel trace + PMC traffic
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/test3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/test3.log
grep -E "heads|passed|failed|FAILED|Error|rc=" gpurun_out/test3.log | tail -30
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench3_e2e.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench3_e2e.log
tail -2 gpurun_out/bench3_e2e.log
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof3 -o r3 -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof3.log 2>&1
cd $GRAFT_REPO_ROOT; python tools_dev/rocpd_summary.py $(find /tmp/prof3 -name "*.db" | head -1) 60 > gpurun_out/prof3_e2e_summary.txt 2>&1; head -30 gpurun_out/prof3_e2e_summary.txt
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv --kernel-include-regex "sca_fused|tsa_fused|conv3d_mfma|occ_heads" -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --scope hotpath --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
find /tmp/pmc_$c -name "*counter_collection.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/pmc_${c}_counters.csv \;
done
cd $GRAFT_REPO_ROOT; ls -la gpurun_out; head -3 gpurun_out/pmc_FETCH_SIZE_counters.csv
python tools_dev/pmc_summary.py gpurun_out/pmc_FETCH_SIZE_counters.csv gpurun_out/pmc_WRITE_SIZE_counters.csv
