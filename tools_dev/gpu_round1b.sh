#!/bin/bash
# second GPU contact: new kernels' parity tests, backbone settings probe, bench + kernel trace
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/test2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/test2.log
grep -E "max|mismatch|rows hip|agreement|passed|failed|FAILED|Error|rc=" gpurun_out/test2.log | tail -70
for fm in FAST NORMAL; do
  for args in "--mode autocast --benchmark 0" "--mode bf16 --benchmark 0" "--mode bf16 --benchmark 1" "--mode fp16 --benchmark 1" "--mode bf16 --benchmark 1 --nhwc 0"; do
    MIOPEN_FIND_MODE=$fm timeout 600 python tools_dev/backbone_probe.py $args 2>&1 | grep -E "^backbone|Error|error" | tail -2 >> gpurun_out/backbone_probe.log
  done
done
cat gpurun_out/backbone_probe.log
timeout 300 python bench.py --scope hotpath --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench2_hot.log 2>&1
tail -1 gpurun_out/bench2_hot.log
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -o r2 -- python $GRAFT_REPO_ROOT/bench.py --scope hotpath --steps 8 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof2.log 2>&1
cd $GRAFT_REPO_ROOT; python tools_dev/rocpd_summary.py $(find gpurun_out/prof2 -name "*.db" | head -1) > gpurun_out/prof2_summary.txt 2>&1; head -40 gpurun_out/prof2_summary.txt
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "sca_fused|tsa_fused|conv3d_mfma|occ_heads" -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --scope hotpath --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT; ls -laR gpurun_out/pmc_* | head -30
