#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -s -x > gpurun_out/test4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/test4.log
grep -E "^[a-z_0-9]+: max|fused|passed|failed|FAILED|Error|rc=" gpurun_out/test4.log | tail -40
for args in "--mode plan" "--mode plan_fused" "--mode plan_fused_fp16"; do
  timeout 600 python tools_dev/backbone_probe.py $args 2>&1 | grep -E "^backbone|plan vs|Error|error|Traceback" | tail -6 >> gpurun_out/backbone_probe2.log
done
cat gpurun_out/backbone_probe2.log
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench4_e2e.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench4_e2e.log
tail -3 gpurun_out/bench4_e2e.log
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-miopen-fusion > gpurun_out/bench4_e2e_nofusion.log 2>&1
tail -1 gpurun_out/bench4_e2e_nofusion.log
timeout 300 python bench.py --scope hotpath --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench4_hot.log 2>&1
tail -1 gpurun_out/bench4_hot.log
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof4 -o r4 -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof4.log 2>&1
cd $GRAFT_REPO_ROOT; python tools_dev/rocpd_summary.py $(find /tmp/prof4 -name "*.db" | head -1) 70 > gpurun_out/prof4_e2e_summary.txt 2>&1; head -45 gpurun_out/prof4_e2e_summary.txt | cut -c1-180
