#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_backbone.py -m gpu -q -s > gpurun_out/test11.log 2>&1; echo "pytest rc=$?" >> gpurun_out/test11.log
grep -E "hip_tail|conv1x1|passed|failed|FAILED|Error|error|rc=" gpurun_out/test11.log | tail -20
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --backbone-plan folded > gpurun_out/bench11_folded.log 2>&1
tail -1 gpurun_out/bench11_folded.log | cut -c1-250
