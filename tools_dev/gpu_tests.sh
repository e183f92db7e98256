#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -s > gpurun_out/test.log 2>&1; echo "pytest rc=$?" >> gpurun_out/test.log
grep -E "max\||mismatch|visible|rows hip|agreement|passed|failed|FAILED|Error|rc=" gpurun_out/test.log | tail -60
