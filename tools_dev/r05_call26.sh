#!/bin/bash
# the suspected pattern in isolation: lane-mask micro-victim (v_cmp -> s_and_b64 [-> s_and_saveexec_b64]) next to the value projection
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( HZ_MASK=1 timeout 40 python tools_dev/hazard_repro.py 40 ) > gpurun_out/r05_c26_mask_micro.log 2>&1; grep -E "REPRO|Error|error" gpurun_out/r05_c26_mask_micro.log | cut -c1-260
