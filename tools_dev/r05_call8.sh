#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/gpurun_out/r05_c8
( HZ_SKIP_PC=1 timeout 200 python tools_dev/hazard_micro.py 40 ) > ${T}_micro.log 2>&1; grep -E "MICRO|Error|error" ${T}_micro.log | cut -c1-200
