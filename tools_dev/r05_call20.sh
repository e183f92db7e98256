#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( HZ_ONLY_VPROJ=1 timeout 200 python tools_dev/hazard_repro.py 30 ) > gpurun_out/r05_c20_repro.log 2>&1; grep -E "REPRO|Error|error" gpurun_out/r05_c20_repro.log | cut -c1-260
