#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python tools_dev/hazard_gather_micro.py 60 ) > gpurun_out/r05_c15_gather_micro.log 2>&1; grep -E "GATHER-MICRO|STORE-MICRO|Error|error" gpurun_out/r05_c15_gather_micro.log | cut -c1-200
