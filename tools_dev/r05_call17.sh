#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 0 1 2 3; do ( OCC_TSA_VARIANT=$v timeout 300 python tools_dev/hazard_tsa_standalone.py 60 ) > gpurun_out/r05_c17_tsa_standalone_v$v.log 2>&1; echo "== variant $v"; grep -E "TSA-STANDALONE|Error|error" gpurun_out/r05_c17_tsa_standalone_v$v.log | cut -c1-200; done
