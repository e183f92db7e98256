#!/bin/bash
# per-block / per-wave phase timeline of the chain kernels (stamped build, OCC_CHAIN_TRACE): gpurun -- bash tools_dev/chain_trace.sh <tag> [rows,rows]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04_c18}
CHAIN_SWEEP=0 CHAIN_FLOOR=0 CHAIN_ROUNDS=0 CHAIN_TRACE=1 CHAIN_TRACE_ROWS=${2:-16384,32768,40000} timeout 300 python tools_dev/chain_probe.py > gpurun_out/${T}_chain_trace.txt 2>&1; grep -v amdgpu.ids gpurun_out/${T}_chain_trace.txt | cut -c1-200
