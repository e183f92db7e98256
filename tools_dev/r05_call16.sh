#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/gpurun_out/r05_c16
H="timeout 150 python tools_dev/hazard_matrix.py 100"
hz() { name=$1; shift; ( env HZ_LOAD=vproj "$@" $H $name ) > ${T}_hz_$name.log 2>&1; grep -E "HAZARD|Error" ${T}_hz_$name.log | cut -c1-170 | head -3; }
hz base A=1
hz sleep_before_tsa HZ_SLEEP_BEFORE=tsa_fused_forward
hz sleep_before_tsa_long HZ_SLEEP_BEFORE=tsa_fused_forward HZ_SLEEP_CYCLES=1000000
hz sleep_before_tsa_sca HZ_SLEEP_BEFORE=tsa_fused_forward,sca_fused_forward
hz sleep_before_chains HZ_SLEEP_BEFORE=linear_ln_chain,encoder_ffn_chain,linear_pair_chain
