#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/gpurun_out/r05_c10
H="timeout 150 python tools_dev/hazard_matrix.py 60"
hz() { name=$1; shift; ( env OCC_VPROJ_OVERLAP=0 HZ_LOAD=chainA "$@" $H $name ) > ${T}_hz_$name.log 2>&1; grep -E "HAZARD|Error" ${T}_hz_$name.log | cut -c1-150 | head -3; }
hz base A=1
hz no_logits OCC_DBG_TSA=32
hz no_offsets OCC_DBG_TSA=64
hz no_logits_offsets OCC_DBG_TSA=96
hz no_softmax OCC_DBG_TSA=128
hz all_zero_row OCC_DBG_TSA=256
hz no_l_o_s OCC_DBG_TSA=224
