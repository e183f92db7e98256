#!/bin/bash
# Round 5, hazard: synthetic aggressors — which ingredient of an MFMA kernel disturbs the gathers?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/gpurun_out/r05_c14
H="timeout 150 python tools_dev/hazard_matrix.py 60"
hz() { name=$1; shift; ( env "$@" $H $name ) > ${T}_hz_$name.log 2>&1; grep -E "HAZARD|Error" ${T}_hz_$name.log | cut -c1-170 | head -3; }
hz spin_mfma_only HZ_LOAD=spin0
hz spin_mfma_lds HZ_LOAD=spin1
hz spin_lds_only HZ_LOAD=spin2
hz spin_pkfma_only HZ_LOAD=spin3
hz chainA HZ_LOAD=chainA
