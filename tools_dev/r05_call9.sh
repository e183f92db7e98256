#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/gpurun_out/r05_c9
H="timeout 150 python tools_dev/hazard_matrix.py 60"
hz() { name=$1; shift; ( env OCC_VPROJ_OVERLAP=0 "$@" $H $name ) > ${T}_hz_$name.log 2>&1; grep -E "HAZARD|Error" ${T}_hz_$name.log | cut -c1-200 | head -3; }
hz chainA_load HZ_LOAD=chainA
hz chainA_load_noweights HZ_LOAD=chainA OCC_CHAIN_ABLATE=1
hz chainA_load_nomfma HZ_LOAD=chainA OCC_CHAIN_ABLATE=2
hz chainA_load_nofrag HZ_LOAD=chainA OCC_CHAIN_ABLATE=4
hz chainA_load_now_nomfma HZ_LOAD=chainA OCC_CHAIN_ABLATE=3
hz gemm_only HZ_LOAD=gemm
hz copy_only HZ_LOAD=copy
hz range_only HZ_LOAD=range
