#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/gpurun_out/r05_c6
H="timeout 150 python tools_dev/hazard_matrix.py 60"
hz() { name=$1; shift; ( env "$@" $H $name ) > ${T}_hz_$name.log 2>&1; grep -E "HAZARD|   rep|      " ${T}_hz_$name.log | cut -c1-330 | head -24; }
hz tiled_vproj OCC_VPROJ_RESIDENT=0
hz tiled_vproj_wave OCC_VPROJ_RESIDENT=0 OCC_TSA_TILE=0
hz default A=1
hz nooverlap_ownload_vproj OCC_VPROJ_OVERLAP=0 HZ_LOAD=vproj
hz nooverlap_ownload_both OCC_VPROJ_OVERLAP=0 HZ_LOAD=vproj,range,copy,gemm
