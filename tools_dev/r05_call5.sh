#!/bin/bash
# Round 5, GPU call 5: who is the aggressor / what is the victim of the side-stream hazard.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/gpurun_out/r05_c5
H="timeout 150 python tools_dev/hazard_matrix.py 150"
hz() { name=$1; shift; ( env "$@" $H $name ) > ${T}_hz_$name.log 2>&1; grep -E "HAZARD" ${T}_hz_$name.log | cut -c1-300; }
hz default A=1
hz tiled_vproj OCC_VPROJ_RESIDENT=0
hz f32_rows OCC_SCA_VALUES=f32
hz slab_wait OCC_DBG_TSA=4
hz nooverlap_ownload_vproj OCC_VPROJ_OVERLAP=0 HZ_LOAD=vproj
hz nooverlap_ownload_range OCC_VPROJ_OVERLAP=0 HZ_LOAD=range
hz nooverlap_ownload_both OCC_VPROJ_OVERLAP=0 HZ_LOAD=vproj,range
hz overlap_noload HZ_LOAD=none
hz wave_overlap_noload OCC_TSA_TILE=0 HZ_LOAD=none
