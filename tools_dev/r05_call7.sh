#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/gpurun_out/r05_c7
H="timeout 150 python tools_dev/hazard_matrix.py 60"
hz() { name=$1; shift; ( env OCC_VPROJ_OVERLAP=0 HZ_LOAD=vproj "$@" $H $name ) > ${T}_hz_$name.log 2>&1; grep -E "HAZARD|Error" ${T}_hz_$name.log | cut -c1-200 | head -3; }
hz base A=1
hz f32planes_resident HZ_PLANES=f32
hz f32planes_tiled HZ_PLANES=f32 OCC_VPROJ_RESIDENT=0
hz f16_tiled OCC_VPROJ_RESIDENT=0
hz vp_noclamp_tiled OCC_VPROJ_RESIDENT=0 OCC_DBG_VP=1
hz vp_nocvt_tiled OCC_VPROJ_RESIDENT=0 OCC_DBG_VP=2
hz vp_nostore_tiled OCC_VPROJ_RESIDENT=0 OCC_DBG_VP=4
hz vp_nocvt_resident OCC_DBG_VP=2
hz vp_nostore_resident OCC_DBG_VP=4
hz tsa_exp2 OCC_DBG_TSA=8
hz tsa_rcp OCC_DBG_TSA=16
hz tsa_exp2_rcp OCC_DBG_TSA=24
hz tsa_wave_exp2_rcp OCC_TSA_TILE=0 OCC_DBG_TSA=24
