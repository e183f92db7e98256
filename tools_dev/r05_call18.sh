#!/bin/bash
# Round 5: the gathers without the IEEE division expansion (occ::fdiv) — does the hazard go away?  Suite + A/B.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/gpurun_out/r05_c18
( timeout 300 python tools_dev/hazard_tsa_standalone.py 100 ) > ${T}_tsa_standalone.log 2>&1; grep -E "TSA-STANDALONE|Error" ${T}_tsa_standalone.log | cut -c1-200
H="timeout 200 python tools_dev/hazard_matrix.py 150"
hz() { name=$1; shift; ( env "$@" $H $name ) > ${T}_hz_$name.log 2>&1; grep -E "HAZARD|Error" ${T}_hz_$name.log | cut -c1-200 | head -2; }
hz onestream_vproj_load HZ_LOAD=vproj
hz onestream_chainA_load HZ_LOAD=chainA
hz overlap_copy_gemm OCC_VPROJ_OVERLAP=1
hz overlap_vproj_load OCC_VPROJ_OVERLAP=1 HZ_LOAD=vproj
hz overlap_all_loads OCC_VPROJ_OVERLAP=1 HZ_LOAD=vproj,chainA,copy,gemm
hz overlap_tile_vproj_load OCC_VPROJ_OVERLAP=1 OCC_TSA_TILE=1 HZ_LOAD=vproj
( time timeout 600 python -m pytest tests -m gpu -q ) > ${T}_tests.log 2>&1; grep -E "passed|failed|FAILED" ${T}_tests.log | tail -6 | cut -c1-200
B="timeout 100 python bench.py --scope hotpath --steps 40 --warmup 6 --no-cpu-baseline --no-extras"
run() { name=$1; shift; ( env "$@" $B ) > ${T}_hot_$name.log 2>&1; echo "$name: $(grep '^{' ${T}_hot_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); m=d["mfma_kernels"]; r=d["roofline"]; print(round(d["ms_per_step"],3), "ms/step; lin", round(m["linear_ms_per_step"],3), "sca", round(r["launch_ms"],4), "tsa", round(r["tsa_launch_ms"],4))' 2>/dev/null || tail -2 ${T}_hot_$name.log)"; }
run onestream_a A=1
run overlap_a OCC_VPROJ_OVERLAP=1
run onestream_b A=1
run overlap_b OCC_VPROJ_OVERLAP=1
