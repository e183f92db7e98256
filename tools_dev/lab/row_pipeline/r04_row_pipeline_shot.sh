#!/bin/bash
# ONE short GPU call at the end of round 4 (6.6 GPU-minutes were left): the row pipeline's opt-in parity test and an
# A/B of the hot path with / without it.  Results -> gpurun_out/r04_rowpipe_*.  (At the time of this call the same-kernel
# serialisation events were ON by default: "on2" / "on3" / "e2e_on2" and the failing test are WITH them, "on2_noserial" without;
# the default has since been flipped, OCC_ROW_PIPELINE_SERIAL=1 turns them on.)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=gpurun_out/r04_rowpipe
( time OCC_TEST_ROW_PIPELINE=1 timeout 150 python -m pytest tests/test_gpu_row_pipeline.py -m gpu -x -q -s -k "2" ) > ${T}_test.log 2>&1; tail -6 ${T}_test.log
B="timeout 100 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline --no-extras"
OCC_ENCODER_ROW_PIPELINE=2 $B > ${T}_on2.log 2>&1; grep '^{' ${T}_on2.log | cut -c1-230
$B > ${T}_off.log 2>&1; grep '^{' ${T}_off.log | cut -c1-230
OCC_ENCODER_ROW_PIPELINE=2 OCC_ROW_PIPELINE_SERIAL=0 $B > ${T}_on2_noserial.log 2>&1; grep '^{' ${T}_on2_noserial.log | cut -c1-230
OCC_ENCODER_ROW_PIPELINE=3 $B > ${T}_on3.log 2>&1; grep '^{' ${T}_on3.log | cut -c1-230
OCC_ENCODER_ROW_PIPELINE=2 timeout 100 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > ${T}_e2e_on2.log 2>&1; grep '^{' ${T}_e2e_on2.log | cut -c1-230
