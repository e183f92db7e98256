#!/bin/bash
# First GPU call of round 5 (prepared at the end of round 4): does the encoder's row pipeline pay
# once the launch cost is out of the way?  ~2 minutes of box time.  Results -> gpurun_out/r05_rowpipe_*.
#   1. opt-in parity tests: Python-driven pipeline (2, 3 bands) and the native launcher (1, 2, 3 bands) against the standard
#      chain path — csrc/encoder_bands.hip has never run on an MI355X before this call;
#   2. hot-path A/B on ONE box: default | native launcher unbanded (K = 1: launch cost only) | native K = 2 with each
#      scheduling flag | native K = 3 | Python-driven K = 2;
#   3. launch-side time of each mode (tools_dev/row_pipeline_host_probe.py measures the Python-driven modes).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=gpurun_out/r05_rowpipe
( time OCC_TEST_ROW_PIPELINE=1 timeout 240 python -m pytest tests/test_gpu_row_pipeline.py -m gpu -q -s ) > ${T}_test.log 2>&1; tail -12 ${T}_test.log
B="timeout 100 python bench.py --scope hotpath --steps 40 --warmup 6 --no-cpu-baseline --no-extras"
run() { name=$1; shift; env "$@" $B > ${T}_$name.log 2>&1; echo "$name: $(grep '^{' ${T}_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), "ms/step", d["config"].get("encoder_row_pipeline"))' 2>/dev/null || tail -2 ${T}_$name.log)"; }
run default OCC_ENCODER_ROW_PIPELINE=0
run native_k1 OCC_ENCODER_ROW_PIPELINE=1 OCC_ROW_PIPELINE_NATIVE=1
run native_k2 OCC_ENCODER_ROW_PIPELINE=2 OCC_ROW_PIPELINE_NATIVE=1
run native_k2_stagger OCC_ENCODER_ROW_PIPELINE=2 OCC_ROW_PIPELINE_NATIVE=1 OCC_ROW_PIPELINE_FLAGS=1
run native_k2_bandmajor OCC_ENCODER_ROW_PIPELINE=2 OCC_ROW_PIPELINE_NATIVE=1 OCC_ROW_PIPELINE_FLAGS=2
run native_k2_stagger_bandmajor OCC_ENCODER_ROW_PIPELINE=2 OCC_ROW_PIPELINE_NATIVE=1 OCC_ROW_PIPELINE_FLAGS=3
run native_k3_stagger OCC_ENCODER_ROW_PIPELINE=3 OCC_ROW_PIPELINE_NATIVE=1 OCC_ROW_PIPELINE_FLAGS=1
run python_k2 OCC_ENCODER_ROW_PIPELINE=2
run default_again OCC_ENCODER_ROW_PIPELINE=0
timeout 100 python bench.py --scope hotpath --steps 40 --warmup 6 --no-cpu-baseline --no-extras --step-graph > ${T}_default_graph.log 2>&1; grep '^{' ${T}_default_graph.log | cut -c1-200
OCC_ENCODER_ROW_PIPELINE=2 OCC_ROW_PIPELINE_NATIVE=1 OCC_ROW_PIPELINE_FLAGS=1 timeout 100 python bench.py --scope hotpath --steps 40 --warmup 6 --no-cpu-baseline --no-extras --step-graph > ${T}_native_k2_stagger_graph.log 2>&1; grep '^{' ${T}_native_k2_stagger_graph.log | cut -c1-200; tail -2 ${T}_native_k2_stagger_graph.log | cut -c1-200
timeout 90 python tools_dev/row_pipeline_host_probe.py > ${T}_host.log 2>&1; tail -5 ${T}_host.log
