#!/bin/bash
# Round 5, GPU call 1: (1) the new range-scale / concurrency / ADVICE tests + the tests they touch, (2) stage bisect of the row
# pipeline's wrong rows, (3) the row pipeline's first native run + A/B, (4) the default bench line with the
# headline_feature_parity and extra.train legs.  Results -> gpurun_out/r05_c1_*.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=gpurun_out/r05_c1
( time timeout 200 python -m pytest tests/test_gpu_value_range.py tests/test_gpu_concurrency.py -m gpu -q -s ) > ${T}_newtests.log 2>&1
grep -E "amp |passed|failed|Error|error" ${T}_newtests.log | cut -c1-250 | tail -30
( time timeout 500 python -m pytest tests -m gpu -q ) > ${T}_tests.log 2>&1
tail -5 ${T}_tests.log
( time timeout 150 python tools_dev/row_pipeline_bisect.py 2 3 ) > ${T}_bisect.log 2>&1; grep -v "^$" ${T}_bisect.log | tail -14 | cut -c1-400
( time HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 timeout 150 python tools_dev/row_pipeline_bisect.py 2 3 ) > ${T}_bisect_noreclaim.log 2>&1; grep "mode=serial" ${T}_bisect_noreclaim.log | cut -c1-300
( time OCC_TEST_ROW_PIPELINE=1 timeout 200 python -m pytest tests/test_gpu_row_pipeline.py -m gpu -q -s ) > ${T}_rowpipe_test.log 2>&1; tail -8 ${T}_rowpipe_test.log | cut -c1-300
B="timeout 100 python bench.py --scope hotpath --steps 40 --warmup 6 --no-cpu-baseline --no-extras"
run() { name=$1; shift; env "$@" $B > ${T}_hot_$name.log 2>&1; echo "$name: $(grep '^{' ${T}_hot_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), "ms/step; enqueue", round(d.get("host_enqueue_ms_per_step") or 0,3), d["config"].get("encoder_row_pipeline"))' 2>/dev/null || tail -2 ${T}_hot_$name.log)"; }
run default OCC_ENCODER_ROW_PIPELINE=0
run native_k1 OCC_ENCODER_ROW_PIPELINE=1 OCC_ROW_PIPELINE_NATIVE=1
run native_k2 OCC_ENCODER_ROW_PIPELINE=2 OCC_ROW_PIPELINE_NATIVE=1
run native_k2_stagger OCC_ENCODER_ROW_PIPELINE=2 OCC_ROW_PIPELINE_NATIVE=1 OCC_ROW_PIPELINE_FLAGS=1
run default_again OCC_ENCODER_ROW_PIPELINE=0
( time timeout 400 python bench.py ) > ${T}_bench_e2e.log 2>&1; grep '^{' ${T}_bench_e2e.log | python -c '
import sys,json
d=json.loads(sys.stdin.read())
print("e2e", d["value"], d["ms_per_step"], "enq", d.get("host_enqueue_ms_per_step"))
print("headline_feature_parity", json.dumps(d.get("headline_feature_parity"))[:900])
print("extra", {k:(v.get("value"), v.get("ms_per_step"), v.get("error")) for k,v in d.get("extra",{}).items()})
print("roofline", {k:d["roofline"].get(k) for k in ("launch_ms","frac","frac_alg","traffic")})
print("cpu_baseline", d["cpu_baseline"].get("value"), d["cpu_baseline"].get("parity_max_abs_diff"))
' || tail -20 ${T}_bench_e2e.log
grep -i "warn" ${T}_bench_e2e.log | head -5
