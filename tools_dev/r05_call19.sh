#!/bin/bash
# Round 5, last check after the training-path kernels lost their divisions: smoke, whole suite, default bench.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/gpurun_out/r05_c19
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > ${T}_smoke.log 2>&1; grep -v libdrm ${T}_smoke.log | tail -3
( time timeout 600 python -m pytest tests -m gpu -q ) > ${T}_tests.log 2>&1; grep -E "passed|failed|FAILED" ${T}_tests.log | tail -5
( timeout 100 python -m pytest tests/test_gpu_value_range.py -m gpu -q -s -k fdiv 2>&1 | grep -E "fdiv:|passed|failed" ) | tail -2
( time timeout 400 python bench.py ) > ${T}_bench_default.log 2>&1; grep '^{' ${T}_bench_default.log | python -c '
import sys,json
d=json.loads(sys.stdin.read())
print("e2e", round(d["value"],1), round(d["ms_per_step"],3))
print("roofline", {k:d["roofline"].get(k) for k in ("launch_ms","frac","frac_alg","traffic_over_compulsory","tsa_launch_ms")}, "lin", round(d["mfma_kernels"]["linear_ms_per_step"],3))
print("headline_feature_parity", d["headline_feature_parity"]["max_abs_diff"])
print("extra", {k:(round(v.get("value",0),1), round(v.get("ms_per_step",0),3), v.get("error")) for k,v in d.get("extra",{}).items()})
'; grep -i "warn" ${T}_bench_default.log | head -3
