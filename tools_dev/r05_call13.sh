#!/bin/bash
# Round 5, last check: smoke(), the whole GPU suite, the default bench line — what the driver runs at round end.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=$GRAFT_REPO_ROOT/gpurun_out/r05_c13
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > ${T}_smoke.log 2>&1; grep -v libdrm ${T}_smoke.log | tail -8
( time timeout 600 python -m pytest tests -m gpu -q ) > ${T}_tests.log 2>&1; grep -E "passed|failed|FAILED" ${T}_tests.log | tail -5
( time timeout 400 python bench.py ) > ${T}_bench_default.log 2>&1; grep '^{' ${T}_bench_default.log | python -c '
import sys,json
d=json.loads(sys.stdin.read())
print("e2e", round(d["value"],1), round(d["ms_per_step"],3), "enq", d.get("host_enqueue_ms_per_step"))
print("roofline", {k:d["roofline"].get(k) for k in ("launch_ms","frac","frac_alg","traffic","traffic_over_compulsory","tsa_launch_ms")})
print("headline_feature_parity", d["headline_feature_parity"]["max_abs_diff"])
print("extra", {k:(round(v.get("value",0),1), round(v.get("ms_per_step",0),3), v.get("error")) for k,v in d.get("extra",{}).items()})
'; grep -i "warn" ${T}_bench_default.log | head -3; grep real ${T}_bench_default.log
