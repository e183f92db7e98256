#!/bin/bash
# last sanity of the round on the final tree: smoke() + the concurrency stress test
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3; timeout 60 python -m pytest tests/test_gpu_concurrency.py -m gpu -q 2>&1 | tail -2 ) > gpurun_out/r05_c25_last_sanity.log 2>&1
cat gpurun_out/r05_c25_last_sanity.log | cut -c1-200
