#!/bin/bash
# round-6 closing evidence on the final code: full GPU suite, the default bench line (with its extra legs), hot path, training
# (default and deterministic backward).  The kernel trace / PMC / traffic json of r06_final2 stay valid: csrc/sca_fused.hip and the
# csrc headers are unchanged since (same source digest), the hot-path kernels were not touched.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r06_final3}
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/${T}_tests.log 2>&1; tail -3 gpurun_out/${T}_tests.log
( time timeout 900 python bench.py --steps 30 --warmup 5 ) > gpurun_out/${T}_bench_e2e.log 2>&1; grep '^{' gpurun_out/${T}_bench_e2e.log | cut -c1-200
timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench_hot.log 2>&1; grep '^{' gpurun_out/${T}_bench_hot.log | cut -c1-160
timeout 600 python bench.py --mode train --steps 6 --warmup 3 --passes 3 --no-cpu-baseline > gpurun_out/${T}_bench_train.log 2>&1; grep '^{' gpurun_out/${T}_bench_train.log | cut -c1-220
OCC_MSDA_BWD_DETERMINISTIC=1 timeout 600 python bench.py --mode train --steps 6 --warmup 3 --passes 3 --no-cpu-baseline > gpurun_out/${T}_bench_train_det.log 2>&1; grep '^{' gpurun_out/${T}_bench_train_det.log | cut -c1-220
