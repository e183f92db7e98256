#!/bin/bash
# first GPU contact: parity tests, first bench line, kernel trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|Compute Unit|gfx" | head -8 > gpurun_out/rocminfo.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/test1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/test1.log
tail -5 gpurun_out/test1.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench1.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench1.log
tail -3 gpurun_out/bench1.log
timeout 300 python bench.py --scope hotpath --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench1_hot.log 2>&1
tail -2 gpurun_out/bench1_hot.log
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --scope hotpath --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof1 | head -20
