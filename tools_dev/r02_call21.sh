#!/bin/bash
# round 2, call 21 (last GPU seconds): bench.py --streams 2 vs 1 (samples dealt to two HIP streams)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for n in 1 2; do
  timeout -k 5 60 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-kernel-timing --streams $n > gpurun_out/r02c21_streams_$n.log 2>&1; grep '^{' gpurun_out/r02c21_streams_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('streams=$n', d['value'], d['ms_per_step'])"; tail -2 gpurun_out/r02c21_streams_$n.log | cut -c1-200
done
