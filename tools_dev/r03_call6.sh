#!/bin/bash
# round 3, call 6: fused conv3d_2+heads+decode and the Conv3d XCD tile order: tests, A/B bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r03c6_tests.log 2>&1; tail -12 gpurun_out/r03c6_tests.log | cut -c1-200
for f in 1 0; do
  OCC_DECODER_FUSE_HEADS=$f timeout 300 python bench.py --scope hotpath --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c6_bench_hot_fuse$f.log 2>&1
  grep '^{' gpurun_out/r03c6_bench_hot_fuse$f.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fuse=$f', d['value'], d['ms_per_step'], d.get('value_no_instrumentation'), json.dumps(d.get('mfma_kernels')), d['roofline'].get('launch_ms'))"
done
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03c6_bench_e2e.log 2>&1; grep '^{' gpurun_out/r03c6_bench_e2e.log | cut -c1-260
