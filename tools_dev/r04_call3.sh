#!/bin/bash
# round 4, call 3: chain kernel with per-block k-step rotation: parity + block-count sweep
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_linear.py -m gpu -q -k "chain" -x > gpurun_out/r04_c3_tests.log 2>&1; tail -3 gpurun_out/r04_c3_tests.log
timeout 300 python tools_dev/chain_probe.py > gpurun_out/r04_c3_chain_probe.txt 2>&1; cat gpurun_out/r04_c3_chain_probe.txt
