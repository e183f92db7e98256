#!/bin/bash
# lab: the gather-related GPU tests against the mask-free-set-up variant library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 150 python tools_dev/lab/maskfree_setup/pytest_variant.py tests/test_gpu_msda.py tests/test_gpu_concurrency.py tests/test_gpu_value_range.py tests/test_gpu_fullsize.py -m gpu -q -x ) > gpurun_out/r05_c24_maskfree_tests.log 2>&1
tail -8 gpurun_out/r05_c24_maskfree_tests.log | cut -c1-200
