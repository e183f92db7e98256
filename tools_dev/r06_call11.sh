#!/bin/bash
# round 6 call 11: point_sampling with VGPR predicates (mask parity), smoke(), 2 ranks sharing the GPU through bench.py's torchrun path
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r06_c11
( timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_concurrency.py tests/test_gpu_hazard_repro.py tests/test_gpu_fullsize.py -m gpu -q ) > gpurun_out/${T}_tests.log 2>&1; tail -3 gpurun_out/${T}_tests.log | cut -c1-200
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/${T}_smoke.log 2>&1; tail -3 gpurun_out/${T}_smoke.log
( OCC_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-extras ) > gpurun_out/${T}_bench_2ranks.log 2>&1; grep '^{' gpurun_out/${T}_bench_2ranks.log | cut -c1-600 || tail -20 gpurun_out/${T}_bench_2ranks.log
grep -q '^{' gpurun_out/${T}_bench_2ranks.log || tail -30 gpurun_out/${T}_bench_2ranks.log | cut -c1-300
