#!/bin/bash
# round 2, call 13: block-aggregated binning passes of the deformable-attention backward, SCA training path with the
# query Linears applied before the rebatch, heads' weight gradient on linear_wgrad, faster partial reduction.
# Every rocprofv3 / long command under `timeout -k 5` (call 12 lost 25 minutes to a profiler that ignored SIGTERM).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout -k 5 600 python -m pytest tests/test_gpu_linear.py tests/test_gpu_backward.py tests/test_gpu_training.py tests/test_gpu_backbone.py tests/test_gpu_decoder.py tests/test_gpu_modules.py tests/test_gpu_fullsize.py -m gpu -q -k "not (base_geometry or hires or images_to_voxels)" ) > gpurun_out/r02c13_tests.log 2>&1; tail -8 gpurun_out/r02c13_tests.log | cut -c1-200
run_train() { # name, env...
  name=$1; shift
  env "$@" timeout -k 5 200 python bench.py --mode train --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r02c13_train_$name.log 2>&1
  echo "$name: $(grep '^{' gpurun_out/r02c13_train_$name.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])' 2>&1 | tail -1)"
}
run_train all_on A=1
run_train wave_bins OCC_MSDA_BWD_BIN=wave
run_train ref_rebatch OCC_SCA_TRAIN_REBATCH=reference
run_train no_wgrad_only OCC_TRAIN_WGRAD_ONLY=0
(cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o r -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/r02c13_trace.log 2>&1)
DB=$(find /tmp/prof_tr -name "*.db" | head -1)
timeout -k 5 120 python tools_dev/rocpd_summary.py $DB 400 --last-ms 400 > gpurun_out/r02c13_train_trace_summary.txt 2>&1; head -16 gpurun_out/r02c13_train_trace_summary.txt | cut -c1-170
timeout -k 5 90 tools_dev/bin/ta_probe 3 > gpurun_out/r02c13_ta_probe_3waves.txt 2>&1; cut -c1-230 gpurun_out/r02c13_ta_probe_3waves.txt
timeout -k 5 90 tools_dev/bin/ta_probe 8 > gpurun_out/r02c13_ta_probe_8waves.txt 2>&1; tail -12 gpurun_out/r02c13_ta_probe_8waves.txt | cut -c1-230
