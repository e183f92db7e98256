#!/bin/bash
# the whole gpu suite under every opt-in storage / kernel switch (plumbing check of the non-default paths)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in "OCC_SCA_VALUES=f16" "OCC_SCA_VALUES=f32" "OCC_SCA_HEAD_MAJOR=0" "OCC_LINEAR_CHAIN=0" "OCC_TRAIN_FUSED_CONV=0" "OCC_TRAIN_FUSED_LN=0" "OCC_MSDA_BWD_DETERMINISTIC=1"; do
  tag=$(echo $cfg | tr '=' '_')
  ( env $cfg timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r06_matrix_$tag.log 2>&1
  echo "$cfg: $(grep -E 'passed|failed' gpurun_out/r06_matrix_$tag.log | tail -1)"
  grep -E "^FAILED|^ERROR" gpurun_out/r06_matrix_$tag.log | head -8
done
