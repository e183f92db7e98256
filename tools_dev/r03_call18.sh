#!/bin/bash
# round 3, call 18: ablation of the fp16-row SCA gather (temporary instrumentation, results wrong by construction)
# bits: 1 every corner offset out of range (load instructions issued, no memory request), 2 no fma on the rows,
#       4 no gather at all, 8 no per-camera bilinear setup, 16 no softmax
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for a in 0 1 2 3 4 12 28 16 8 0; do
  echo "ablate $a" >> gpurun_out/r03c18_sca_ablation.txt
  OCC_SCA_ABLATE=$a timeout 200 python tools_dev/sca_probe.py 40 2>&1 | grep "f16 values" | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('   median_ms %.4f min_ms %.4f' % (d['median_ms'], d['min_ms']))" >> gpurun_out/r03c18_sca_ablation.txt
done
cat gpurun_out/r03c18_sca_ablation.txt
