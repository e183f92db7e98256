#!/bin/bash
# lab A/B: shipped library against the mask-free-set-up variant (tools_dev/lab/maskfree_setup): output hashes, step time,
# per-kernel averages from a kernel trace of each
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/r05_c23_maskfree_ab.log; : > $L
for t in product maskfree product maskfree; do timeout 60 python tools_dev/lab/maskfree_setup/ab.py $t 30 2 2>&1 | grep -E "^AB|Error|error" >> $L; done
for t in product maskfree; do
  (cd /tmp && timeout 80 rocprofv3 --kernel-trace --stats -d /tmp/prof_$t -o r -- python $GRAFT_REPO_ROOT/tools_dev/lab/maskfree_setup/ab.py $t 20 1 > /dev/null 2>&1)
  DB=$(find /tmp/prof_$t -name "*.db" | head -1)
  echo "== kernel trace, $t" >> $L; python tools_dev/rocpd_summary.py $DB 14 2>&1 | cut -c1-150 >> $L
done
cat $L | cut -c1-230
