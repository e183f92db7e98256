cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
(cd /tmp && OCC_MSDA_BWD_DETERMINISTIC=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o r -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 6 --warmup 3 --passes 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r06_c25_trace.log 2>&1)
DB=$(find /tmp/prof_tr -name "*.db" | head -1)
python tools_dev/rocpd_summary.py $DB 40 --last-ms 250 2>&1 | grep -E "msda|total" | cut -c1-150
