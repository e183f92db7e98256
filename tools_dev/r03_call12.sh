#!/bin/bash
# round 3, call 12: full GPU suite with the tests' printed maxima kept (-s), after the last test edits
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -s ) > gpurun_out/r03c12_tests.log 2>&1; tail -4 gpurun_out/r03c12_tests.log | cut -c1-200
grep -h "max|\|max diff\|maxdiff\|golden\|bit-\|agreement\|relative\|worst\|vs oracle\|vs the\|mismatch" gpurun_out/r03c12_tests.log | grep -v "^tests/\|Warning\|assert" | cut -c1-200 > gpurun_out/r03c12_parity_prints.txt; wc -l gpurun_out/r03c12_parity_prints.txt
