#!/bin/bash
# Runs the -m gpu suite one test file per process (a crash in one file does not hide the others) and writes a summary.
# usage: bash profiles/gputests.sh <tag>      output: gpurun_out/<tag>/gputest_<file>.txt, gpurun_out/<tag>/gputest_summary.txt
TAG=${1:-r03}; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
: > $OUT/gputest_summary.txt
for f in tests/test_gpu_*.py; do
  b=$(basename $f .py)
  timeout 1200 python -m pytest $f -m gpu -q -s > $OUT/gputest_$b.txt 2>&1
  echo "$b rc=$? $(grep -E "passed|failed|error|no tests ran" $OUT/gputest_$b.txt | tail -1)" >> $OUT/gputest_summary.txt
done
cat $OUT/gputest_summary.txt
