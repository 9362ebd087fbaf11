#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for spec in "uniform 10000 8" "dense 10000 8" "skewed 10000 8" "uniform 27000 1" "dense 27000 1" "skewed 27000 1"; do
  set -- $spec
  rm -rf /tmp/nmsprof
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/nmsprof -o p -- python $R/scratch/nms_sets.py $1 $2 $3 20 2>/dev/null | grep "ms/call"
  db=$(find /tmp/nmsprof -name '*results.db' | head -1)
  python $R/scripts/rocprof_summary.py $db | grep -v "at::\|rocclr" | head -14 | cut -c1-60,90-140
done
