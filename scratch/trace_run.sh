#!/bin/bash
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/tr; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
python $R/scratch/trace_gaps.py /tmp/tr
