#!/bin/bash
# timed layout with the second sub-batch stream's res4 blocks started N us after the first's (DAFNE_RES4_PHASE_US), alternating
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for us in ${PHASES:-0 20 35 50}; do
    v=$(DAFNE_RES4_PHASE_US=$us python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s (min %.1f max %.1f) %.3f ms' % (d['value'], d['value_min'], d['value_max'], d['ms_per_step']))")
    echo "rep $rep phase=$us us: $v"
  done
done
