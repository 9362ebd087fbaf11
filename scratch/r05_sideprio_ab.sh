#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for p in 0 -1; do
    v=$(DAFNE_SIDE_PRIO=$p python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s (min %.1f max %.1f) %.3f ms' % (d['value'], d['value_min'], d['value_max'], d['ms_per_step']))")
    echo "rep $rep side stream priority $p: $v"
  done
done
