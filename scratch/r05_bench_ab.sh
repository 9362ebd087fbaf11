#!/bin/bash
# whole-step A/B: round 4's kernel library against the tree's (same host code), alternating; timed layout and serial layout
cd ${GRAFT_REPO_ROOT:-/root/repo}
V=scratch/variants
for rep in 1 2 3; do
  for lib in $V/libr04.so ""; do
    for mode in pipelined serial; do
      v=$(DAFNE_AMD_LIB=$lib python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --mode $mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s (min %.1f max %.1f) %.3f ms' % (d['value'], d['value_min'], d['value_max'], d['ms_per_step']))")
      echo "rep $rep lib=${lib:-r05} $mode: $v"
    done
  done
done
