#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for sz in ${SIZES:-4,4 5,3 7,1 4,3,1 3,3,2 4,2,2 5,2,1 3,2,2,1}; do
    ns=$(echo $sz | tr ',' '\n' | wc -l)
    v=$(DAFNE_SPLIT_SIZES=$sz python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --splits $ns ${BFLAGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s (min %.1f max %.1f) %.3f ms' % (d['value'], d['value_min'], d['value_max'], d['ms_per_step']))")
    echo "rep $rep sub-batches $sz: $v"
  done
done
