#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for sz in 8,8 10,6 5,6,5 6,4,6 6,5,5; do
  ns=$(echo $sz | tr ',' '\n' | wc -l)
  echo "== batch 16 sub-batches $sz"
  SPLITS=$ns DAFNE_SPLIT_SIZES=$sz python scratch/fp8_ab.py 10 2>&1 | grep -E "^[12] (bf16|fp8 patch)"
done
for sz in 4,4 5,3 3,2,3; do
  ns=$(echo $sz | tr ',' '\n' | wc -l)
  v=$(DAFNE_SPLIT_SIZES=$sz python bench.py --depth 50 --steps 30 --warmup 5 --no-extras --no-cpu-baseline --splits $ns 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s (min %.1f max %.1f) %.3f ms' % (d['value'], d['value_min'], d['value_max'], d['ms_per_step']))")
  echo "R50 b8 sub-batches $sz: $v"
done
