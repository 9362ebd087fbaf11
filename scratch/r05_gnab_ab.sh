#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
V=scratch/variants
echo "== check base"; python scratch/rp_gnab_check.py 2>&1 | tail -8
echo "== check gnab"; DAFNE_AMD_LIB=$V/librp_gnab.so python scratch/rp_gnab_check.py 2>&1 | tail -8
for rep in 1 2 3; do
  for lib in "" $V/librp_gnab.so; do
    echo "== rp rep $rep lib=${lib:-base}"
    DAFNE_AMD_LIB=$lib python scratch/rp_micro.py 8 2>&1 | grep "^rp"
  done
done
for rep in 1 2; do
  for lib in "" $V/librp_gnab.so; do
    v=$(DAFNE_AMD_LIB=$lib python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s (min %.1f max %.1f) %.3f ms equal=%s' % (d['value'], d['value_min'], d['value_max'], d['ms_per_step'], d.get('timed_path_equals_immediate')))")
    echo "bench rep $rep lib=${lib:-base}: $v"
  done
done
