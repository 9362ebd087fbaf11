#!/bin/bash
# timed layout with the second sub-batch's backbone held until the first reaches its head towers (DAFNE_STREAM_SKEW=1), with and
# without the tower kernel's grid capped at half the chip, alternating
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for cfg in "0 0" "1 0" "1 128" "1 160" "0 128"; do
    set -- $cfg
    v=$(DAFNE_STREAM_SKEW=$1 DAFNE_RP_GRID=$2 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s (min %.1f max %.1f) %.3f ms' % (d['value'], d['value_min'], d['value_max'], d['ms_per_step']))")
    echo "rep $rep skew=$1 rp_grid=$2: $v"
  done
done
