#!/bin/bash
# scheduling knobs re-measured on the unequal three-sub-batch layout (default), alternating; one env assignment per variant
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { v=$(env $1 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s (min %.1f max %.1f) %.3f ms' % (d['value'], d['value_min'], d['value_max'], d['ms_per_step']))"); echo "$1 $2: $v"; }
for rep in 1 2; do
  run "X=0" ""
  run "GPU_MAX_HW_QUEUES=8" ""
  run "GPU_MAX_HW_QUEUES=8 DAFNE_SPLIT_SIZES=3,2,2,1" "--splits 4"
  run "GPU_MAX_HW_QUEUES=8 DAFNE_SPLIT_SIZES=2,3,1,2" "--splits 4"
  run "DAFNE_RP_GRID=224" ""
  run "DAFNE_RP_GRID=192" ""
  run "DAFNE_RP_PAIR_SHARED=1" ""
  run "DAFNE_STREAM_GRID=224" ""
  run "X=0" "--no-defer"
done
