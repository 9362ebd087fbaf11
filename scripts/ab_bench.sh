#!/bin/bash
# same-box A/B of library / engine variants: each line "NAME ENV..." runs bench.py --no-extras twice, alternating
# usage: scripts/ab_bench.sh "A|" "B|DAFNE_FUSE_GNFIN=0" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for spec in "$@"; do
    name=${spec%%|*}; envs=${spec#*|}
    v=$(env $envs python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))")
    echo "$rep $name: $v"
  done
done
