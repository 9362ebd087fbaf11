#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for sp in 1 2 3 4; do
  v=$(python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --splits $sp 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))")
  echo "splits $sp: $v"
done
v=$(python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --mode serial 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))")
echo "serial: $v"
done
