#!/bin/bash
# same-call A/B of the deferred post-process in bench.py (alternating runs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for f in "" "--no-defer"; do
    v=$(python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))")
    echo "$rep ${f:-defer}: $v"
  done
done
