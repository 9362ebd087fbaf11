#!/bin/bash
# whole-block kernels writing Y over X (DAFNE_INPLACE_RES, default 1) against separate buffers; timed and serial layouts, alternating
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_conv.py -q -x -k "bottleneck_body or block_mid or block_narrow" 2>&1 | tail -2
for rep in 1 2 3; do
  for ip in 0 1; do
    for mode in pipelined serial; do
      v=$(DAFNE_INPLACE_RES=$ip python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --mode $mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s (min %.1f max %.1f) %.3f ms' % (d['value'], d['value_min'], d['value_max'], d['ms_per_step']))")
      echo "rep $rep inplace=$ip $mode: $v"
    done
  done
done
