#!/bin/bash
# sample power / clocks (rocm-smi) while the pipelined bench runs: is the step power-bound?
cd ${GRAFT_REPO_ROOT:-/root/repo}
MODE=${1:-pipelined}
python bench.py --steps 1500 --warmup 5 --no-extras --no-cpu-baseline --mode $MODE > /tmp/pp_bench.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|fclk|busy" | tr '\n' ';'
  echo
  sleep 0.7
done
wait $BP
python -c "import json;d=json.load(open('/tmp/pp_bench.json'));print('%s: %.1f img/s %.3f ms' % ('$MODE', d['value'], d['ms_per_step']))"
