#!/bin/bash
# sample socket power / shader clock (rocm-smi) WHILE the bench runs: is the step power-bound?
# usage: scratch/power_probe.sh [pipelined|serial] [steps]     (6000 steps ~ 36 s; samples start after the start-up)
cd ${GRAFT_REPO_ROOT:-/root/repo}
MODE=${1:-pipelined}
STEPS=${2:-6000}
echo "cap: $(rocm-smi --showmaxpower 2>/dev/null | grep -o 'Power (W): [0-9.]*')"
python bench.py --steps $STEPS --warmup 5 --no-extras --no-cpu-baseline --mode $MODE > /tmp/pp_bench.json 2>/dev/null &
BP=$!
sleep 9
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk" | sed -e 's/.*Power (W): /W /' -e 's/.*sclk clock level: [0-9]*: (/sclk /' -e 's/)//' | tr '\n' ' '
  echo
  sleep 1
done
python -c "import json;d=json.load(open('/tmp/pp_bench.json'));print('$MODE: %.1f img/s %.3f ms per step over %d steps' % (d['value'], d['ms_per_step'], d['steps']))"
