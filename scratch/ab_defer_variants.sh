cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for spec in "base|" "pair|DAFNE_RP_PAIR_SHARED=1" "layer0|DAFNE_RP_LAYER0=1" "grid224|DAFNE_RP_GRID=224" "grid216|DAFNE_RP_GRID=216"; do
    name=${spec%%|*}; envs=${spec#*|}
    v=$(env $envs python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s' % d['value'])")
    echo "$rep $name: $v"
  done
done
