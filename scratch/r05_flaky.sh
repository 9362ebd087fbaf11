#!/bin/bash
# flaky-test bisect: the whole model test file, repeated, under engine toggles
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in ${CFGS:-"X=0" "DAFNE_PIPELINE_SPLITS=2" "DAFNE_INPLACE_RES=0" "DAFNE_B1_SETS=2"}; do
  for i in 1 2 3 4; do
    r=$(env $cfg python -m pytest tests/test_gpu_model.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed" | tr '\n' ' ')
    echo "$cfg run $i: $r"
  done
done
