#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}/scratch/wt_old
for i in $(seq 1 ${N:-10}); do
  python -m pytest tests/test_gpu_model.py -q -m gpu -x --tb=short > /tmp/out.txt 2>&1
  if grep -q failed /tmp/out.txt; then echo "old tree run $i FAILED"; grep -v amdgpu.ids /tmp/out.txt | tail -12; else echo "old tree run $i ok"; fi
done
