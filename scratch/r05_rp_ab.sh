#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
V=scratch/variants
python -m pytest tests/test_gpu_conv.py -q -k "resident_patch" 2>&1 | tail -1
python scratch/rp_gnab_check.py 2>&1 | tail -8 | cut -c1-150
for rep in 1 2 3; do
  for lib in $V/librp_base.so ""; do
    echo "== rp rep $rep lib=${lib:-new}"
    DAFNE_AMD_LIB=$lib python scratch/rp_micro.py 8 2>&1 | grep "^rp"
    DAFNE_AMD_LIB=$lib python scratch/rp_micro.py 4 2>&1 | grep "^rp"
  done
done
