#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
V=scratch/variants
python -m pytest tests/test_gpu_conv.py -q -k "bottleneck_body" 2>&1 | tail -1
for rep in 1 2 3; do
  for lib in $V/libbn_base.so ""; do
    echo "== bneck rep $rep lib=${lib:-new}"
    DAFNE_AMD_LIB=$lib python scratch/bneck_micro.py 8 2>&1 | grep -E "fused|bit-identical" | grep -v unfused
    DAFNE_AMD_LIB=$lib python scratch/bneck_micro.py 4 2>&1 | grep -E "K=4 fused"
  done
done
DAFNE_BNECK_STAMPS=1 DAFNE_AMD_LIB=$V/libbn_new_t.so python scratch/bneck_micro.py 8 2>&1 | tail -1
