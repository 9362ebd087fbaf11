#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
V=scratch/variants
for lib in "" $V/libwr_d3.so $V/libwr_d4.so $V/libwr_d6.so; do
  echo "== tests lib=${lib:-d2}"; DAFNE_AMD_LIB=$lib python -m pytest tests/test_gpu_conv.py -q -k "conv_wr" 2>&1 | tail -1
done
for rep in 1 2; do
  for lib in "" $V/libwr_d3.so $V/libwr_d4.so $V/libwr_d6.so; do
    echo "== rep $rep lib=${lib:-d2}"
    DAFNE_AMD_LIB=$lib python scratch/wr_micro.py 8 2>&1 | grep "conv_wr S=" | awk '{print $0}' | sed -e 's/conv_igemm//' | cut -c1-130
  done
done
