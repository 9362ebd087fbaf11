#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
V=scratch/variants
for shape in "2 40 72 1" "3 33 65 0" "8 128 128 1" "1 4 32 1" "5 100 100 1"; do echo "-- dbg $shape"; python scratch/mid_dbg.py $shape 2>&1 | grep -E "equal|rc"; done
python -m pytest tests/test_gpu_conv.py -q -k "block_mid" 2>&1 | tail -2
for rep in 1 2 3; do
  for lib in $V/libmid_base.so ""; do
    echo "== rep $rep lib=${lib:-new}"
    DAFNE_AMD_LIB=$lib python scratch/mid_micro.py 8 1 2>&1 | tail -1
    DAFNE_AMD_LIB=$lib python scratch/mid_micro.py 8 0 2>&1 | tail -1
    DAFNE_AMD_LIB=$lib python scratch/mid_micro.py 4 1 2>&1 | tail -1
  done
done
DAFNE_MID_STAMPS=1 DAFNE_AMD_LIB=$V/libmid_t.so python scratch/mid_micro.py 8 1 2>&1 | tail -1
