#!/bin/bash
# round 5: s_setprio alternation between the two waves of a SIMD (conv_bneck phase A, conv3x3_rp), same-box A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
V=scratch/variants
for rep in 1 2; do
  for lib in "" $V/libbn_prio1.so $V/libbn_prio4.so $V/libbn_prio16.so; do
    echo "== bneck rep $rep lib=${lib:-base}"
    DAFNE_AMD_LIB=$lib python scratch/bneck_micro.py 8 2>&1 | grep -E "fused|bit-identical" | grep -v unfused
  done
done
for lib in $V/libbn_base_t.so $V/libbn_prio4t.so; do
  echo "== stamps lib=$lib"
  DAFNE_BNECK_STAMPS=1 DAFNE_AMD_LIB=$lib python scratch/bneck_micro.py 8 2>&1 | tail -2
done
for rep in 1 2; do
  for lib in "" $V/librp_prio1.so $V/librp_prio4.so; do
    echo "== rp rep $rep lib=${lib:-base}"
    DAFNE_AMD_LIB=$lib python scratch/rp_micro.py 8 2>&1 | grep "^rp"
  done
done
