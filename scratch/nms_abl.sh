#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for lib in "" scratch/variants/libnms_NOATOMIC.so scratch/variants/libnms_NOFAST.so scratch/variants/libnms_NOEXACT.so; do
  for spec in "uniform 10000 8" "dense 10000 8" "skewed 27000 1"; do
    r=$(DAFNE_AMD_LIB=$lib python scratch/nms_sets.py $spec 50 2>/dev/null | grep events)
    echo "${lib:-base} | $spec | $r"
  done
done
