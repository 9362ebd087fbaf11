#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
nf=0
for i in $(seq 1 ${N:-24}); do
  python -m pytest scratch/test_flaky_diag2.py -q -m gpu --tb=short -p no:cacheprovider > /tmp/out.txt 2>&1
  if grep -q " failed" /tmp/out.txt; then nf=$((nf+1)); echo "run $i FAILED"; grep -vE "amdgpu" /tmp/out.txt | grep -E "^E " | head -10; fi
done
tail -1 /tmp/out.txt; echo "$nf failed of ${N:-24}"
