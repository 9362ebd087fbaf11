#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
nf=0
for i in $(seq 1 ${N:-12}); do
  python -m pytest tests/test_inference_loop.py -q -x -m gpu --tb=short > /tmp/out.txt 2>&1
  if grep -q " failed" /tmp/out.txt; then nf=$((nf+1)); echo "run $i FAILED"; grep -vE "amdgpu" /tmp/out.txt | grep -E "^E |test_inference_loop.py:[0-9]+: in|^tests.*Error" | head -14; fi
done
tail -1 /tmp/out.txt; echo "$nf failed of ${N:-12}"
