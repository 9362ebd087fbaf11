#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
nf=0
for i in $(seq 1 ${N:-14}); do
  if [ -n "$KEXPR" ]; then env ${CFG:-X=0} python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -k "$KEXPR" > /tmp/out.txt 2>&1
  else env ${CFG:-X=0} python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short > /tmp/out.txt 2>&1; fi
  if grep -q " failed" /tmp/out.txt; then nf=$((nf+1)); echo "run $i FAILED"; grep -v amdgpu.ids /tmp/out.txt | grep -E "^E |^/root|Error|assert" | head -6; fi
done
tail -1 /tmp/out.txt
echo "$nf failed runs of ${N:-14}"
