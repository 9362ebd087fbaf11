#!/bin/bash
# soak: the files that flaked earlier in the round, repeated, then the whole GPU suite twice
cd ${GRAFT_REPO_ROOT:-/root/repo}
f=0
for i in $(seq 1 8); do
  for t in tests/test_gpu_model.py tests/test_inference_loop.py tests/test_gpu_reproducible.py; do
    python -m pytest $t -q -m gpu --tb=line > /tmp/o.txt 2>&1 || { f=$((f+1)); echo "run $i $t FAILED"; grep -E "^/root|Error|assert" /tmp/o.txt | head -5; }
  done
done
echo "repeated files: $f failures in 24 runs"
for i in 1 2; do python -m pytest tests -q -m gpu 2>&1 | tail -1; done
