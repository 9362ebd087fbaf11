#!/bin/bash
# rocprofv3 PMC passes of one bench command (run on the GPU box through gpurun):
#   FETCH_SIZE, WRITE_SIZE               -> HBM traffic per launch (MI355X_MICROARCH.md: separate passes, 2 x FETCH_SIZE)
#   SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CU_CYCLES + GRBM_GUI_ACTIVE -> matrix-pipe utilisation per kernel
# Counters are collected with --kernel-trace only (no other trace domain).  Output: gpurun_out/pmc_<tag>/summary.txt
# and gpurun_out/pmc_<tag>/pmc_traffic.json (copy both into profiles/).
TAG=${1:-v9}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras ${BENCH_FLAGS:-}"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  d=$OUT/$(echo $c | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $d -o p --output-format csv -- $CMD > $d.log 2>&1
done
python $ROOT/scripts/pmc_summary.py $OUT $TAG
