#!/bin/bash
# PMC passes over scratch/rp_micro.py (tower layer on conv3x3_patch / conv3x3_rp, GN on load or not): where the waves wait.
# usage (through gpurun): bash scratch/rp_pmc.sh <tag> [script]   -> gpurun_out/<tag>/summary.txt
TAG=${1:-rp_pmc}
SCRIPT=${2:-scratch/rp_micro.py}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
i=0
while read -r c; do
  [ -z "$c" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/p$i -o p --output-format csv -- python $ROOT/$SCRIPT > $OUT/p$i.log 2>&1
  echo "pass $i: $c -> $?" >> $OUT/passes.txt
done <<'EOF'
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_CMD_FIFO_FULL
SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
EOF
python $ROOT/scratch/pmc_any.py $OUT ${KERNELS:-conv3x3_rp conv3x3_patch} > $OUT/summary.txt 2>&1
find $OUT -name '*.csv' -size +2M -delete
cat $OUT/passes.txt; head -80 $OUT/summary.txt
