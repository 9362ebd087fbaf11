#!/bin/bash
# Memory-system PMC passes (L2 hit rate, fabric request latency / stalls, TLB, wave wait states) over the serial layout of bench.py:
# which kernels wait on what.  usage (through gpurun): bash scratch/mem_pmc.sh <tag>   -> gpurun_out/<tag>/summary.txt
TAG=${1:-mem_pmc}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD=${CMD:-"python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --mode serial"}
i=0
while read -r c; do
  [ -z "$c" ] && continue
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $OUT/p$i -o p --output-format csv -- $CMD > $OUT/p$i.log 2>&1
  echo "pass $i: $c -> $?" >> $OUT/passes.txt
done <<'EOF'
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum
TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_CYCLE_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum
EOF
python $ROOT/scratch/pmc_any.py $OUT ${KERNELS:-conv stem} > $OUT/summary.txt 2>&1
find $OUT -name '*.csv' -size +4M -delete
cat $OUT/passes.txt
