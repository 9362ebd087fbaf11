#!/bin/bash
# One GPU-box call (through gpurun) that produces the round's measurement evidence under gpurun_out/<tag>/:
#   bench.json                 default bench.py line (pipelined layout, all extras)
#   kernel_stats_isolated.txt  rocprofv3 --kernel-trace --stats of `bench.py --mode serial` (whole batch, one stream)
#   kernel_stats_pipelined.txt the same for the default (timed) layout
#   pmc_isolated/              PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) of the serial layout
# usage: scripts/profile_round.sh <tag> [nopmc]
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp; export TMPDIR=/tmp
for mode in serial pipelined; do
  rm -rf $OUT/prof_$mode
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$mode -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --mode $mode > $OUT/prof_$mode.log 2>&1
  db=$(find $OUT/prof_$mode -name '*results.db' | head -1)
  name=isolated; [ $mode = pipelined ] && name=pipelined
  [ -n "$db" ] && python $ROOT/scripts/rocprof_summary.py $db > $OUT/kernel_stats_$name.txt
  rm -rf $OUT/prof_$mode
done
if [ "$2" != "nopmc" ]; then
  BENCH_FLAGS="--mode serial" PMC_ROUND=${PMC_ROUND:-r03} bash $ROOT/scripts/pmc_passes.sh ${TAG}_isolated > $OUT/pmc_isolated.log 2>&1
  cp $ROOT/gpurun_out/pmc_${TAG}_isolated/summary.txt $OUT/pmc_isolated_summary.txt 2>/dev/null
  cp $ROOT/gpurun_out/pmc_${TAG}_isolated/pmc_traffic.json $OUT/pmc_traffic_isolated.json 2>/dev/null
  rm -rf $ROOT/gpurun_out/pmc_${TAG}_isolated/FETCH_SIZE $ROOT/gpurun_out/pmc_${TAG}_isolated/WRITE_SIZE $ROOT/gpurun_out/pmc_${TAG}_isolated/SQ_VALU_MFMA_BUSY_CYCLES
fi
tail -c 400 $OUT/bench.json
