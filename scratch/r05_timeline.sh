#!/bin/bash
# rocprofv3 kernel trace of the timed layout -> concurrency timeline + the dispatches of one steady step
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/${1:-r05n}; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras ${BENCH_FLAGS:-} > $OUT/prof.log 2>&1
db=$(find $OUT/prof -name '*results.db' | head -1)
python $ROOT/scripts/rocprof_timeline.py $db 2 > $OUT/timeline.txt
python $ROOT/scripts/rocprof_dump_step.py $db > $OUT/step_dump.txt; python $ROOT/scratch/r05_stepdump.py $db > $OUT/step_full.txt
rm -rf $OUT/prof
tail -30 $OUT/timeline.txt
