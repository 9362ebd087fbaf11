cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum; do
  rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmcws_$c -o p --output-format csv -- python $GRAFT_REPO_ROOT/scratch/ws_micro.py "$@" > /dev/null 2>&1
  python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/pmcws_$c/p_counter_collection.csv")))
acc = collections.defaultdict(list)
for r in rows:
    if "conv" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:70], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("$c", k, "n=%d avg=%.1f" % (len(v), sum(v) / len(v)))
PY
done
