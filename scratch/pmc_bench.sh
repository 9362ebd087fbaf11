# HBM traffic of the conv kernels of one bench step: FETCH_SIZE / WRITE_SIZE in separate passes (MI355X_MICROARCH.md)
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc6_$c -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
done
python - <<PY
import csv, collections, os
root = os.environ["GRAFT_REPO_ROOT"]
out = []
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(root + "/gpurun_out/pmc6_%s/p_counter_collection.csv" % c)))
    acc = collections.defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"]
        if "conv" in n or "gn_" in n or "nms" in n:
            k = n.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
            acc[k].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        out.append("%-12s %-42s launches %5d  total %10.1f MB  avg/launch %8.2f MB" % (c, k[:42], len(v), sum(v) / 1024, sum(v) / len(v) / 1024))
open(root + "/gpurun_out/pmc6_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
PY
