"""Per-kernel means of whatever counters the rocprofv3 --pmc passes under <dir>/*/ collected.  usage: pmc_any.py <dir> [kernel substring ...]"""
import collections, csv, glob, os, sys
root = sys.argv[1]
subs = sys.argv[2:] or ["conv3x3_rp"]
per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in sorted(per):
    if not any(s in k for s in subs):
        continue
    print("%s   (%d dispatches, mean %.1f us)" % (k, len(dur[k]), sum(dur[k]) / len(dur[k]) / 1e3))
    for c in sorted(per[k]):
        v = per[k][c]
        print("    %-34s %14.4g   (n %d, min %.4g max %.4g)" % (c, sum(v) / len(v), len(v), min(v), max(v)))
