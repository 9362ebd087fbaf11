# rocprofv3 --kernel-trace CSV of a pipelined bench run -> busy union, concurrency, per-queue gaps
import csv, sys, glob, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"][:50]) for r in rows]
ev.sort()
# steady window: last 60 % of the trace
t0 = ev[int(len(ev) * 0.4)][0]; t1 = max(e[1] for e in ev)
ev = [e for e in ev if e[0] >= t0]
wall = t1 - t0
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, q, n in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, q, n in ev)
print("window %.1f ms: GPU busy (union) %.1f %%, mean concurrency %.2f kernels" % (wall / 1e6, 100 * busy / wall, tot / wall))
byq = collections.defaultdict(list)
for e in ev: byq[e[2]].append(e)
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    pos = [g for g in gaps if g > 0]
    kt = sum(e - s for s, e, _, _ in lst)
    print("queue %s: %5d kernels, kernel time %.1f %% of wall, gaps: mean %.1f us, total %.1f %% of wall, >20us: %d" % (
        q, len(lst), 100 * kt / wall, (sum(pos) / max(len(pos), 1)) / 1e3, 100 * sum(pos) / wall, sum(1 for g in pos if g > 20000)))
# largest gaps of the busiest queue, with the kernels around them
q = max(byq, key=lambda k: len(byq[k])); lst = byq[q]
big = sorted(((lst[i + 1][0] - lst[i][1], i) for i in range(len(lst) - 1)), reverse=True)[:14]
for g, i in sorted(big, key=lambda x: x[1]):
    print("  gap %7.1f us at t=%8.2f ms after %-42s before %s" % (g / 1e3, (lst[i][1] - t0) / 1e6, lst[i][3], lst[i + 1][3]))
# idle (no kernel on any queue) intervals > 15 us
idle = []
cur_e = ev[0][1]
for s, e, qq, n in ev[1:]:
    if s > cur_e + 15000: idle.append((s - cur_e, cur_e, n))
    cur_e = max(cur_e, e)
print("idle intervals > 15 us: %d, total %.2f ms" % (len(idle), sum(i[0] for i in idle) / 1e6))
for d, at, n in idle[:25]:
    print("  idle %7.1f us at t=%8.2f ms, next kernel %s" % (d / 1e3, (at - t0) / 1e6, n))
# mini timeline around the second step boundary of the busiest queue
bnd = sorted(((lst[i + 1][0] - lst[i][1], i) for i in range(len(lst) - 1) if "preprocess" in lst[i + 1][3]), reverse=True)
bi = sorted(i for _, i in bnd[:8])[2]
c = lst[bi][1]
print("timeline around t=%.2f ms (queue %s ends its step):" % ((c - t0) / 1e6, q))
for s, e, qq, n in ev:
    if e > c - 700000 and s < c + 900000:
        print("   q%s  %9.1f -> %9.1f us  %s" % (qq, (s - c) / 1e3, (e - c) / 1e3, n.replace("(anonymous namespace)::", "").replace("void ", "")[:44]))
