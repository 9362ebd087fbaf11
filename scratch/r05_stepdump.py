#!/usr/bin/env python3
"""kernel-trace .db of a pipelined bench run -> every dispatch of one steady step (between two preprocess pairs), per queue:
start offset us, duration, gap to the previous dispatch on that queue; then per-queue busy / idle totals."""
import sqlite3, sys
from collections import defaultdict
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = c.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
pre = [s for n, s, e, q in rows if "preprocess_kernel" in n]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
t0, t1 = pre[-2 * k], pre[-2 * k + 2]
print("step window %.2f ms" % ((t1 - t0) / 1e6))
last = {}
busy = defaultdict(float); gaps = defaultdict(float)
for n, s, e, q in rows:
    if s < t0 or s >= t1: continue
    short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    g = (s - last[q]) / 1e3 if q in last else 0.0
    print("%8.1f %7.1f gap %6.1f q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, g, q, short))
    last[q] = e; busy[q] += (e - s) / 1e3
    if g > 0: gaps[q] += g
for q in busy: print("queue %s busy %.0f us, gaps %.0f us" % (q, busy[q], gaps[q]))
