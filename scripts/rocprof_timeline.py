#!/usr/bin/env python3
"""rocprofv3 --kernel-trace results .db of a `bench.py` run (pipelined layout) -> concurrency timeline of its steady steps (the
window between the first preprocess dispatch of the 9th-from-last and of the last step; a step = `splits` preprocess dispatches):
share of the window with 0 / 1 / 2 / 3 / >= 4 kernels resident, kernel families while they are the only kernel on the GPU.
Usage: python scripts/rocprof_timeline.py bench_results.db [splits=3]"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t]
if "kernels" in tabs:
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
else:
    t = [x for x in kd if x.startswith("rocpd_kernel_dispatch")][0]
    sym = [x for x in tabs if x.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (t, sym)).fetchall()
splits = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pre = [s for n, s, e in rows if "preprocess_kernel" in n]
starts = pre[::splits]
t0, t_end = starts[-9], starts[-1]
print("steps in the window: %s ms" % [round((b - a) / 1e6, 2) for a, b in zip(starts[-9:], starts[-8:])])
rows = [(n, max(s, t0), min(e, t_end)) for n, s, e in rows if e > t0 and s < t_end]
ev = []
for n, s, e in rows:
    ev.append((s, 1, n))
    ev.append((e, -1, n))
ev.sort()
level_time = defaultdict(float)
alone = defaultdict(float)
active = defaultdict(int)
cur = 0
last = t0
for t, d, n in ev:
    dt = t - last
    if dt > 0:
        level_time[min(cur, 4)] += dt
        if cur == 1:
            k = [a for a, v in active.items() if v > 0][0]
            alone[k] += dt
    last = t
    cur += d
    active[n] += d
wall = last - t0
print("window %.2f ms, %d dispatches" % (wall / 1e6, len(rows)))
for k in sorted(level_time):
    print("  %s kernels resident: %6.2f %%" % (("%d" % k) if k < 4 else ">=4", 100 * level_time[k] / wall))
print("kernel families running ALONE (share of the window):")
for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-70s %5.2f %%" % (k[:70], 100 * v / wall))
