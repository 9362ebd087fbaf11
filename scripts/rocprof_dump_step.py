#!/usr/bin/env python3
"""rocprofv3 --kernel-trace results .db of a pipelined `bench.py` run -> the dispatches of one steady step as text
(start offset us, duration us, queue, kernel), from 300 us before the decode of the 6th-from-last step to 700 us after.
Usage: python scripts/rocprof_dump_step.py bench_results.db"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")).fetchall()
dec = [s for n, s, e, q in rows if "decode_collect" in n]
t0 = dec[-6] - 300000
t1 = dec[-6] + 700000
for n, s, e, q in rows:
    if e > t0 and s < t1:
        short = n.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        print("%9.1f %8.1f  q%-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, short))
