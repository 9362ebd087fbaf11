#!/usr/bin/env python3
"""rocprofv3 (--kernel-trace --stats) results .db -> plain-text per-kernel table.
Usage: python scripts/rocprof_summary.py gpurun_out/prof/bench_results.db > profiles/rNN_x.txt"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print("%-90s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for name, calls, tot, avg, pct in rows:
    print("%-90s %8d %14.1f %12.3f %7.2f" % (name[:90], calls, tot, avg, pct))
