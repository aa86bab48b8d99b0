#!/usr/bin/env python3
"""Per-kernel summary (calls, avg/min/max us, share) from a rocprofv3 rocpd sqlite database.
usage: python tools/rocpd_stats.py gpurun_out/prof_x/trace_results.db [> profiles/xyz.txt]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, start, end from kernels").fetchall() if "name" in cols else []
agg = {}
for name, st, en in rows:
    d = (en - st) / 1000.0
    a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values()) or 1.0
print("%-100s %7s %10s %10s %10s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "share"))
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-100s %7d %10.2f %10.2f %10.2f %6.1f%%" % (name[:100], a[0], a[1] / a[0], a[2], a[3],
                                                      100 * a[1] / tot))
