#!/usr/bin/env python3
"""Kernel timeline of the last N steps from a rocprofv3 rocpd sqlite database: start / end of
every boxattn kernel relative to the first kernel of the step (shows what actually overlaps).
usage: python tools/rocpd_timeline.py trace_results.db [n_steps]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = con.execute("select name, start, end from kernels order by start").fetchall()
rows = [(n, s, e) for n, s, e in rows if "boxattn" in n]
# a step starts with the forward kernel
starts = [i for i, r in enumerate(rows) if "fwd2_kernel" in r[0] or "fwd_inst_wide" in r[0]]
for si in starts[-n_steps:]:
    t0 = rows[si][1]
    nxt = [i for i in starts if i > si]
    end = nxt[0] if nxt else len(rows)
    print("step:")
    for n, s, e in rows[si:end]:
        short = n.split("::")[-1].split("(")[0][:48]
        print("  %-50s %8.1f -> %8.1f us  (%6.1f)" % (short, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
