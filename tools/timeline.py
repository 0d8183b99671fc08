#!/usr/bin/env python3
"""timeline.py trace.db [n_last] -- the last n rx:: kernel dispatches of a rocprofv3 --kernel-trace database in start order,
with duration and the idle gap to the previous kernel's end (us).  Finds launch bubbles the per-kernel sums hide."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = c.execute("select name, start, end from kernels where name like '%rx::%' or name like '%rocclr%' order by start").fetchall()
prev = None
for name, s, e in rows[-n:]:
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print("%-40s dur %9.1f us   gap %8.1f us" % (name.split("(")[0][-40:], (e - s) / 1e3, gap))
    prev = e
