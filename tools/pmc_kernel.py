#!/usr/bin/env python3
"""print per-kernel counter sums of a rocprofv3 --pmc rocpd database: tools/pmc_kernel.py db [kernel-substring]"""
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
q = ("select kernel_name, counter_name, count(*), sum(value), avg(duration) from counters_collection "
     "group by kernel_name, counter_name order by kernel_name")
for k, cn, n, s, d in c.execute(q):
    if pat in k:
        print("%-40s %-24s calls %4d sum %18.1f avg_us %10.1f" % (k.split("(")[0][:40], cn, n, s, d / 1e3))
