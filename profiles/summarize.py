#!/usr/bin/env python3
"""Turn rocprofv3 (rocpd sqlite) outputs into the text summaries committed under profiles/.
usage: summarize.py kernel_trace.db [pmc_fetch.db] [pmc_write.db] > profiles/rNN_summary.txt
HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE are KiB, collected in separate --pmc passes;
on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x (both raw and x2 are printed)."""
import sqlite3
import sys


def stats(db):
    c = sqlite3.connect(db)
    print("== kernel trace: %s" % db)
    print("%-58s %6s %14s %12s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "%"))
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-58s %6d %14.1f %12.1f %7.2f" % (name.split("(")[0][:58], calls, total / 1e3, avg / 1e3, pct))
    row = c.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x "
                    "from kernels where name like '%k_polar%' limit 1").fetchone()
    if row:
        print("k_polar dispatch: vgpr=%s agpr=%s sgpr=%s lds=%s scratch=%s grid=%s wg=%s" % row[1:])


def pmc(db):
    c = sqlite3.connect(db)
    print("== counters: %s" % db)
    q = ("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection "
         "group by kernel_name, counter_name order by sum(value) desc")
    print("%-44s %-12s %6s %16s %16s %12s" % ("kernel", "counter", "calls", "sum_KiB", "avg_KiB/launch", "avg_us"))
    for k, cn, n, s, a, d in c.execute(q):
        print("%-44s %-12s %6d %16.1f %16.1f %12.1f" % (k.split("(")[0][:44], cn, n, s, a, d / 1e3))


if __name__ == "__main__":
    stats(sys.argv[1])
    for p in sys.argv[2:]:
        pmc(p)
