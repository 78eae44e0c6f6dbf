#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 results database: tools/kstats.py <results.db> [n]"""
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = c.execute(f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0, sum(d.end-d.start)/1000.0 "
                 f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 5 desc").fetchall()
tot = sum(r[4] for r in rows)
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("%-70s n=%6d avg %9.1f min %8.1f us %5.1f%%" % (r[0][:70], r[1], r[2], r[3], 100 * r[4] / tot))
