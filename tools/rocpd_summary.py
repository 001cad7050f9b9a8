#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd database (…_results.db) into the small CSV summaries kept under profiles/:
   kernel statistics (name, calls, total/avg/min/max ns, %) and, when counters were collected, per-kernel counter sums.
usage: python tools/rocpd_summary.py <results.db> <out_prefix>"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
out = sys.argv[2]
cur = db.cursor()
rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
with open(out + "_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0].split("(")[0], r[1], r[2], round(r[3], 1), round(100.0 * r[2] / tot, 2), r[4], r[5]])
        print("%-28s calls %4d  avg %12.1f ns  %5.2f%%" % (r[0].split("(")[0], r[1], r[3], 100.0 * r[2] / tot))
try:
    cols = [c[1] for c in cur.execute("pragma table_info('counters_collection')")]
    n = cur.execute("select count(*) from counters_collection").fetchone()[0]
except sqlite3.Error:
    n = 0
if n:
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    q = "select %s, counter_name, count(*), sum(value) from counters_collection group by %s, counter_name order by sum(value) desc" % (namecol, namecol)
    with open(out + "_counters.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "Dispatches", "Sum", "PerDispatch"])
        for r in cur.execute(q).fetchall():
            w.writerow([r[0].split("(")[0], r[1], r[2], r[3], round(r[3] / r[2], 1)])
            print("%-28s %-12s dispatches %4d  per dispatch %16.1f" % (r[0].split("(")[0], r[1], r[2], r[3] / r[2]))
