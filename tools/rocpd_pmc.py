#!/usr/bin/env python
"""Sum of each PMC counter per dispatch for kernels matching a substring (rocprofv3 --pmc rocpd DB).
Usage: tools/rocpd_pmc.py DB substring"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
cur = db.cursor()
rows = cur.execute("select dispatch_id, counter_name, sum(counter_value), count(*), min(duration) from pmc_events "
                   "where name like ? group by dispatch_id, counter_name", ("%" + pat + "%",)).fetchall()
by = {}
for did, cname, val, n, dur in rows:
    by.setdefault(cname, []).append((val, n, dur))
print("%-28s %16s %8s %10s" % ("counter (mean per dispatch)", "sum", "inst", "dur_us"))
for cname, vals in sorted(by.items()):
    v = sum(x[0] for x in vals) / len(vals)
    print("%-28s %16.0f %8d %10.1f   (%d dispatches)" % (cname, v, vals[0][1], sum(x[2] for x in vals) / len(vals) / 1e3, len(vals)))
