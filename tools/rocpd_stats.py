#!/usr/bin/env python
"""Per-kernel summary (count, total, avg, min, max, share) from a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db on ROCm 7.2).
Usage: tools/rocpd_stats.py DB [--top N] > profiles/xxx_kernel_stats.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for name, s, e in rows:
        d = (e - s) / 1e3
        a = agg.setdefault(short(name), [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    t0 = min(r[1] for r in rows)
    t1 = max(r[2] for r in rows)
    print("# kernels: %d dispatches, %.3f ms total kernel time, %.3f ms first-start..last-end" % (len(rows), total / 1e3, (t1 - t0) / 1e6))
    print("%-112s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-112s %8d %12.1f %10.2f %10.2f %10.2f %6.2f%%" % (name, a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / total))


if __name__ == "__main__":
    main()
