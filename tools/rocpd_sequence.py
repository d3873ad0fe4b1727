#!/usr/bin/env python
"""Dispatch sequence of ONE sampler step (one hipGraph replay) from a rocprofv3 rocpd database: kernel, grid size, start
offset, duration and the idle gap before it - the view that shows what a B=1 step is made of, launch by launch.
Usage: tools/rocpd_sequence.py DB [--first KERNEL_SUBSTR] [--nth N] > profiles/xxx_step_sequence.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return re.sub(r"^void ", "", name)[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    first = sys.argv[sys.argv.index("--first") + 1] if "--first" in sys.argv else "step_cond_kernel"
    nth = int(sys.argv[sys.argv.index("--nth") + 1]) if "--nth" in sys.argv else -2
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    extra = [c for c in ("grid_x", "workgroup_x", "grid_size_x", "workgroup_size_x") if c in cols]
    rows = cur.execute("select %s, start, end%s from kernels order by start" % (name_col, "".join(", " + c for c in extra))).fetchall()
    marks = [i for i, r in enumerate(rows) if first in r[0]]
    if len(marks) < 3:
        sys.exit("fewer than 3 '%s' dispatches in the trace" % first)
    lo, hi = marks[nth], marks[nth + 1]
    t0, prev_end, busy = rows[lo][1], rows[lo][1], 0.0
    print("# dispatches %d..%d of %d (one step = %d launches), columns: %s" % (lo, hi, len(rows), hi - lo, extra))
    print("%4s %-70s %10s %9s %8s %7s" % ("#", "kernel", "wgs", "t_us", "dur_us", "gap_us"))
    for i, r in enumerate(rows[lo:hi]):
        name, s, e = r[0], r[1], r[2]
        wgs = ""
        if len(extra) >= 2 and r[4]:
            wgs = "%d" % (r[3] // max(1, r[4]))
        print("%4d %-70s %10s %9.1f %8.2f %7.2f" % (i, short(name), wgs, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
        busy += (e - s) / 1e3
        prev_end = max(prev_end, e)
    print("# step span %.1f us, sum of kernel durations %.1f us" % ((prev_end - t0) / 1e3, busy))


if __name__ == "__main__":
    main()
