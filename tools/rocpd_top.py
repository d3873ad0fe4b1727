#!/usr/bin/env python
"""The longest dispatches of ONE training step (between the last two `adam_kernel` launches) from a rocprofv3 rocpd database,
with their grid and the kernel that ran before them - which launches of a 3000-launch step are worth a look.
Usage: tools/rocpd_top.py DB [--top N] [--mark KERNEL_SUBSTR]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return re.sub(r"^void ", "", name)[:64]


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 60
    mark = sys.argv[sys.argv.index("--mark") + 1] if "--mark" in sys.argv else "adam_kernel"
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    dims = [c for c in ("grid_x", "grid_y", "grid_z", "workgroup_x", "workgroup_y", "workgroup_z") if c in cols]
    rows = cur.execute("select %s, start, end%s from kernels order by start" % (name_col, "".join(", " + c for c in dims))).fetchall()
    marks = [i for i, r in enumerate(rows) if mark in r[0]]
    if len(marks) < 2:
        sys.exit("fewer than 2 '%s' dispatches" % mark)
    lo, hi = marks[-2] + 1, marks[-1] + 1
    step = rows[lo:hi]
    total = sum(r[2] - r[1] for r in step) / 1e3
    print("# one step = dispatches %d..%d (%d launches), %.1f us of kernel time, %.1f us first start .. last end; dims = %s"
          % (lo, hi, len(step), total, (step[-1][2] - step[0][1]) / 1e3, dims))
    order = sorted(range(len(step)), key=lambda i: step[i][1] - step[i][2])[:top]
    print("%5s %-64s %9s %6s  %-22s %s" % ("#", "kernel", "dur_us", "pct", "workgroups (x,y,z)", "previous kernel"))
    for i in order:
        r = step[i]
        wg = ""
        if len(dims) == 6:
            wg = "%d,%d,%d" % (r[3] // max(1, r[6]), r[4] // max(1, r[7]), r[5] // max(1, r[8]))
        print("%5d %-64s %9.1f %5.1f%%  %-22s %s" % (i, short(r[0]), (r[2] - r[1]) / 1e3, 100 * (r[2] - r[1]) / 1e3 / total, wg,
                                                  short(step[i - 1][0]) if i else ""))


if __name__ == "__main__":
    main()
