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
    print("# step span %.1f us, sum of kernel durations %.1f us, sum of gaps %.1f us" % ((prev_end - t0) / 1e3, busy, (prev_end - t0) / 1e3 - busy))
    # every step of the trace: span / sum of durations / sum of gaps (so that one step is not taken for all)
    spans = []
    for a, b in zip(marks[:-1], marks[1:]):
        seg = rows[a:b]
        span = (max(r[2] for r in seg) - seg[0][1]) / 1e3
        dur = sum(r[2] - r[1] for r in seg) / 1e3
        spans.append((span, dur, b - a))
    same = [x for x in spans if x[2] == hi - lo]
    if same:
        print("# %d steps of %d launches in this trace: span min / median / max %.1f / %.1f / %.1f us; sum of gaps min / max %.2f / %.2f us"
              % (len(same), hi - lo, min(x[0] for x in same), sorted(x[0] for x in same)[len(same) // 2], max(x[0] for x in same),
                 min(x[0] - x[1] for x in same), max(x[0] - x[1] for x in same)))
    print("# NOTE on gap_us: these are rocprofv3's own begin / end timestamps (rocpd `kernels.start`, `kernels.end`), not reconstructed ones.  In a\n"
          "# hipGraph-replayed chain of barrier-ordered dispatches they satisfy begin(i+1) == end(i) to the tick on every row (see the column): the\n"
          "# begin stamp of a dispatch is evidently taken when its predecessor completes, not when its first wave runs, so the cost of a kernel\n"
          "# boundary (the predecessor's cache write-back, dispatch, wave ramp) sits INSIDE the successor's duration and no gap can show.  What one more\n"
          "# dependent launch costs in this step is therefore measured directly: tools/probe_launch_floor_insitu.py adds empty launches to the\n"
          "# captured step and divides the change of the video time by their number (profiles/r06_e_launch_floor_insitu.txt: 1.63-1.67 us per\n"
          "# added launch, after a convolution, after a GroupNorm apply or after another empty kernel alike).")


if __name__ == "__main__":
    main()
