#!/usr/bin/env python
"""Where the GPU is idle inside ONE video of the headline run, from a rocprofv3 rocpd database: the span from the first dispatch of a video (the
region encoder's first kernel after the previous video's last decode kernel) to its last, the sum of kernel durations, and the largest idle gaps
with the kernels on both sides.  A video = everything between two long host-side pauses is NOT assumed: videos are cut at the dispatches of
`--first` (default: the first `step_cond_kernel` after a non-sampler kernel).
Usage: tools/rocpd_video_gaps.py DB > profiles/xxx_video_gaps.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return re.sub(r"^void ", "", name)[:60]


db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
steps = [i for i, r in enumerate(rows) if "step_cond_kernel" in r[0]]
# a video's sampler = 100 consecutive steps; its first step_cond follows a gap in the step spacing (the decode + next video's encode in between)
from collections import Counter
launches = Counter(b - a for a, b in zip(steps[:-1], steps[1:])).most_common(1)[0][0]
starts = [b for a, b in zip(steps[:-1], steps[1:]) if b - a > launches + 20]
print("# %d dispatches, %d sampler steps of %d launches, %d videos" % (len(rows), len(steps), launches, len(starts)))
for vi in range(len(starts) - 1):
    # the video = from the first kernel after the previous video's sampler ... through this video's decode: cut at the dispatch after the
    # last sampler step's final kernel of the PREVIOUS video (approximated: this video's first step to the next video's first step)
    lo, hi = starts[vi], starts[vi + 1]
    seg = rows[lo:hi]
    span = (seg[-1][2] - seg[0][1]) / 1e3
    busy = sum(r[2] - r[1] for r in seg) / 1e3
    gaps = sorted(((seg[i + 1][1] - max(r[2] for r in seg[max(0, i - 3):i + 1])) / 1e3, i) for i in range(len(seg) - 1))
    print("video %d: %d dispatches, span %.1f us (first step .. next video's first step), kernels %.1f us, idle %.1f us" % (vi, len(seg), span, busy, span - busy))
    n_in_sampler = 0
    for g, i in reversed(gaps[-12:]):
        print("    gap %8.1f us after #%d %-50s before %-50s" % (g, i, short(seg[i][0]), short(seg[i + 1][0])))
    last_step = max(i for i, r in enumerate(seg) if "step_cond_kernel" in r[0])
    tail = seg[last_step + launches:]
    if tail:
        print("    after the sampler: %d dispatches, span %.1f us, kernels %.1f us" % (len(tail), (tail[-1][2] - tail[0][1]) / 1e3, sum(r[2] - r[1] for r in tail) / 1e3))
        agg = {}
        for r in tail:
            k = short(r[0])
            agg.setdefault(k, [0, 0.0])
            agg[k][0] += 1
            agg[k][1] += (r[2] - r[1]) / 1e3
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print("        %-60s %4d %9.1f us" % (k, v[0], v[1]))
