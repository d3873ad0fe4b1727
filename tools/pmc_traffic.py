#!/usr/bin/env python
"""HBM-side bytes from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/pmc_targets.py (separate passes, no tracing
domains - MI355X_MICROARCH.md HBM section).  gfx950 correction: FETCH_SIZE tallies 128-byte requests at 64 B for 16 B/lane
streaming reads -> doubled; WRITE_SIZE as reported.  Usage:
  pmc_traffic.py WINO_FETCH_DB WINO_WRITE_DB WARP_FETCH_DB WARP_WRITE_DB N_WINO_LAUNCHES N_REDUCE N_ITERS > profiles/r02_traffic.json"""
import json
import sqlite3
import sys


def last(db, counter, kernel, count):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select dispatch_id, sum(counter_value) from pmc_events where name like ? and counter_name = ? group by dispatch_id "
                       "order by dispatch_id", ("%" + kernel + "%", counter)).fetchall()
    rows = rows[-count:] if count > 0 else []
    return len(rows), sum(r[1] for r in rows) * 1024.0          # the counters are reported in KB


wf, ww, pf, pw = sys.argv[1:5]
n_wino, n_red, iters = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/pmc_targets.py {wino,warp} %d (tools/prof_traffic.sh)" % iters,
       "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B for 16 B/lane streaming reads, MI355X_MICROARCH.md); WRITE_SIZE as reported"}
cf, f1 = last(wf, "FETCH_SIZE", "conv_wino_kernel", n_wino * iters)
cr, f2 = last(wf, "FETCH_SIZE", "conv_splitk_reduce_kernel", n_red * iters)
_, w1 = last(ww, "WRITE_SIZE", "conv_wino_kernel", n_wino * iters)
_, w2 = last(ww, "WRITE_SIZE", "conv_splitk_reduce_kernel", n_red * iters)
out["wino_dispatches_counted"] = [cf, cr]
out["wino_fetch_bytes_per_step"] = round(2.0 * (f1 + f2) / iters)
out["wino_write_bytes_per_step"] = round((w1 + w2) / iters)
out["wino_bytes_per_step"] = out["wino_fetch_bytes_per_step"] + out["wino_write_bytes_per_step"]
tot_f = tot_w = 0.0
for k, per in (("warp_cl_kernel", 3), ("warp_planar", 2)):
    _, f = last(pf, "FETCH_SIZE", k, per * iters)
    _, w = last(pw, "WRITE_SIZE", k, per * iters)
    tot_f += f
    tot_w += w
out["warp_fetch_bytes_per_video"] = round(2.0 * tot_f / iters)
out["warp_write_bytes_per_video"] = round(tot_w / iters)
out["warp_bytes_per_video"] = out["warp_fetch_bytes_per_video"] + out["warp_write_bytes_per_video"]
print(json.dumps(out, indent=1))
