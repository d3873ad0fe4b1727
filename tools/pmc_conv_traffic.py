#!/usr/bin/env python
"""HBM-side bytes of the convolution launches of one UNet step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
over tools/unet_step.py N.  Usage: pmc_conv_traffic.py FETCH_DB WRITE_DB N  -> JSON (profiles/*_traffic.json)"""
import json
import sqlite3
import sys

KERNELS = ("conv_wino_kernel", "conv_ksw_kernel", "conv_igemm_kernel", "conv_splitk_reduce_kernel")


def total(db, counter):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for k in KERNELS:
        rows = cur.execute("select dispatch_id, sum(counter_value) from pmc_events where name like ? and counter_name = ? "
                           "group by dispatch_id", ("%" + k + "%", counter)).fetchall()
        out[k] = (len(rows), sum(r[1] for r in rows))
    return out


fetch, write, n = total(sys.argv[1], "FETCH_SIZE"), total(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
kb = 1024.0            # FETCH_SIZE / WRITE_SIZE are reported in KB
f = sum(v[1] for v in fetch.values()) * kb * 2.0 / n        # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
w = sum(v[1] for v in write.values()) * kb / n
print(json.dumps({
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/unet_step.py %d (tools/prof_traffic.sh)" % n,
    "correction": "FETCH_SIZE doubled (MI355X_MICROARCH.md HBM section: gfx950 tallies 128-B requests at 64 B for 16 B/lane streaming reads); WRITE_SIZE as reported (uncalibrated)",
    "launches_per_step": {k: v[0] // n for k, v in fetch.items()},
    "conv_fetch_bytes_per_step": round(f), "conv_write_bytes_per_step": round(w), "conv_bytes_per_step": round(f + w)}, indent=1))
