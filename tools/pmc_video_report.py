#!/usr/bin/env python
"""Whole-step counters of a B = 1 sampler step and of the decode's warp launches, from the four rocprofv3 --pmc passes of
tools/prof_step_pmc.sh over tools/pmc_video.py (FETCH_SIZE; WRITE_SIZE; SQ set; TCC set - separate passes, no tracing domains).
  pmc_video_report.py FETCH_DB WRITE_DB SQ_DB TCC_DB OUT_JSON > table.txt
The last sampler step = the dispatches after the second-to-last sampler_update_kernel up to and including the last one.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE x2 for 16 B/lane streaming reads; WRITE_SIZE as reported; both in KB."""
import json
import re
import sqlite3
import sys


SIMDS = 1024.0          # 256 CUs x 4 SIMDs


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def dispatches(db):
    """[(dispatch_id, kernel, {counter: value}, duration_ns)] in submission order."""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select dispatch_id, name, counter_name, sum(counter_value), min(duration), count(*) from pmc_events "
                       "group by dispatch_id, counter_name order by dispatch_id").fetchall()
    out, idx = [], {}
    for did, name, cname, val, dur, inst in rows:
        if did not in idx:
            idx[did] = len(out)
            out.append((did, short(name), {}, dur))
        # GRBM_GUI_ACTIVE: one value per XCD instance, each the same clock count -> mean; everything else: sum over instances
        out[idx[did]][2][cname] = val / inst if cname == "GRBM_GUI_ACTIVE" else val
    return out


def last_step(ds):
    marks = [i for i, d in enumerate(ds) if d[1].startswith("sampler_update_kernel")]
    if len(marks) < 2:
        sys.exit("fewer than 2 sampler_update_kernel dispatches")
    return ds[marks[-2] + 1: marks[-1] + 1], ds[marks[-1] + 1:]


def family(kernel, prev_family):
    if kernel.startswith("conv_wino_kernel"):
        return "winograd"
    if kernel.startswith("conv_splitk_reduce_kernel"):
        return prev_family if prev_family in ("winograd", "direct") else "direct"
    if re.match(r"conv_(igemm|ksw|pw|smalln)", kernel):
        return "direct"
    if "attn" in kernel or "attention" in kernel:
        return "attention"
    if kernel.startswith("gn_") or "norm" in kernel:
        return "norm"
    return "other"


def main():
    fdb, wdb, sdb, tdb, out_json = sys.argv[1:6]
    passes = {"fetch": dispatches(fdb), "write": dispatches(wdb), "sq": dispatches(sdb), "tcc": dispatches(tdb)}
    steps = {k: last_step(v) for k, v in passes.items()}
    n = len(steps["fetch"][0])
    for k, (st, _) in steps.items():
        if len(st) != n or [d[1] for d in st] != [d[1] for d in steps["fetch"][0]]:
            sys.exit("pass %s saw a different dispatch sequence (%d vs %d launches)" % (k, len(st), n))
    fams, per_kernel, prev = {}, {}, None
    tot = {"fetch_bytes": 0.0, "write_bytes": 0.0, "mfma_busy_cycles": 0.0, "busy_cu_cycles": 0.0, "gui_active_cycles": 0.0, "dur_us_serialised": 0.0}
    for i in range(n):
        kern = steps["fetch"][0][i][1]
        fam = family(kern, prev)
        prev = fam
        f = 2.0 * 1024.0 * steps["fetch"][0][i][2].get("FETCH_SIZE", 0.0)
        w = 1024.0 * steps["write"][0][i][2].get("WRITE_SIZE", 0.0)
        sq = steps["sq"][0][i][2]
        tc = steps["tcc"][0][i][2]
        dur = steps["sq"][0][i][3] / 1e3
        for key, grp in ((fam, fams), (re.sub(r"<.*$", "", kern), per_kernel)):
            g = grp.setdefault(key, {"launches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0, "mfma_busy_cycles": 0.0, "busy_cu_cycles": 0.0,
                                     "wave_cycles": 0.0, "wait_any": 0.0, "tcc_hit": 0.0, "tcc_req": 0.0, "gui_active_cycles": 0.0, "dur_us_serialised": 0.0})
            g["launches"] += 1
            g["fetch_bytes"] += f
            g["write_bytes"] += w
            g["mfma_busy_cycles"] += sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            g["busy_cu_cycles"] += sq.get("SQ_BUSY_CU_CYCLES", 0.0)
            g["wave_cycles"] += sq.get("SQ_WAVE_CYCLES", 0.0)
            g["wait_any"] += sq.get("SQ_WAIT_ANY", 0.0)
            g["tcc_hit"] += tc.get("TCC_HIT_sum", 0.0)
            g["tcc_req"] += tc.get("TCC_REQ_sum", 0.0)
            g["gui_active_cycles"] += tc.get("GRBM_GUI_ACTIVE", 0.0)
            g["dur_us_serialised"] += dur
        tot["fetch_bytes"] += f
        tot["write_bytes"] += w
        tot["mfma_busy_cycles"] += sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        tot["busy_cu_cycles"] += sq.get("SQ_BUSY_CU_CYCLES", 0.0)
        tot["gui_active_cycles"] += tc.get("GRBM_GUI_ACTIVE", 0.0)
        tot["dur_us_serialised"] += dur
    # the decode's warp launches (after the last sampler step)
    warp = {"launches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0, "dur_us_serialised": 0.0, "sq": {}, "tcc": {}, "list": []}
    tails = {k: [d for d in v[1] if d[1].startswith("warp_")] for k, v in steps.items()}
    for i, d in enumerate(tails["fetch"]):
        warp["launches"] += 1
        warp["fetch_bytes"] += 2.0 * 1024.0 * d[2].get("FETCH_SIZE", 0.0)
        warp["write_bytes"] += 1024.0 * tails["write"][i][2].get("WRITE_SIZE", 0.0)
        warp["dur_us_serialised"] += tails["sq"][i][3] / 1e3
        warp["list"].append({"kernel": d[1][:60], "fetch_bytes": round(2.0 * 1024.0 * d[2].get("FETCH_SIZE", 0.0)),
                             "write_bytes": round(1024.0 * tails["write"][i][2].get("WRITE_SIZE", 0.0)),
                             "dur_us_serialised": round(tails["sq"][i][3] / 1e3, 1)})
        for k2, v2 in tails["sq"][i][2].items():
            warp["sq"][k2] = warp["sq"].get(k2, 0.0) + v2
        for k2, v2 in tails["tcc"][i][2].items():
            warp["tcc"][k2] = warp["tcc"].get(k2, 0.0) + v2
    res = {"source": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_* | TCC_* + GRBM_GUI_ACTIVE, one pass each) over tools/pmc_video.py "
                     "(tools/prof_step_pmc.sh); last of 3 eager sampler steps, B = 1, 40 frames",
           "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B for 16 B/lane streaming reads, MI355X_MICROARCH.md); WRITE_SIZE as reported.  "
                         "Upper bound: isolated 64-byte segments are counted exactly (profiles/r03_x_fetch_calib.txt) - halve a family's fetch_bytes for the lower bound",
           "launches_per_step": n,
           "step_fetch_bytes": round(tot["fetch_bytes"]), "step_write_bytes": round(tot["write_bytes"]),
           "step_bytes": round(tot["fetch_bytes"] + tot["write_bytes"]),
           "step_mfma_busy_cycles": round(tot["mfma_busy_cycles"]), "step_busy_cu_cycles": round(tot["busy_cu_cycles"]),
           "step_gui_active_cycles": round(tot["gui_active_cycles"]),
           "step_mfma_util": round(tot["mfma_busy_cycles"] / max(1.0, SIMDS * tot["gui_active_cycles"]), 4),
           "mfma_util_definition": "SQ_VALU_MFMA_BUSY_CYCLES (summed over the SQ instances) / (1024 SIMDs x GRBM_GUI_ACTIVE cycles of the same dispatches)",
           "families": {k: {kk: (round(vv) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in fams.items()},
           "wino_bytes_per_step": round(fams.get("winograd", {}).get("fetch_bytes", 0) + fams.get("winograd", {}).get("write_bytes", 0)),
           "direct_bytes_per_step": round(fams.get("direct", {}).get("fetch_bytes", 0) + fams.get("direct", {}).get("write_bytes", 0)),
           "warp_launches": warp["launches"], "warp_fetch_bytes_per_video": round(warp["fetch_bytes"]),
           "warp_write_bytes_per_video": round(warp["write_bytes"]), "warp_bytes_per_video": round(warp["fetch_bytes"] + warp["write_bytes"]),
           "warp_sq": {k: round(v) for k, v in warp["sq"].items()}, "warp_tcc": {k: round(v) for k, v in warp["tcc"].items()},
           "warp_dur_us_serialised": round(warp["dur_us_serialised"], 1), "warp_launch_list": warp["list"]}
    try:        # the build these counters were taken on (bench.py prints it beside its own live fingerprint)
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from cvpr23_lfdm_amd._build import source_fingerprint
        res["build"] = source_fingerprint()
    except Exception:
        res["build"] = None
    with open(out_json, "w") as f:
        json.dump(res, f, indent=1)
    print("# last sampler step: %d launches; HBM-side %.1f MB fetched (x2-corrected) + %.1f MB written; matrix-pipe utilisation (MFMA-busy cycles / "
          "(1024 SIMDs x GUI-active cycles)) = %.3f" % (n, tot["fetch_bytes"] / 1e6, tot["write_bytes"] / 1e6, res["step_mfma_util"]))
    print("%-34s %5s %10s %10s %9s %9s %8s %9s" % ("kernel", "n", "fetch MB", "write MB", "mfma util", "wait/wave", "L2 hit", "us (ser.)"))
    for grp in (fams, per_kernel):
        for k, g in sorted(grp.items(), key=lambda kv: -kv[1]["dur_us_serialised"]):
            print("%-34s %5d %10.2f %10.2f %9.3f %9.3f %8.3f %9.1f" % (
                k[:34], g["launches"], g["fetch_bytes"] / 1e6, g["write_bytes"] / 1e6, g["mfma_busy_cycles"] / max(1.0, SIMDS * g["gui_active_cycles"]),
                g["wait_any"] / max(1.0, g["wave_cycles"]), g["tcc_hit"] / max(1.0, g["tcc_req"]), g["dur_us_serialised"]))
        print()
    print("# warp launches of the decode: %d launches, %.1f MB fetched + %.1f MB written, %.1f us serialised"
          % (warp["launches"], warp["fetch_bytes"] / 1e6, warp["write_bytes"] / 1e6, warp["dur_us_serialised"]))
    for w in warp["list"]:
        print("  %-60s %8.1f MB fetched %8.1f MB written %7.1f us" % (w["kernel"], w["fetch_bytes"] / 1e6, w["write_bytes"] / 1e6, w["dur_us_serialised"]))
    for k2, v2 in sorted(warp["sq"].items()):
        print("  %-28s %16.0f" % (k2, v2))
    for k2, v2 in sorted(warp["tcc"].items()):
        print("  %-28s %16.0f" % (k2, v2))


if __name__ == "__main__":
    main()
