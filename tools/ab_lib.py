#!/usr/bin/env python
"""A/B of two builds of liblfdm_hip.so (or of two environments) on ONE box: the headline - ms per video in bench.py's own timed region -
alternating base / new for N rounds, so that box-to-box and run-to-run drift cancels.
    tools/ab_lib.py TAG [--base scratch/liblfdm_hip_r6base.so] [--rounds 3] [--new-env K=V ...] [--base-env K=V ...]
base leg: LFDM_HIP_LIB = the base library (+ --base-env); new leg: the in-tree library (+ --new-env).  Writes gpurun_out/TAG/ab.txt."""
import argparse
import json
import os
import subprocess
import sys

R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--base", default="scratch/liblfdm_hip_r6base.so")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--new-env", nargs="*", default=[])
ap.add_argument("--base-env", nargs="*", default=[])
a = ap.parse_args()
out_dir = os.path.join(R, "gpurun_out", a.tag)
os.makedirs(out_dir, exist_ok=True)
cmd = [sys.executable, os.path.join(R, "bench.py"), "--gpus", "1", "--steps", str(a.steps), "--warmup", "3", "--no-cpu-baseline", "--no-roofline",
       "--no-gpu-eager-baseline", "--train-steps", "0", "--lfae-train-steps", "0", "--blocks", "1"]


def leg(extra):
    env = dict(os.environ)
    env.update(dict(kv.split("=", 1) for kv in extra))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=R)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return None, r.stderr[-400:]
    return json.loads(lines[-1])["ms_per_step"], ""


base_env = (["LFDM_HIP_LIB=" + os.path.join(R, a.base)] if a.base != "none" else []) + a.base_env
rows, log = [], ["# A/B on one box: ms per video (%d timed videos after 3 warm-up).  base: %s   new: in-tree library %s" %
                 (a.steps, " ".join(base_env), " ".join(a.new_env))]
for i in range(a.rounds):
    b, eb = leg(base_env)
    n, en = leg(a.new_env)
    rows.append((b, n))
    log.append("round %d: base %s  new %s %s%s" % (i + 1, b, n, eb, en))
    print(log[-1], flush=True)
ok = [(b, n) for b, n in rows if b and n]
if ok:
    mb, mn = sum(b for b, _ in ok) / len(ok), sum(n for _, n in ok) / len(ok)
    log.append("mean: base %.2f ms  new %.2f ms  -> %+.2f ms per video (%+.2f %%)" % (mb, mn, mn - mb, 100 * (mn - mb) / mb))
    print(log[-1])
open(os.path.join(out_dir, "ab.txt"), "w").write("\n".join(log) + "\n")
