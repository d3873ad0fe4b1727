#!/usr/bin/env python
"""Kernel census of one LFAE stage-1 training step on the GPU (bench.py's lfae_train leg alone): launches per step, native share,
the top vendor and native kernels.  Usage: lfae_census.py [--batch 32] [--top 40]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--top", type=int, default=40)
a = ap.parse_args()
os.environ["LFDM_CENSUS_TOP"] = str(a.top)
import bench  # noqa: E402

out = bench.lfae_train_bench("cuda:0", 0, 1, a.steps, 2, a.batch)
k = out.pop("kernels", {})
print(json.dumps(out))
for key in ("launches_per_step", "native_launches", "device_us_per_step", "native_time_share", "vendor_families"):
    print(key, k.get(key))
print("# vendor")
for r in k.get("vendor_top", []):
    print("%6d %10.1f  %s" % (r["calls"], r["us"], r["kernel"]))
print("# native")
for r in k.get("native_top", []):
    print("%6d %10.1f  %s" % (r["calls"], r["us"], r["kernel"]))
