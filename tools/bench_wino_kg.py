#!/usr/bin/env python
"""Winograd launches of a B = 1 sampler step at the levels where a workgroup is alone on its CU (16x16, 8x8, 4x4), timed the way they
run in the step: every launch reads its OWN filter pack (a ring of packs larger than the 256 MB Infinity Cache, so the filters arrive
cold), split-K launches include their reduce pass, and the GroupNorm partial sums are requested like `Unet3D._conv` does.  A hipGraph of
the whole ring is replayed; the figure is us per convolution.  Sweeps the K-group count G (LFDM_WINO_KG), split-K and the column tile.
  python tools/bench_wino_kg.py [--quick]
GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr23_lfdm_amd import ops  # noqa: E402

FRAMES = 40
# (name, cin, cout, s, launches per step)
SHAPES = [
    ("64->128 @16", 64, 128, 16, 1), ("128->128 @16", 128, 128, 16, 5), ("256->128 @16", 256, 128, 16, 1),
    ("128->256 @8", 128, 256, 8, 1), ("256->256 @8", 256, 256, 8, 5), ("512->256 @8", 512, 256, 8, 1),
    ("256->512 @4", 256, 512, 4, 1), ("512->512 @4", 512, 512, 4, 9), ("1024->512 @4", 1024, 512, 4, 1),
    ("64->64 @32", 64, 64, 32, 9),
]
# (label, LFDM_WINO_KG, ksplit (0 = the plan's), LFDM_WINO_BN)
CONFIGS = [("r3 plan", "0", 0, "32"), ("auto", "auto", 0, "32")] + \
          [("G%d k%d" % (g, k), str(g), k, "32") for g in (2, 3) for k in (1, 2, 3, 4, 6)] + \
          [("G1 k%d" % k, "0", k, "32") for k in (1, 2, 3, 4, 6, 8)] + [("G1 k1 bn64", "0", 1, "64"), ("G1 k2 bn64", "0", 2, "64"), ("G2 k1 bn64", "2", 1, "64"), ("G2 k2 bn64", "2", 2, "64")]


def graph_us(fns, replays=20):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (replays * len(fns))


def main():
    dev = "cuda"
    quick = "--quick" in sys.argv
    print("%-22s %-12s %8s %6s %7s %6s" % ("shape", "config", "us/conv", "ksplit", "wgs", "TF/s"))
    best_sum, base_sum = 0.0, 0.0
    for name, cin, cout, s, count in SHAPES:
        m = FRAMES * s * s
        gf = 2.0 * m * cout * cin * 9 / 1e9
        x = torch.randn(m, cin, device=dev)
        bias = torch.randn(cout, device=dev)
        out = torch.empty(m, cout, device=dev)
        wbytes = 16 * cin * cout * 4
        ring = max(4, min(24, int(300e6 // wbytes) + 1))
        packs = [ops.pack_wino_weight(torch.randn(cout, cin, 3, 3, device=dev) * 0.05) for _ in range(ring)]
        wd = ops.pack_conv_weight(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
        rows = []
        for label, kg, ks, bn in CONFIGS:
            if quick and label not in ("r3 plan", "auto", "G3 k1", "G3 k2", "G3 k3", "G2 k1", "G2 k2", "G2 k3"):
                continue
            if ks > cin // 16:
                continue
            os.environ["LFDM_WINO_BN"] = bn
            if kg is None:
                os.environ.pop("LFDM_WINO_KG", None)
            else:
                os.environ["LFDM_WINO_KG"] = kg
            fns, keep = [], []
            nwg = 0
            for ww in packs:
                pp, _ = ops.conv_params(x, wd, cout, 3, 3, FRAMES, s, s, bias=bias, out=out, ksplit=ks, weight_wino=ww)
                tile_rows, ksplit = ops.conv_plan(pp)
                if ksplit > 1:
                    part = torch.empty(ops.conv_partial_floats(pp), device=dev)
                    pp.partial = part.data_ptr()
                    keep.append(part)
                pixels = m
                if pixels % tile_rows == 0 and (cout // 8) % 4 == 0 and ((ksplit == 1 and 32 % (cout // 8) == 0) or
                                                                       (ksplit > 1 and 256 % (pp.coutp // 4) == 0 and pp.coutp == cout)):
                    gnp = torch.empty(pixels // tile_rows, 16, device=dev)
                    pp.gn_partial, pp.gn_groups, pp.gn_pixels = gnp.data_ptr(), 8, pixels
                    keep.append(gnp)
                keep.append(pp)
                fns.append(lambda pp=pp: ops.conv_launch(pp))
                nwg = ((FRAMES * (s // 2) ** 2 + 31) // 32) * (pp.coutp // int(bn)) * ksplit
            us = graph_us(fns)
            rows.append((label, us, ksplit, nwg))
            print("%-22s %-12s %8.2f %6d %7d %6.1f" % (name, label, us, ksplit, nwg, gf / us * 1e3), flush=True)
        base = [r for r in rows if r[0] == "r3 plan"][0][1]
        best = min(rows, key=lambda r: r[1])
        auto = [r for r in rows if r[0] == "auto"][0][1]
        print("   -> %-20s best %-12s %.2f us (r3 plan %.2f, auto %.2f)" % (name, best[0], best[1], base, auto), flush=True)
        best_sum += best[1] * count
        base_sum += base * count
        del packs
        torch.cuda.empty_cache()
    print("weighted per step (launch counts of a C2 step): r3 plan %.1f us, best per shape %.1f us" % (base_sum, best_sum))
    for k in ("LFDM_WINO_BN", "LFDM_WINO_KG"):
        os.environ.pop(k, None)


if __name__ == "__main__":
    main()
