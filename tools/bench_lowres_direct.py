#!/usr/bin/env python
"""Round-6 A/B asked by the round-5 review: at the 4x4 / 8x8 levels of a B = 1 step, the DIRECT 3x3 form on the weight-stationary pointwise
schedule (conv_pw.hip: operands straight to registers, K split over the waves) against the Winograd launch the sampler uses (split-K slabs
reduced inside the launch).  The direct form is measured at its best case: the pointwise kernel on a PRE-BUILT im2col of the input
(M x 9 C_in rows; the gather itself is not timed), i.e. the register-operand GEMM with 9/16 of the Winograd filter bytes and no transforms.
Every launch reads its own cold filter out of a ring larger than the Infinity Cache, launches are replayed from a hipGraph.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr23_lfdm_amd import ops  # noqa: E402

FRAMES = 40


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
    print("%-16s %-44s %8s %7s" % ("shape", "schedule", "us/conv", "TF/s"))
    for name, cin, cout, s in (("512->512 @4", 512, 512, 4), ("256->512 @4", 256, 512, 4), ("1024->512 @4", 1024, 512, 4),
                               ("256->256 @8", 256, 256, 8), ("512->256 @8", 512, 256, 8)):
        m = FRAMES * s * s
        gf = 2.0 * m * cout * cin * 9 / 1e9
        x = torch.randn(m, cin, device=dev)
        bias = torch.randn(cout, device=dev)
        out = torch.empty(m, cout, device=dev)
        ring = max(4, min(24, int(300e6 // (16 * cin * cout * 4)) + 1))
        counters = torch.zeros(4096, dtype=torch.int32, device=dev)
        wd = ops.pack_conv_weight(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
        # --- the sampler's launch: Winograd F(2x2), split-K reduced in the launch, GroupNorm partial sums requested
        fns, keep = [], []
        for _ in range(ring):
            ww = ops.pack_wino_weight(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
            pp, _ = ops.conv_params(x, wd, cout, 3, 3, FRAMES, s, s, bias=bias, out=out, weight_wino=ww, tile_counters=counters)
            pp.gn_partial = 1
            tile_rows, ksplit = ops.conv_plan(pp)
            part = torch.empty(max(1, ops.conv_partial_floats(pp)), device=dev)
            pp.partial = part.data_ptr()
            parts = max(1, (cout // 8) // 32)
            gnp = torch.empty(m // tile_rows * parts, 16, device=dev)
            pp.gn_partial, pp.gn_groups, pp.gn_pixels = gnp.data_ptr(), 8, m
            keep += [ww, part, gnp, pp]
            fns.append(lambda pp=pp: ops.conv_launch(pp))
        us = graph_us(fns)
        print("%-16s %-44s %8.2f %7.1f" % (name, "Winograd F(2x2), ksplit %d reduced in-launch" % ksplit, us, gf / us * 1e3), flush=True)
        del fns, keep
        # --- direct form, best case: 1x1 pointwise GEMM over K = 9 C_in (im2col prebuilt, not timed)
        col = torch.randn(m, 9 * cin, device=dev)
        fns, keep = [], []
        for _ in range(ring):
            w1 = ops.pack_conv_weight(torch.randn(cout, 9 * cin, 1, 1, device=dev) * 0.02)
            wp = ops.pack_pw_weight(w1)
            pp, _ = ops.conv_params(col, w1, cout, 1, 1, FRAMES, s, s, bias=bias, out=out, weight_pw=wp)
            kind = ops.conv_schedule(pp)
            tile_rows, ksplit = ops.conv_plan(pp)
            part = torch.empty(max(1, ops.conv_partial_floats(pp)), device=dev)
            pp.partial = part.data_ptr()
            keep += [w1, wp, part, pp]
            fns.append(lambda pp=pp: ops.conv_launch(pp))
        us = graph_us(fns)
        print("%-16s %-44s %8.2f %7.1f" % (name, "direct 3x3 as K = 9 C GEMM, schedule %d ksplit %d" % (kind, ksplit), us, gf / us * 1e3), flush=True)
        del fns, keep, col
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
