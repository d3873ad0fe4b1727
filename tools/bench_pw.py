#!/usr/bin/env python
"""The 1x1 projections of one B = 1, T = 40 sampler step (to_qkv with the LayerNorm fold, to_out, res_conv) on the
LDS-staged schedules (LFDM_PW=0: conv_igemm / conv_ksw + their split-K reduce launch) against the pointwise schedule (conv_pw.hip),
the latter with the planner's tile shape and with every forced (TN, KW).  us per call inside a replayed hipGraph.  GPU only."""
import ctypes
import os
import sys

import torch

os.environ.setdefault("LFDM_PW_MAXM", "1000000")      # read once by the library: the sweep includes the 40 960-row level

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr23_lfdm_amd import ops  # noqa: E402

T = int(os.environ.get("FRAMES", "40"))
# (name, resolution, c0, c1, cout, LayerNorm fold, count per step)
SHAPES = [
    ("to_out 256->64 @32", 32, 256, 0, 64, False, 6), ("res 64+64->64 @32", 32, 64, 64, 64, False, 1), ("heads res 64+64->128 @32", 32, 64, 64, 128, False, 1),
    ("res 64->128 @16", 16, 64, 0, 128, False, 1), ("qkv 128->768 @16", 16, 128, 0, 768, True, 2), ("to_out 256->128 @16", 16, 256, 0, 128, False, 2),
    ("res 128+128->64 @16", 16, 128, 128, 64, False, 1), ("to_out 256->64 @16", 16, 256, 0, 64, False, 2),
    ("res 128->256 @8", 8, 128, 0, 256, False, 1), ("qkv 256->768 @8", 8, 256, 0, 768, True, 2), ("to_out 256->256 @8", 8, 256, 0, 256, False, 2),
    ("res 256+256->128 @8", 8, 256, 256, 128, False, 1), ("qkv 128->768 @8", 8, 128, 0, 768, True, 2), ("to_out 256->128 @8", 8, 256, 0, 128, False, 2),
    ("res 256->512 @4", 4, 256, 0, 512, False, 1), ("qkv 512->768 @4", 4, 512, 0, 768, True, 4), ("to_out 256->512 @4", 4, 256, 0, 512, False, 4),
    ("res 512+512->256 @4", 4, 512, 512, 256, False, 1), ("qkv 256->768 @4", 4, 256, 0, 768, True, 2), ("to_out 256->256 @4", 4, 256, 0, 256, False, 2),
]


def timed(fn, n=20, reps=10):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


def main():
    dev = "cuda"
    lib = ops._lib()
    tot = {"old": 0.0, "pw": 0.0, "best": 0.0}
    print("%-26s %7s | %8s | %8s (tn,kw) | forced (tn,kw): us ..." % ("shape", "rows", "old us", "pw us"))
    for name, s, c0, c1, cout, ln, count in SHAPES:
        m = T * s * s
        x0 = torch.randn(m, c0, device=dev)
        x1 = torch.randn(m, c1, device=dev) if c1 else None
        cin = c0 + c1
        w = torch.randn(cout, cin, device=dev) / cin ** 0.5
        kw = {}
        if ln:
            packed, wsum = ops.pack_ln_conv_weight(w, torch.rand(cin, device=dev) + 0.5)
            kw["ln_wsum"] = wsum
        else:
            packed = ops.pack_conv_weight(w)
        res = torch.randn(m, cout, device=dev)
        out = torch.empty(m, cout, device=dev)
        bias = None if ln else torch.randn(cout, device=dev)

        def make():
            pp, _ = ops.conv_params(x0, packed, cout, 1, 1, T, s, s, src1=x1, bias=bias, residual=None if ln else res, out=out, **kw)
            _, ks = ops.conv_plan(pp)
            keep = None
            if ks > 1:
                keep = torch.empty(ops.conv_partial_floats(pp), device=dev)
                pp.partial = keep.data_ptr()
            return pp, keep, ks, lib.lfdm_conv2d_schedule(ctypes.byref(pp))

        os.environ["LFDM_PW"] = "0"
        pp, keep, ks, kind = make()
        t_old = timed(lambda: ops.conv_launch(pp))
        os.environ["LFDM_PW"] = "2"        # every eligible geometry
        os.environ.pop("LFDM_PW_TN", None)
        os.environ.pop("LFDM_PW_KW", None)
        pp, keep, _, kind_pw = make()
        assert kind_pw == 3, kind_pw
        t_pw = timed(lambda: ops.conv_launch(pp))
        forced = []
        for kwv in (4, 2, 1):
            if cin % (32 * kwv):
                continue
            for tn in (1, 2, 3):
                os.environ["LFDM_PW_TN"], os.environ["LFDM_PW_KW"] = str(tn), str(kwv)
                forced.append(((tn, kwv), timed(lambda: ops.conv_launch(pp))))
        os.environ.pop("LFDM_PW_TN", None)
        os.environ.pop("LFDM_PW_KW", None)
        best = min(forced, key=lambda f: f[1])
        print("%-26s %7d | %6.1f k%d s%d | %6.1f | best %s %5.1f | %s" % (
            name, m, t_old, ks, kind, t_pw, best[0], best[1], " ".join("%d,%d:%.1f" % (a[0], a[1], b) for a, b in forced)))
        tot["old"] += count * t_old
        tot["pw"] += count * t_pw
        tot["best"] += count * best[1]
    print("weighted per step: old %.0f us   pointwise (planner's shape) %.0f us   pointwise (best forced shape) %.0f us" % (
        tot["old"], tot["pw"], tot["best"]))


if __name__ == "__main__":
    main()
