#!/usr/bin/env python
"""Micro-benchmark of lfdm_conv2d_cl_f32 on the contraction shapes of one C2 UNet step
 and of the LFAE decode; prints TFLOP/s per shape.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr23_lfdm_amd import ops  # noqa: E402

FRAMES = int(os.environ.get("FRAMES", "40"))
SHAPES = [
    # (name, cin, cout, k, s, count_per_step)
    ("3x3 64->64 @32", 64, 64, 3, 32, 9), ("3x3 128->64 @32", 128, 64, 3, 32, 3), ("1x1 64->768 @32", 64, 768, 1, 32, 5),
    ("1x1 256->64 @32", 256, 64, 1, 32, 5), ("1x1 128->64 @32", 128, 64, 1, 32, 2),
    ("3x3 64->128 @16", 64, 128, 3, 16, 1), ("3x3 128->128 @16", 128, 128, 3, 16, 5), ("3x3 256->128 @16", 256, 128, 3, 16, 1),
    ("1x1 128->768 @16", 128, 768, 1, 16, 4), ("1x1 256->128 @16", 256, 128, 1, 16, 4),
    ("3x3 128->256 @8", 128, 256, 3, 8, 1), ("3x3 256->256 @8", 256, 256, 3, 8, 5), ("3x3 512->256 @8", 512, 256, 3, 8, 1),
    ("1x1 256->768 @8", 256, 768, 1, 8, 4), ("1x1 256->256 @8", 256, 256, 1, 8, 4),
    ("3x3 256->512 @4", 256, 512, 3, 4, 1), ("3x3 512->512 @4", 512, 512, 3, 4, 7), ("3x3 1024->256 @4", 1024, 256, 3, 4, 1),
    ("1x1 512->768 @4", 512, 768, 1, 4, 5), ("1x1 256->512 @4", 256, 512, 1, 4, 5),
    ("dec 3x3 256->256 @32", 256, 256, 3, 32, 0), ("dec 3x3 128->64 @128(8f)", 128, 64, 3, 128, 0),
]


def main():
    dev = "cuda"
    import importlib
    unet_mod = importlib.import_module("cvpr23_lfdm_amd.unet")
    helper = unet_mod.Unet3D.__new__(unet_mod.Unet3D)      # only for the split-K heuristic
    tot_ms = 0.0
    tot_gf = 0.0
    tot_wino = 0.0
    print("%-28s %9s %8s %8s %8s %5s" % ("shape", "GFLOP", "us", "TF/s", "ksplit", "rows"))
    for name, cin, cout, k, s, count in SHAPES:
        n_img = FRAMES if "8f" not in name else 8
        m = n_img * s * s
        x = torch.randn(m, cin, device=dev)
        b = torch.randn(cout, device=dev)
        out = torch.empty(m, cout, device=dev)
        raw = torch.randn(cout, cin, k, k, device=dev) * 0.05
        w = ops.pack_conv_weight(raw)
        ww = ops.pack_wino_weight(raw) if (k == 3 and cin % 16 == 0) else None
        gf = 2.0 * m * cout * cin * k * k / 1e9
        cols = []
        for wino in ([False, True] if ww is not None else [False]):
            os.environ["LFDM_WINO"] = "1" if wino else "0"
            pp, _ = ops.conv_params(x, w, cout, k, k, n_img, s, s, bias=b, out=out, ksplit=int(os.environ.get("KSPLIT", "0")),
                                    weight_wino=ww)
            rows_per_tile, ksplit = ops.conv_plan(pp)
            if ksplit > 1:
                partial = torch.empty(ksplit * m * w.shape[1], device=dev)
                pp.partial = partial.data_ptr()
            run = lambda: ops.conv_launch(pp)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 20
            e0.record()
            for _ in range(iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            cols.append((e0.elapsed_time(e1) * 1e3 / iters, ksplit, rows_per_tile))
        os.environ["LFDM_WINO"] = "0"
        us, ksplit, rows_per_tile = cols[0]
        extra = ""
        if len(cols) > 1:
            extra = "   | winograd %8.1f us  k=%d  (x%.2f)" % (cols[1][0], cols[1][1], us / cols[1][0])
            tot_wino += min(us, cols[1][0]) * count / 1e3
        else:
            tot_wino += us * count / 1e3
        print("%-28s %9.2f %8.1f %8.1f %8d %5d%s" % (name, gf, us, gf / us * 1e3, ksplit, rows_per_tile, extra))
        tot_ms += us * count / 1e3
        tot_gf += gf * count
    print("weighted per UNet step: %.1f GFLOP in %.3f ms -> %.1f TF/s" % (tot_gf, tot_ms, tot_gf / tot_ms))
    print("with the faster of direct / Winograd per shape: %.3f ms" % tot_wino)


if __name__ == "__main__":
    main()
