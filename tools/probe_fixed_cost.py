#!/usr/bin/env python
"""Per-launch fixed cost of the conv kernels: 1x1 convs at M = 40960 rows with K = 32..512 (1..16 chunks), forced onto
the KSW schedule (LFDM_CONV_FORCE=1) or the igemm schedule (=0); time vs chunks -> slope and intercept."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr23_lfdm_amd import ops  # noqa: E402

m, s, frames = 40960, 32, 40
for cout in (64, 128):
    for cin in (32, 64, 128, 256, 512):
        x = torch.randn(m, cin, device="cuda")
        w = ops.pack_conv_weight(torch.randn(cout, cin, 1, 1, device="cuda") * 0.05)
        out = torch.empty(m, cout, device="cuda")
        pp, _ = ops.conv_params(x, w, cout, 1, 1, frames, s, s, out=out)
        rows, ks = ops.conv_plan(pp)
        for _ in range(5):
            ops.conv_launch(pp)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                ops.conv_launch(pp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        print("force=%s cout=%3d cin=%3d chunks=%2d rows=%d ksplit=%d  %.2f us" % (os.environ.get("LFDM_CONV_FORCE", "-"), cout, cin, cin // 32, rows, ks, us))
