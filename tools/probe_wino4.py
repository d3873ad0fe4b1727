#!/usr/bin/env python
"""A few launches of ONE batched 3x3 shape on the F(4x4) (default) or F(2x2) (LFDM_WINO4=0) Winograd schedule - the target of
rocprofv3 --pmc passes (tools/prof_wino4.sh).  Usage: probe_wino4.py [n_img cin cout h w residual]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr23_lfdm_amd import ops  # noqa: E402

os.environ.setdefault("LFDM_WINO4_MIN", "1")
a = [int(v) for v in sys.argv[1:]] + [320, 256, 256, 32, 32, 1][len(sys.argv) - 1:]
n, cin, cout, h, w, resid = a
dev = "cuda"
g = torch.Generator().manual_seed(1)
x = torch.randn(n * h * w, cin, generator=g).to(dev)
wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)).to(dev)
res = torch.randn(n * h * w, cout, generator=g).to(dev) if resid else None
ww, w4 = ops.pack_wino_weight(wt), ops.pack_wino4_weight(wt)
out = torch.empty(n * h * w, cout, device=dev)
pp, _ = ops.conv_params(x, None, cout, 3, 3, n, h, w, residual=res, out=out, weight_wino=ww, weight_wino4=w4)
for _ in range(3):
    ops.conv_launch(pp)
torch.cuda.synchronize()
print("PROBE done")
