#!/usr/bin/env python
"""Runs one conv shape N times (for rocprofv3 --pmc passes). Usage: probe_one_conv.py cin cout k s frames iters"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr23_lfdm_amd import ops  # noqa: E402

cin, cout, k, s, frames, iters = [int(v) for v in sys.argv[1:7]]
m = frames * s * s
x = torch.randn(m, cin, device="cuda")
raw = torch.randn(cout, cin, k, k, device="cuda") * 0.05
w = ops.pack_conv_weight(raw)
ww = ops.pack_wino_weight(raw) if (k == 3 and cin % 16 == 0) else None      # LFDM_WINO=0 in the environment forces the direct form
b = torch.randn(cout, device="cuda")
out = torch.empty(m, cout, device="cuda")
pp, _ = ops.conv_params(x, w, cout, k, k, frames, s, s, bias=b, out=out, weight_wino=ww)
rows, ks = ops.conv_plan(pp)
if ks > 1:
    partial = torch.empty(ks * m * w.shape[1], device="cuda")
    pp.partial = partial.data_ptr()
for _ in range(iters):
    ops.conv_launch(pp)
torch.cuda.synchronize()
