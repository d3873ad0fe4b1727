#!/usr/bin/env python
"""linear_attention_bwd (context + apply kernels) at the DM training step's shapes (320 frames), event-timed.  Usage: bench_linattn_bwd.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvpr23_lfdm_amd import train_ops  # noqa: E402

for hw in (1024, 256, 64, 16):
    nf = 320
    qkv = torch.randn(nf * hw, 768, device="cuda")
    dout = torch.randn(nf * hw, 256, device="cuda")
    fn = lambda: train_ops.linear_attention_bwd(qkv, dout, nf, hw)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    gb = nf * hw * (768 * 2 + 256 + 768) * 4 / 1e9            # qkv read twice (context + apply), dout, dqkv write
    print("hw %5d  %8.1f us per call (context + apply)   %.2f TB/s of qkv x2 + dout + dqkv" % (hw, us, gb / us * 1e-6 * 1e6 / 1e6 * 1e0 if False else gb / (us * 1e-6) / 1e3), flush=True)
