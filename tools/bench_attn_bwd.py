#!/usr/bin/env python
"""attention_bwd_kernel at the DM training step's shapes (B = 8 videos of 40 frames; temporal attention at 32x32 / 16x16 / 8x8), event-timed:
the 40-row form against the 48-row form (LFDM_ATTN_BWD_ROWS40=0).  Usage: bench_attn_bwd.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from cvpr23_lfdm_amd import train_ops
    torch.manual_seed(0)
    for hw in (1024, 256, 64):
        b, frames = 8, 40
        qkv = torch.randn(b * frames * hw, 768, device="cuda")
        dout = torch.randn(b * frames * hw, 256, device="cuda")
        bias = torch.randn(8, frames, frames, device="cuda")
        cos, sin = torch.rand(frames, 16, device="cuda"), torch.rand(frames, 16, device="cuda")
        fn = lambda: train_ops.attention_bwd(qkv, dout, b, frames, hw, 0, bias=bias, rot_cos=cos, rot_sin=sin)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print("hw %5d  %8.1f us per call (incl. the bias-partial sum)" % (hw, e0.elapsed_time(e1) * 100), flush=True)
    sys.exit(0)

for knob in ("1", "0"):
    env = dict(os.environ, LFDM_ATTN_BWD_ROWS40=knob)
    print("# LFDM_ATTN_BWD_ROWS40=%s (%s)" % (knob, "40-row tiles, wave-level LDS ordering" if knob == "1" else "48-row tiles, workgroup barriers"), flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
