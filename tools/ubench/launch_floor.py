#!/usr/bin/env python
"""What one kernel boundary costs inside a replayed hipGraph on this box: N back-to-back launches of (a) a 1-workgroup
kernel (step_cond on one row), (b) a small streaming kernel (GroupNorm apply of a 1.3 MB tensor), captured and replayed;
and the same (a) launched eagerly.  Prints us per launch."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvpr23_lfdm_amd import ops

dev = "cuda"
N = 200
step_part = torch.randn(4, 256, device=dev); sample = torch.randn(1, 256, device=dev)
step_dev = torch.zeros(1, dtype=torch.int32, device=dev); out = torch.empty(1, 256, device=dev)


def tiny():
    ops.step_cond(step_part, sample, step_dev, out)


def timeit(fn, n, graph):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (5 * n)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n, (time.perf_counter() - t0) * 1e6 / n


print("tiny kernel, graph replay : %.2f us/launch" % timeit(tiny, N, True))
# two DIFFERENT tiny kernels alternating (different code, different LDS / register configuration): what a real step looks like
tdev = torch.zeros(4, dtype=torch.int32, device=dev); freqs = torch.randn(32, device=dev); semb = torch.empty(4, 64, device=dev)
lin_x = torch.randn(1, 64, device=dev); lin_w = torch.randn(64, 64, device=dev); lin_b = torch.randn(64, device=dev); lin_o = torch.empty(1, 64, device=dev)


def two_kinds():
    ops.step_cond(step_part, sample, step_dev, out)
    ops.linear_small(lin_x, lin_w, lin_b, out=lin_o)


print("two different tiny kernels alternating, graph replay : %.2f us/launch" % (timeit(two_kinds, N // 2, True) / 2))
xs = torch.randn(640, 512, device=dev); g512 = torch.ones(512, device=dev); b512 = torch.zeros(512, device=dev)
wsx = torch.empty(256 * 128 + 2048, device=dev); ys = torch.empty_like(xs); pt = torch.randn(5, 16, device=dev).abs()


def tiny_then_stream():
    ops.step_cond(step_part, sample, step_dev, out)
    ops.groupnorm_apply_cl(xs, 1, g512, b512, pt, 5, out=ys, ws=wsx)


print("tiny kernel + 1.3 MB GroupNorm apply alternating : %.2f us per PAIR (apply alone 3.9, tiny alone 1.6)" % timeit(tiny_then_stream, N // 2, True))
print("tiny kernel, eager        : %.2f us/launch (host %.2f us/launch)" % timeit(tiny, N, False))
for rows, ch in ((640, 512), (2560, 256), (10240, 128), (40960, 64)):
    x = torch.randn(rows, ch, device=dev); gamma = torch.ones(ch, device=dev); beta = torch.zeros(ch, device=dev)
    ws = torch.empty(256 * 128 + 2048, device=dev)
    y = torch.empty_like(x)
    fn = lambda: ops.groupnorm_silu_cl(x, 1, gamma, beta, out=y, ws=ws)
    print("groupnorm_silu (stats+apply, 2 launches) %5d x %3d: %.2f us per call in a graph" % (rows, ch, timeit(fn, 50, True)))
    for nchunk in (rows // 128, max(1, rows // 1280), 1):
        part = torch.randn(nchunk, 16, device=dev).abs()
        fn2 = lambda: ops.groupnorm_apply_cl(x, 1, gamma, beta, part, nchunk, out=y, ws=ws)
        print("groupnorm_apply (1 launch, %3d partial chunks) %5d x %3d: %.2f us per call in a graph" % (nchunk, rows, ch, timeit(fn2, 50, True)))
