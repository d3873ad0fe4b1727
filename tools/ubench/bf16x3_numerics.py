#!/usr/bin/env python
"""CPU prediction of the error of the 3-term bf16 split (tools/ubench/bf16x3_gemm.hip) against plain fp32 accumulation, both
against fp64: numpy only, runs anywhere.  The six bf16 x bf16 products are exact in fp32; what differs from the fp32 contraction
is (a) the three dropped products (mid*lo, lo*mid, lo*lo: ~2^-24 relative each) and (b) six fp32 accumulation chains instead of one."""
import numpy as np


def bf16_rne(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return (u & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split3(x):
    h = bf16_rne(x)
    r1 = (x - h).astype(np.float32)
    m = bf16_rne(r1)
    r2 = (r1 - m).astype(np.float32)
    return h, m, bf16_rne(r2)


def main():
    rng = np.random.default_rng(0)
    print("%-22s %12s %12s %12s %12s" % ("shape (M, N, K)", "fp32", "bf16x3 / 6", "bf16x3 / 9", "bf16x2 / 3"))
    for m, n, k in ((256, 256, 256), (256, 512, 512), (128, 256, 2304), (128, 512, 4096)):
        a = rng.standard_normal((m, k)).astype(np.float32)
        w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
        ref = a.astype(np.float64) @ w.astype(np.float64).T
        scale = np.abs(ref).max()
        f32 = a @ w.T
        ah, am, al = split3(a)
        wh, wm, wl = split3(w)
        mm = lambda x, y: (x @ y.T).astype(np.float32)
        six = mm(al, wh) + mm(ah, wl) + mm(am, wm) + mm(am, wh) + mm(ah, wm) + mm(ah, wh)
        nine = six + mm(am, wl) + mm(al, wm) + mm(al, wl)
        three = mm(am, wh) + mm(ah, wm) + mm(ah, wh)
        e = lambda c: np.abs(c.astype(np.float64) - ref).max() / scale
        print("%-22s %12.3e %12.3e %12.3e %12.3e" % ((m, n, k), e(f32), e(six), e(nine), e(three)))
    print("(max |error| / max |C| against fp64; numpy's fp32 matmul stands in for the matrix pipe's fp32 accumulate)")


if __name__ == "__main__":
    main()
