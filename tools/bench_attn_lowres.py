#!/usr/bin/env python
"""The attention blocks (without to_out) of the low-resolution levels of one B = 1, T = 40 sampler step: separate launches
(LayerNorm-folded to_qkv projection [+ split-K reduce] + attention core) against the one-launch kernels of attn_lowres.hip.
us per block inside a replayed hipGraph (20 blocks per replay, each on its own input and weights: cold operands).  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvpr23_lfdm_amd import ops  # noqa: E402

T = 40
NCOPY = 20


def timed(fns, reps=10):
    for f in fns[:2]:
        f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (len(fns) * reps)


def main():
    dev = "cuda"
    print("%-28s %9s %9s" % ("block", "separate", "one launch"))
    for kind, s, c in (("linear", 16, 128), ("linear", 8, 256), ("linear", 4, 512), ("linear", 4, 256), ("linear", 8, 128),
                       ("temporal", 16, 128), ("temporal", 8, 256), ("temporal", 4, 512), ("temporal", 4, 256), ("temporal", 8, 128),
                       ("spatial", 4, 512)):
        hw, rows = s * s, T * s * s
        old, new = [], []
        keep = []
        bias = torch.randn(8, T, T, device=dev)
        cos, sin = torch.rand(T, 16, device=dev), torch.rand(T, 16, device=dev)
        for i in range(NCOPY):
            x = torch.randn(rows, c, device=dev)
            w = torch.randn(768, c, device=dev) / c ** 0.5
            gam = torch.rand(c, device=dev) + 0.5
            packed, wsum = ops.pack_ln_conv_weight(w, gam)
            wf = (w * gam.reshape(1, -1)).contiguous()
            qkv = torch.empty(rows, 768, device=dev)
            att = torch.empty(rows, 256, device=dev)
            pp, _ = ops.conv_params(x, packed, 768, 1, 1, T, s, s, out=qkv, ln_wsum=wsum)
            _, ks = ops.conv_plan(pp)
            part = None
            if ks > 1:
                part = torch.empty(ops.conv_partial_floats(pp), device=dev)
                pp.partial = part.data_ptr()
            ws = torch.empty(T * 8 * 32 * 32, device=dev)
            keep.append((x, w, packed, wsum, wf, qkv, att, pp, part, ws))
            if kind == "linear":
                old.append(lambda pp=pp, qkv=qkv, att=att, ws=ws: (ops.conv_launch(pp), ops.linear_attention_cl(qkv, T, hw, out=att, ws=ws)))
                new.append(lambda x=x, wf=wf, wsum=wsum, att=att: ops.linear_attention_lowres_cl(x, wf, wsum, T, hw, out=att))
            elif kind == "temporal":
                old.append(lambda pp=pp, qkv=qkv, att=att: (ops.conv_launch(pp), ops.attention_cl(qkv, 1, T, hw, 0, bias=bias, rot_cos=cos, rot_sin=sin, out=att)))
                new.append(lambda x=x, wf=wf, wsum=wsum, att=att: ops.attention_lowres_cl(x, wf, wsum, 1, T, hw, 0, bias=bias, rot_cos=cos, rot_sin=sin, out=att))
            else:
                old.append(lambda pp=pp, qkv=qkv, att=att: (ops.conv_launch(pp), ops.attention_cl(qkv, 1, T, hw, 1, out=att)))
                new.append(lambda x=x, wf=wf, wsum=wsum, att=att: ops.attention_lowres_cl(x, wf, wsum, 1, T, hw, 1, out=att))
        print("%-28s %9.1f %9.1f" % ("%s %dx%d C=%d" % (kind, s, s, c), timed(old), timed(new)))


if __name__ == "__main__":
    main()
