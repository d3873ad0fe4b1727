"""Stand-in for rotary_embedding_torch==0.1.5 (reference requirements.txt:14), which is
not installed here.  Restates the published algorithm of the `freqs_for='lang'` path that
`RotaryEmbedding(32)` takes (call sites: video_flow_diffusion.py:395,329-331):
  freqs_j = theta^(-2j/dim), j < dim/2 ; angle(pos, 2j) = angle(pos, 2j+1) = pos * freqs_j
  rotate(t)[2j] = t[2j] cos - t[2j+1] sin ; rotate(t)[2j+1] = t[2j+1] cos + t[2j] sin
PARITY UNPINNED: the reference holds no test for this package.
"""
import torch
from torch import nn


class RotaryEmbedding(nn.Module):
    def __init__(self, dim, theta=10000):
        super().__init__()
        exponents = torch.arange(0, dim, 2)[: dim // 2].float() / dim
        self.register_buffer("freqs", 1.0 / (theta ** exponents))

    def angles(self, n, device, dtype):
        pos = torch.arange(n, device=device).type(self.freqs.dtype)
        ang = pos[:, None] * self.freqs[None, :]            # (n, dim/2)
        return ang.repeat_interleave(2, dim=-1).to(dtype)   # (n, dim): pairs share an angle

    def rotate_queries_or_keys(self, t, seq_dim=-2):
        assert seq_dim == -2
        ang = self.angles(t.shape[-2], t.device, t.dtype)
        even, odd = t[..., 0::2], t[..., 1::2]
        swapped = torch.stack((-odd, even), dim=-1).flatten(-2)
        return t * ang.cos() + swapped * ang.sin()
