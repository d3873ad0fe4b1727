"""Stand-in for einops_exts==0.0.3 (reference requirements.txt:2); only
`rearrange_many` is used by the reference (video_flow_diffusion.py:13,254,321)."""
from einops import rearrange


def rearrange_many(tensors, pattern, **axes):
    return tuple(rearrange(t, pattern, **axes) for t in tensors)
