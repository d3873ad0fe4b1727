"""Independent pin of the rotary position embedding (reference call sites
DM/modules/video_flow_diffusion.py:15 (import), :395 (`RotaryEmbedding(min(32, attn_dim_head))`), :329-331
(`rotate_queries_or_keys` on q and k of the temporal attention)).

The reference takes the function from the third-party package rotary_embedding_torch==0.1.5, which is absent from this image; the
fixtures were minted through the 28-line restatement oracle/ref_shims/rotary_embedding_torch.py.  This file checks that restatement,
the oracle's own tables and the product's `rotary_freqs` against an implementation written by somebody else that IS installed:
GPT-J's rotary embedding in `transformers` (the same interleaved-pair convention: angle(pos, 2j) = angle(pos, 2j+1) =
pos * theta^(-2j/dim); (x[2j], x[2j+1]) rotated as a pair), at the temporal attention's shape - 40 frames, dim_head 32, theta 10 000.
CPU only; no GPU, no reference import.
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
gptj = pytest.importorskip("transformers.models.gptj.modeling_gptj")
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))        # (not ref_shims: its torchvision / cv2 stand-ins must not shadow real packages)

T, DIM = 40, 32


def gptj_rotate(t):
    """t (..., T, DIM) -> GPT-J's rotation of the same tensor: its functions want (batch, seq, heads, dim) and the (seq, dim/2) sin / cos
    halves of `create_sinusoidal_positions`."""
    lead = t.shape[:-2]
    x = t.reshape(-1, T, DIM).permute(1, 0, 2).unsqueeze(0)              # (1, T, heads*, DIM)
    sincos = gptj.create_sinusoidal_positions(T, DIM)                       # (T, DIM): sin | cos
    sin, cos = sincos[:, :DIM // 2].unsqueeze(0), sincos[:, DIM // 2:].unsqueeze(0)
    y = gptj.apply_rotary_pos_emb(x, sin, cos)
    return y.squeeze(0).permute(1, 0, 2).reshape(*lead, T, DIM)


def _q():
    g = torch.Generator().manual_seed(5)
    return torch.randn(2, 3, 8, T, DIM, generator=g)


def test_shim_matches_gptj():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_rotary_shim", os.path.join(os.path.dirname(HERE), "oracle", "ref_shims", "rotary_embedding_torch.py"))
    shim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shim)
    q = _q()
    got = shim.RotaryEmbedding(DIM).rotate_queries_or_keys(q)
    want = gptj_rotate(q)
    assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())


def test_oracle_tables_match_gptj():
    import lfdm_oracle as O
    from cvpr23_lfdm_amd.params import rotary_freqs
    freqs = rotary_freqs(DIM)
    inv = 1.0 / (10000 ** (torch.arange(0, DIM, 2, dtype=torch.int64) / DIM))      # GPT-J's inv_freq
    assert float((freqs - inv.float()).abs().max()) <= 1e-7
    cos, sin = O.rotary_tables(freqs, T)
    q = _q()
    got = O.apply_rotary(q, cos, sin)
    want = gptj_rotate(q)
    assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())


def test_minted_fixture_matches_gptj():
    """The committed golden vector (`ops.npz`: rot_q -> rot_out, written by oracle/make_golden.py through the shim) against GPT-J."""
    path = os.path.join(HERE, "golden", "ops.npz")
    if not os.path.exists(path):
        pytest.skip("ops.npz not generated")
    g = np.load(path)
    q, out = torch.from_numpy(g["rot_q"]), torch.from_numpy(g["rot_out"])
    assert q.shape[-2:] == (T, DIM)
    want = gptj_rotate(q)
    assert float((out - want).abs().max()) <= 2e-6 * float(want.abs().max())
