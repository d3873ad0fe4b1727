"""Python face of the TRAINING (backward) entry points of include/lfdm_hip.h - thin ctypes calls, CL rows.

Used by cvpr23_lfdm_amd/autograd.py (the torch.autograd.Function wrappers that let `loss.backward()` of the
reference's training scripts run through the native kernels).
"""
import ctypes as C

import torch

from ._native import WgradParams
from .ops import _chk, _lib, _p, _stream


def _ws(lib_bytes, like):
    n = (int(lib_bytes) + 3) // 4
    return torch.empty(max(n, 1), dtype=torch.float32, device=like.device)


def sum_leading(x, s, n, out=None):
    lib = _lib()
    _chk(lib, x, out)
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_sum_leading_f32(_p(x), _p(out), n, s, _stream(lib)), "lfdm_sum_leading_f32")
    return out


def colsum(x, out=None):
    """(rows, C) CL rows (row stride = x.stride(0)) -> (C,) column sums (bias gradient)."""
    lib = _lib()
    _chk(lib, x, out)
    rows, c = x.shape
    assert x.stride(1) == 1
    if out is None:
        out = torch.empty(c, dtype=torch.float32, device=x.device)
    nbytes = lib.lfdm_colsum_ws_bytes(rows, c)
    ws = _ws(nbytes, x)
    lib.check(lib.lfdm_colsum_f32(_p(x), rows, c, x.stride(0), _p(out), _p(ws), nbytes, _stream(lib)), "lfdm_colsum_f32")
    return out


def conv_wgrad(x, dy, n_img, hi, wi, hq, wq, kh, kw, stride=1, pad=None):
    """dW in tap-major layout (kh*kw, cin, cout).  x: (n_img*hi*wi, cin) CL, dy: (n_img*hq*wq, cout) CL."""
    lib = _lib()
    _chk(lib, x, dy)
    assert x.stride(1) == 1 and dy.stride(1) == 1
    cin, cout = x.shape[1], dy.shape[1]
    assert x.shape[0] == n_img * hi * wi and dy.shape[0] == n_img * hq * wq
    if pad is None:
        pad = (kh // 2, kw // 2)
    dw = torch.empty(kh * kw, cin, cout, dtype=torch.float32, device=x.device)
    p = WgradParams()
    p.x, p.cin, p.ldx = x.data_ptr(), cin, x.stride(0)
    p.n_img, p.hi, p.wi, p.hq, p.wq = n_img, hi, wi, hq, wq
    p.stride, p.kh, p.kw, p.pad_y, p.pad_x = stride, kh, kw, pad[0], pad[1]
    p.dy, p.cout, p.lddy = dy.data_ptr(), cout, dy.stride(0)
    p.dw = dw.data_ptr()
    nbytes = lib.lfdm_conv2d_wgrad_ws_bytes(C.byref(p))
    ws = _ws(nbytes, x)
    lib.check(lib.lfdm_conv2d_wgrad_cl_f32(C.byref(p), _p(ws), nbytes, _stream(lib)), "lfdm_conv2d_wgrad_cl_f32")
    return dw


def gn_forward_chunks(pixels):
    """number of (sum, sumsq) partial chunks lfdm_groupnorm_silu_cl_f32 writes at the start of its workspace"""
    return min(max((pixels + 15) // 16, 1), 256)


def groupnorm_silu_train(x, batch, gamma, beta, scale_shift=None, residual=None, groups=8, eps=1e-5, silu=True):
    """Forward for training: out-of-place, returns (y, partial) with partial = the statistics the backward needs."""
    from . import ops
    lib = _lib()
    rows, c = x.shape
    pixels = rows // batch
    nbytes = lib.lfdm_groupnorm_ws_bytes(batch, pixels, c)
    ws = _ws(nbytes, x)
    y = torch.empty_like(x)
    ops.groupnorm_silu_cl(x, batch, gamma, beta, scale_shift=scale_shift, residual=residual, groups=groups, eps=eps,
                          silu=silu, out=y, ws=ws)
    nchunk = gn_forward_chunks(pixels)
    return y, ws[: batch * nchunk * groups * 2], nchunk


def groupnorm_silu_bwd(x, dy, batch, gamma, beta, partial, nchunk, scale_shift=None, groups=8, eps=1e-5, silu=True):
    """-> (dx, dgamma, dbeta, dscale_shift or None)."""
    lib = _lib()
    _chk(lib, x, dy, gamma, beta, partial, scale_shift)
    rows, c = x.shape
    pixels = rows // batch
    assert x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(x)
    dgb = torch.empty(2, c, dtype=torch.float32, device=x.device)
    dss = torch.empty(batch, 2 * c, dtype=torch.float32, device=x.device) if scale_shift is not None else None
    nbytes = lib.lfdm_groupnorm_bwd_ws_bytes(batch, pixels, c)
    ws = _ws(nbytes, x)
    lib.check(lib.lfdm_groupnorm_silu_bwd_cl_f32(
        _p(x), _p(dy), _p(dx), batch, pixels, c, groups, _p(gamma), _p(beta), _p(scale_shift),
        scale_shift.stride(0) if scale_shift is not None else 0, eps, 1 if silu else 0, _p(partial), nchunk, _p(dgb),
        _p(dss), 2 * c, _p(ws), nbytes, _stream(lib)), "lfdm_groupnorm_silu_bwd_cl_f32")
    return dx, dgb[0], dgb[1], dss


def layernorm_bwd(x, dy, gamma, eps=1e-5):
    """-> (dx, dgamma)."""
    lib = _lib()
    _chk(lib, x, dy, gamma)
    rows, c = x.shape
    assert x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    nbytes = lib.lfdm_layernorm_bwd_ws_bytes(rows, c)
    ws = _ws(nbytes, x)
    lib.check(lib.lfdm_layernorm_bwd_cl_f32(_p(x), _p(dy), _p(dx), rows, c, _p(gamma), eps, _p(dgamma), _p(ws), nbytes,
                                            _stream(lib)), "lfdm_layernorm_bwd_cl_f32")
    return dx, dgamma


def attention_bwd(qkv, dout, batch, frames, hw, mode, bias=None, rot_cos=None, rot_sin=None):
    """-> (dqkv rows of 768, dbias (8, L, L) or None)."""
    lib = _lib()
    _chk(lib, qkv, dout, bias, rot_cos, rot_sin)
    assert qkv.is_contiguous() and dout.is_contiguous() and qkv.shape[1] == 768 and dout.shape[1] == 256
    dqkv = torch.empty_like(qkv)
    dbias, ws, nbytes = None, None, 0
    if bias is not None:
        seq = frames if mode == 0 else hw
        dbias = torch.empty(8, seq, seq, dtype=torch.float32, device=qkv.device)
        nbytes = lib.lfdm_attention_bwd_ws_bytes(batch, frames, hw, mode)
        ws = _ws(nbytes, qkv)
    lib.check(lib.lfdm_attention_bwd_cl_f32(_p(qkv), _p(dout), _p(dqkv), batch, frames, hw, mode, _p(bias), _p(rot_cos),
                                            _p(rot_sin), _p(dbias), _p(ws), nbytes, _stream(lib)),
              "lfdm_attention_bwd_cl_f32")
    return dqkv, dbias


def linear_attention_bwd(qkv, dout, n_frames, hw):
    lib = _lib()
    _chk(lib, qkv, dout)
    assert qkv.is_contiguous() and dout.is_contiguous()
    dqkv = torch.empty_like(qkv)
    nbytes = lib.lfdm_linear_attention_bwd_ws_bytes(n_frames)
    ws = _ws(nbytes, qkv)
    lib.check(lib.lfdm_linear_attention_bwd_cl_f32(_p(qkv), _p(dout), _p(dqkv), n_frames, hw, _p(ws), nbytes, _stream(lib)),
              "lfdm_linear_attention_bwd_cl_f32")
    return dqkv


def upsample2_pad(x, n_img, h, w, pad, reflect, backward=False):
    """forward: (n*h*w, C) -> (n*(2h+2pad)*(2w+2pad), C); backward: the adjoint."""
    lib = _lib()
    _chk(lib, x)
    assert x.is_contiguous()
    c = x.shape[1]
    rows = n_img * h * w if backward else n_img * (2 * h + 2 * pad) * (2 * w + 2 * pad)
    assert x.shape[0] == (n_img * (2 * h + 2 * pad) * (2 * w + 2 * pad) if backward else n_img * h * w)
    out = torch.empty(rows, c, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_upsample2_pad_cl_f32(_p(x), _p(out), n_img, h, w, c, pad, int(reflect), int(backward), _stream(lib)),
              "lfdm_upsample2_pad_cl_f32")
    return out
