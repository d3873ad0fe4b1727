"""Python face of the TRAINING (backward) entry points of include/lfdm_hip.h - thin ctypes calls, CL rows.

Used by cvpr23_lfdm_amd/autograd.py (the torch.autograd.Function wrappers that let `loss.backward()` of the
reference's training scripts run through the native kernels).
"""
import ctypes as C

import torch

from ._native import ML_MAX, MultiLinearParams, WgradParams
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


def conv_wgrad(x, dy, n_img, hi, wi, hq, wq, kh, kw, stride=1, pad=None, out=None, ci_off=0, dbias=None):
    """Weight gradient.  x: (n_img*hi*wi, cin) CL, dy: (n_img*hq*wq, cout) CL.
    out=None -> a new tensor in the tap-major layout (kh*kw, cin, cout).
    out=(cout, cin_total, ...) contiguous -> the reference layout is written into it at input channels [ci_off, ci_off+cin)
    (filters of <= 16 taps; lfdm_wgrad_params.dw_layout = 1) and `out` is returned.
    dbias: optional (cout,) tensor that receives the bias gradient (column sums of dy) from the same pass."""
    lib = _lib()
    _chk(lib, x, dy, out, dbias)
    assert x.stride(1) == 1 and dy.stride(1) == 1
    cin, cout = x.shape[1], dy.shape[1]
    assert x.shape[0] == n_img * hi * wi and dy.shape[0] == n_img * hq * wq
    if pad is None:
        pad = (kh // 2, kw // 2)
    p = WgradParams()
    if out is None:
        dw = torch.empty(kh * kw, cin, cout, dtype=torch.float32, device=x.device)
    else:
        dw = out
        assert out.is_contiguous() and out.dtype == torch.float32 and out.shape[0] == cout and out.numel() % (cout * kh * kw) == 0
        p.dw_layout, p.dw_cin_total, p.dw_ci_off = 1, out.numel() // (cout * kh * kw), ci_off
    p.x, p.cin, p.ldx = x.data_ptr(), cin, x.stride(0)
    p.n_img, p.hi, p.wi, p.hq, p.wq = n_img, hi, wi, hq, wq
    p.stride, p.kh, p.kw, p.pad_y, p.pad_x = stride, kh, kw, pad[0], pad[1]
    p.dy, p.cout, p.lddy = dy.data_ptr(), cout, dy.stride(0)
    p.dw = dw.data_ptr()
    if dbias is not None:
        assert dbias.is_contiguous() and dbias.numel() == cout
        p.dbias = dbias.data_ptr()
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


def groupnorm_silu_bwd(x, dy, batch, gamma, beta, partial, nchunk, scale_shift=None, groups=8, eps=1e-5, silu=True, dgb=None):
    """-> (dx, dgamma, dbeta, dscale_shift or None).  dgb: optional contiguous (2, C) tensor that receives [dgamma | dbeta]."""
    lib = _lib()
    _chk(lib, x, dy, gamma, beta, partial, scale_shift, dgb)
    rows, c = x.shape
    pixels = rows // batch
    assert x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(x)
    if dgb is None:
        dgb = torch.empty(2, c, dtype=torch.float32, device=x.device)
    assert dgb.is_contiguous() and dgb.numel() == 2 * c
    dss = torch.empty(batch, 2 * c, dtype=torch.float32, device=x.device) if scale_shift is not None else None
    nbytes = lib.lfdm_groupnorm_bwd_ws_bytes(batch, pixels, c)
    ws = _ws(nbytes, x)
    lib.check(lib.lfdm_groupnorm_silu_bwd_cl_f32(
        _p(x), _p(dy), _p(dx), batch, pixels, c, groups, _p(gamma), _p(beta), _p(scale_shift),
        scale_shift.stride(0) if scale_shift is not None else 0, eps, 1 if silu else 0, _p(partial), nchunk, _p(dgb),
        _p(dss), 2 * c, _p(ws), nbytes, _stream(lib)), "lfdm_groupnorm_silu_bwd_cl_f32")
    return dx, dgb[0], dgb[1], dss


def layernorm_bwd(x, dy, gamma, eps=1e-5, dgamma=None, dx_add=None):
    """-> (dx, dgamma).  dgamma: optional contiguous tensor of C elements that receives the gamma gradient; dx_add: optional second
    gradient of x (same shape) summed into dx by the kernel."""
    lib = _lib()
    _chk(lib, x, dy, gamma, dgamma, dx_add)
    rows, c = x.shape
    assert x.is_contiguous() and dy.is_contiguous() and (dx_add is None or (dx_add.is_contiguous() and dx_add.shape == x.shape))
    dx = torch.empty_like(x)
    if dgamma is None:
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    assert dgamma.is_contiguous() and dgamma.numel() == c
    nbytes = lib.lfdm_layernorm_bwd_ws_bytes(rows, c)
    ws = _ws(nbytes, x)
    lib.check(lib.lfdm_layernorm_bwd_add_cl_f32(_p(x), _p(dy), _p(dx_add), _p(dx), rows, c, _p(gamma), eps, _p(dgamma), _p(ws), nbytes,
                                                _stream(lib)), "lfdm_layernorm_bwd_add_cl_f32")
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


ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2
ML_ROWS, ML_KMAX = 16, 1024       # limits of one lfdm_multi_linear_* launch (csrc/train_linear.hip: ml_check)


def _ml_params(x, weights, act):
    rows, k = x.shape
    assert x.is_contiguous() and 0 < len(weights) <= ML_MAX
    p = MultiLinearParams()
    p.n_blocks, p.rows, p.k, p.act, p.x = len(weights), rows, k, act, x.data_ptr()
    for j, w in enumerate(weights):
        assert w.is_contiguous() and w.dim() == 2 and w.shape[1] == k
        p.w[j], p.n[j] = w.data_ptr(), w.shape[0]
    return p


def multi_linear(x, weights, biases, act=ACT_NONE):
    """[act(x) @ w.T + b for w, b in zip(weights, biases)] in ONE launch (lfdm_multi_linear_f32).  x (rows <= 16, k <= 1024)."""
    lib = _lib()
    _chk(lib, x, *weights, *[b for b in biases if b is not None])
    p = _ml_params(x, weights, act)
    ys = []
    for j, (w, b) in enumerate(zip(weights, biases)):
        y = torch.empty(x.shape[0], w.shape[0], dtype=torch.float32, device=x.device)
        ys.append(y)
        p.y[j] = y.data_ptr()
        if b is not None:
            assert b.is_contiguous() and b.numel() == w.shape[0]
            p.bias[j] = b.data_ptr()
    lib.check(lib.lfdm_multi_linear_f32(C.byref(p), _stream(lib)), "lfdm_multi_linear_f32")
    return ys


def multi_linear_bwd(x, weights, dys, act=ACT_NONE, dws=None, dbs=None, want_dx=True):
    """Backward of multi_linear.  dys[j]: (rows, n_j) or None (zero); dws[j] / dbs[j]: tensors to fill, or None (not wanted).
    -> dx (rows, k) or None."""
    lib = _lib()
    n = len(weights)
    dws = dws if dws is not None else [None] * n
    dbs = dbs if dbs is not None else [None] * n
    _chk(lib, x, *weights, *[t for t in list(dys) + list(dws) + list(dbs) if t is not None])
    p = _ml_params(x, weights, act)
    for j in range(n):
        for name, t, numel in (("dy", dys[j], x.shape[0] * weights[j].shape[0]), ("dw", dws[j], weights[j].numel()),
                               ("dbias", dbs[j], weights[j].shape[0])):
            if t is not None:
                assert t.is_contiguous() and t.numel() == numel, name
                getattr(p, name)[j] = t.data_ptr()
    dx = torch.empty_like(x) if want_dx else None
    if dx is not None:
        p.dx = dx.data_ptr()
    nbytes = lib.lfdm_multi_linear_bwd_ws_bytes(C.byref(p))
    ws = _ws(nbytes, x)
    lib.check(lib.lfdm_multi_linear_bwd_f32(C.byref(p), _p(ws), nbytes, _stream(lib)), "lfdm_multi_linear_bwd_f32")
    return dx


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


def relu_bwd(y, dy):
    """dy masked by y > 0 (lfdm_relu_bwd_f32): the backward of a ReLU fused into the producing convolution's epilogue."""
    lib = _lib()
    _chk(lib, y, dy)
    assert y.is_contiguous() and dy.is_contiguous() and y.shape == dy.shape and y.numel() % 4 == 0
    out = torch.empty_like(dy)
    lib.check(lib.lfdm_relu_bwd_f32(_p(y), _p(dy), _p(out), y.numel(), _stream(lib)), "lfdm_relu_bwd_f32")
    return out


POOL_AVG, POOL_SUM, POOL_UP, POOL_UP_QUARTER, POOL_MAX, POOL_MAX_BWD = range(6)


def pool2(x, n_img, h, w, mode, aux=None):
    """lfdm_pool2_cl_f32 on channels-last rows; (h, w) = the FINE resolution.  Modes 0 / 1 / 4: x (n*h*w, C) -> (n*h/2*w/2, C); modes
    2 / 3: x (n*h/2*w/2, C) -> (n*h*w, C); mode 5: x fine, aux = dy coarse -> dx fine."""
    lib = _lib()
    _chk(lib, x, aux)
    assert x.is_contiguous() and (aux is None or aux.is_contiguous())
    c = x.shape[1]
    fine, coarse = n_img * h * w, n_img * (h // 2) * (w // 2)
    assert x.shape[0] == (coarse if mode in (POOL_UP, POOL_UP_QUARTER) else fine)
    out = torch.empty(coarse if mode in (POOL_AVG, POOL_SUM, POOL_MAX) else fine, c, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_pool2_cl_f32(_p(x), _p(aux), _p(out), n_img, h, w, c, mode, _stream(lib)), "lfdm_pool2_cl_f32")
    return out


def im2col_cl(x, n_img, h, w, k, pad):
    """lfdm_im2col_cl_f32: x (n*h*w, c <= 16) rows -> (n*hq*wq, k*k*c), column tap*c + ch (stride 1, zero padding)."""
    lib = _lib()
    _chk(lib, x)
    assert x.stride(1) == 1 and x.shape[0] == n_img * h * w
    c = x.shape[1]
    hq, wq = h + 2 * pad - k + 1, w + 2 * pad - k + 1
    out = torch.empty(n_img * hq * wq, k * k * c, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_im2col_cl_f32(_p(x), _p(out), n_img, h, w, c, x.stride(0), k, pad, _stream(lib)), "lfdm_im2col_cl_f32")
    return out
