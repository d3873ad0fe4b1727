"""Python face of the LFAE stage-1 training glue kernels (include/lfdm_hip.h, ABI version 10; csrc/train_lfae.hip) and the
torch.autograd.Functions built on them - what connects the convolutions of ReconstructionModel.forward (LFAE/modules/model.py:141-217):
BatchNorm with batch statistics (+ ReLU), the anti-alias blur / image pyramid, deform_input + apply_optical, grid_sample with an
explicit grid, the 2x2 SVD.  Forward AND backward of each are liblfdm_hip.so kernels; torch.autograd is the tape.

Feature maps are NCHW tensors in channels-last MEMORY (their (N*H*W, C) row view is what the convolutions read); images are NCHW.
"""
import ctypes as C

import torch
from torch.autograd import Function

from . import ops, train_ops
from ._native import BN_TICKETS, BlurParams, GridSampleParams, WarpBwdParams
from .autograd import grad_out_pair
from .ops import _chk, _lib, _p, _stream

_STATE = {}


def _state(dev):
    """Per-device persistent words the kernels leave zeroed: BatchNorm tickets, the fixed-point scatter's max word + finalize ticket, and
    the 64-bit accumulator (grown on demand, always handed back zeroed by lfdm_fix_finalize_f32)."""
    key = str(dev)
    st = _STATE.get(key)
    if st is None:
        st = _STATE[key] = {"tickets": torch.zeros(BN_TICKETS, dtype=torch.int32, device=dev),
                            "amax": torch.zeros(4, dtype=torch.int32, device=dev), "fix": None}
    return st


def _fix_acc(dev, n):
    """The 64-bit scatter accumulator, grown on demand.  Growing it frees the old one, whose address a captured graph may hold
    (LFAETrainer.step_graphed): params.buffers_epoch is bumped so that such graphs are captured again instead of scatter-adding into - and
    zeroing - freed memory."""
    st = _state(dev)
    if st["fix"] is None or st["fix"].numel() < n:
        from .params import bump_buffers_epoch
        st["fix"] = torch.zeros(n, dtype=torch.int64, device=dev)
        bump_buffers_epoch()
    return st["fix"]


def _rows(x):
    """(N, C, H, W) in channels-last memory -> its (N*H*W, C) row view (a copy only if the memory format is something else)."""
    n, c, h, w = x.shape
    r = x.permute(0, 2, 3, 1)
    if not r.is_contiguous():
        r = r.contiguous()
    return r.reshape(n * h * w, c)


def _rows_ld(x):
    """Like _rows for kernels that take a row stride: a channel slice of a wider channels-last tensor (the halves of a torch.cat's
    gradient) is used where it lies - (N*H*W, C) with stride (ld, 1) - instead of being copied."""
    n, c, h, w = x.shape
    sn, sc, sh, sw = x.stride()
    if sc == 1 and sw % 4 == 0 and sw >= c and sh == w * sw and sn == h * sh and x.storage_offset() % 4 == 0:
        return x.as_strided((n * h * w, c), (sw, 1), x.storage_offset())
    return _rows(x)


def _from_rows(rows, n, h, w):
    return rows.view(n, h, w, rows.shape[1]).permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------------ BatchNorm (batch statistics) + ReLU
def batchnorm_train_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, relu, segments=1):
    """x: (rows, C) rows (stride(1) == 1), `segments` equal batches stacked along the rows, each with its own statistics.
    -> (y (rows, C), stat (segments, 2, C) = [mean | rstd] per segment)."""
    lib = _lib()
    _chk(lib, x, gamma, beta, running_mean, running_var)
    rows, c = x.shape
    assert x.stride(1) == 1 and rows % segments == 0
    seg_rows = rows // segments
    y = torch.empty(rows, c, dtype=torch.float32, device=x.device)
    stat = torch.empty(segments, 2, c, dtype=torch.float32, device=x.device)
    nbytes = lib.lfdm_batchnorm_train_ws_bytes(seg_rows, c, segments)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_batchnorm_train_fwd_cl_f32(_p(x), _p(y), seg_rows, c, segments, x.stride(0), c, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                                  float(momentum), float(eps), int(relu), _p(stat), _p(ws), nbytes, _p(_state(x.device)["tickets"]),
                                                  _stream(lib)), "lfdm_batchnorm_train_fwd_cl_f32")
    return y, stat


def batchnorm_train_bwd(x, dy, gamma, beta, stat, relu, dgamma=None, dbeta=None, dx_add=None):
    lib = _lib()
    _chk(lib, x, dy, gamma, beta, stat, dgamma, dbeta, dx_add)
    rows, c = x.shape
    segments = stat.shape[0] if stat.dim() == 3 else 1
    assert x.stride(1) == 1 and dy.stride(1) == 1 and dy.shape == x.shape and rows % segments == 0
    seg_rows = rows // segments
    dx = torch.empty(rows, c, dtype=torch.float32, device=x.device)
    nbytes = lib.lfdm_batchnorm_train_ws_bytes(seg_rows, c, segments)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_batchnorm_train_bwd_cl_f32(_p(x), _p(dy), _p(dx), seg_rows, c, segments, x.stride(0), dy.stride(0), c, _p(dx_add),
                                                  0 if dx_add is None else dx_add.stride(0), _p(gamma), _p(beta), _p(stat),
                                                  int(relu), _p(dgamma), _p(dbeta), _p(ws), nbytes, _p(_state(x.device)["tickets"]), _stream(lib)),
              "lfdm_batchnorm_train_bwd_cl_f32")
    return dx


class BatchNormReLU(Function):
    """nn.BatchNorm2d (training mode: batch statistics, running statistics updated in place) followed by ReLU when relu=True, on an NCHW
    tensor in channels-last memory (LFAE/modules/util.py:84-90, 108-112, 128-133, 146-150).  segments = S: the batch is S equal
    sub-batches, each normalised with its own statistics, the running statistics updated S times in order - S module calls as one.
    fork=True -> (y, x): x itself as a second output for the block's skip path (ResBlock2d: out += x, util.py:84-92); the skip's gradient
    then arrives here together with y's and the backward kernel stores the sum (like autograd.LayerNormCL / ConvCL)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, relu, segments=1, fork=False):
        n, c, h, w = x.shape
        assert n % segments == 0, "BatchNormReLU: the batch must split evenly into the segments"
        xr = _rows_ld(x.detach())
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        y, stat = batchnorm_train_fwd(xr, g, b, running_mean, running_var, momentum, eps, relu, segments)
        ctx.save_for_backward(xr, gamma, beta, stat)
        ctx.meta = (n, h, w, relu)
        return (_from_rows(y, n, h, w), x) if fork else _from_rows(y, n, h, w)

    @staticmethod
    def backward(ctx, dy, dpass=None):
        xr, gamma, beta, stat = ctx.saved_tensors
        n, h, w, relu = ctx.meta
        need = ctx.needs_input_grad
        dg = db = None
        if need[1] or need[2]:
            _, dg, db = grad_out_pair(gamma, beta)
        dx = batchnorm_train_bwd(xr, _rows_ld(dy), gamma.detach().contiguous(), beta.detach().contiguous(), stat, relu, dg, db,
                                 None if dpass is None else _rows_ld(dpass))
        return _from_rows(dx, n, h, w), dg, db, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ blur + subsample
def _blur_params(x_like, shape, weight, stride, pad_lo, pad_hi, out_like, c_store, scale, bias):
    n, c, h, w = shape
    p = BlurParams()
    p.wgt = weight.data_ptr()
    p.xs_n, p.xs_c, p.xs_h, p.xs_w = x_like.stride()
    p.os_n, p.os_c, p.os_h, p.os_w = out_like.stride()
    p.n_img, p.channels, p.c_store, p.h, p.w = n, c, c_store, h, w
    p.k, p.pad_lo, p.pad_hi, p.stride = weight.shape[-1], pad_lo, pad_hi, stride
    p.scale = None if scale is None else scale.data_ptr()
    p.bias = None if bias is None else bias.data_ptr()
    return p


def blur_geometry(h, w, k, stride):
    """(pad_lo, pad_hi, ho, wo) of AntiAliasInterpolation2d (util.py:236-246, 259-261: pad ka / kb, conv, [::s])."""
    ka = k // 2
    kb = ka - 1 if k % 2 == 0 else ka
    hf, wf = h + ka + kb - k + 1, w + ka + kb - k + 1
    return ka, kb, (hf + stride - 1) // stride, (wf + stride - 1) // stride


def blur_down(x, weight, stride, *, rows4=False, scale=None, bias=None):
    """x: (N, C, H, W) with ANY strides; weight (C, 1, k, k) / (C, k, k).  -> (N, C, ho, wo) contiguous, or with rows4=True an
    (N, 4, ho, wo) tensor in channels-last memory whose channels >= C are zero (the rows a convolution reads; C <= 4)."""
    lib = _lib()
    wgt = weight.reshape(weight.shape[0], weight.shape[-2], weight.shape[-1]).contiguous()
    _chk(lib, x, wgt, scale, bias)
    n, c, h, w = x.shape
    pad_lo, pad_hi, ho, wo = blur_geometry(h, w, wgt.shape[-1], stride)
    if rows4:
        assert c <= 4
        out = torch.empty(n, ho, wo, 4, dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)
    else:
        out = torch.empty(n, c, ho, wo, dtype=torch.float32, device=x.device)
    p = _blur_params(x, x.shape, wgt, stride, pad_lo, pad_hi, out, out.shape[1], scale, bias)
    p.x, p.out = x.data_ptr(), out.data_ptr()
    lib.check(lib.lfdm_blur_down_fwd_f32(C.byref(p), _stream(lib)), "lfdm_blur_down_fwd_f32")
    return out


def blur_down_bwd(dy, weight, x_shape, stride, *, scale=None):
    """dy: gradient of blur_down's output (any strides; channels >= C ignored).  -> dx (N, C, H, W) contiguous."""
    lib = _lib()
    wgt = weight.reshape(weight.shape[0], weight.shape[-2], weight.shape[-1]).contiguous()
    _chk(lib, dy, wgt, scale)
    n, c, h, w = x_shape
    pad_lo, pad_hi, _, _ = blur_geometry(h, w, wgt.shape[-1], stride)
    dx = torch.empty(n, c, h, w, dtype=torch.float32, device=dy.device)
    p = _blur_params(dx, x_shape, wgt, stride, pad_lo, pad_hi, dy, dy.shape[1], scale, None)
    p.dy, p.dx = dy.data_ptr(), dx.data_ptr()
    lib.check(lib.lfdm_blur_down_bwd_f32(C.byref(p), _stream(lib)), "lfdm_blur_down_bwd_f32")
    return dx


class BlurDown(Function):
    """AntiAliasInterpolation2d / one ImagePyramide level (+ optional per-channel affine: the VGG input normalisation)."""

    @staticmethod
    def forward(ctx, x, weight, stride, rows4, scale, bias):
        ctx.save_for_backward(weight, scale)
        ctx.meta = (tuple(x.shape), stride)
        return blur_down(x.detach(), weight, stride, rows4=rows4, scale=scale, bias=bias)

    @staticmethod
    def backward(ctx, dy):
        weight, scale = ctx.saved_tensors
        shape, stride = ctx.meta
        return blur_down_bwd(dy, weight, shape, stride, scale=scale), None, None, None, None, None


# ------------------------------------------------------------------------------------------------ deform_input + apply_optical
def _maps_planar(flow, occ):
    """optical_flow (N, fh, fw, 2) [+ occlusion (N, 1, fh, fw)] -> one contiguous (N, 2 | 3, fh, fw) tensor of planes."""
    planes = flow.permute(0, 3, 1, 2)
    if occ is not None:
        planes = torch.cat((planes, occ), dim=1)
    return planes.contiguous()


def _warp_bwd_common(p, maps, n, h, w, c):
    fh, fw = maps.shape[2], maps.shape[3]
    p.flow_x, p.flow_y = maps.data_ptr(), maps.data_ptr() + 4 * fh * fw
    p.occ = (maps.data_ptr() + 8 * fh * fw) if maps.shape[1] == 3 else None
    p.n_img, p.h, p.w, p.c, p.fh, p.fw = n, h, w, c, fh, fw
    p.fsn = maps.shape[1] * fh * fw
    return fh, fw


def _fold_dmaps(lib, dmaps, n, h, w, fh, fw, planes):
    """dmaps (N, 3, h, w) at output resolution -> (N, planes, fh, fw) gradient of the map tensor."""
    if (fh, fw) == (h, w):
        return dmaps[:, :planes]
    low = torch.empty(n, 3, fh, fw, dtype=torch.float32, device=dmaps.device)
    lib.check(lib.lfdm_resize_adjoint_f32(_p(dmaps), _p(low), n * 3, h, w, fh, fw, _stream(lib)), "lfdm_resize_adjoint_f32")
    return low[:, :planes]


class ApplyOpticalCL(Function):
    """out = grid_sample(src, resize(flow)) * resize(occ) + prev * (1 - resize(occ)) on feature maps (NCHW tensors in channels-last
    memory): Generator.deform_input + apply_optical (generator.py:59-88).  maps: (N, 2 | 3, fh, fw) planes [flow_x, flow_y(, occ)].
    Forward = lfdm_warp_cl_f32 (the sampling path's kernel), backward = lfdm_warp_bwd_f32 (+ fixed-point finalize, resize adjoint)."""

    @staticmethod
    def forward(ctx, src, prev, maps):
        n, c, h, w = src.shape
        sr = _rows(src.detach())
        pr = None if prev is None else _rows(prev.detach())
        m = maps.detach().contiguous()
        fh, fw = m.shape[2], m.shape[3]
        occ = m[:, 2] if m.shape[1] == 3 else None
        out = ops.warp_cl(sr, n, 1, h, w, m[:, 0], m[:, 1], occ, fh, fw, m.shape[1] * fh * fw, 0, prev=pr)
        ctx.save_for_backward(sr, pr, m)
        ctx.meta = (n, c, h, w)
        return _from_rows(out, n, h, w)

    @staticmethod
    def backward(ctx, dout):
        sr, pr, m = ctx.saved_tensors
        n, c, h, w = ctx.meta
        need = ctx.needs_input_grad
        lib = _lib()
        dr = _rows(dout)
        dev = dr.device
        st = _state(dev)
        p = WarpBwdParams()
        fh, fw = _warp_bwd_common(p, m, n, h, w, c)
        p.layout_cl, p.n_div = 1, 1
        p.src, p.ld_src, p.dout, p.ld_dout = sr.data_ptr(), sr.stride(0), dr.data_ptr(), dr.stride(0)
        dprev = None
        if pr is not None:
            p.prev, p.ld_prev = pr.data_ptr(), pr.stride(0)
            if need[1]:
                dprev = torch.empty(n * h * w, c, dtype=torch.float32, device=dev)
                p.dprev, p.ld_dprev = dprev.data_ptr(), c
        dmaps = torch.empty(n, 3, h, w, dtype=torch.float32, device=dev)
        p.dmaps = dmaps.data_ptr()
        acc = None
        if need[0]:
            acc = _fix_acc(dev, n * h * w * c)
            p.dsrc_fix, p.amax_bits = acc.data_ptr(), st["amax"].data_ptr()
            lib.check(lib.lfdm_absmax_f32(_p(dr), n * h * w, c, dr.stride(0), _p(st["amax"]), _stream(lib)), "lfdm_absmax_f32")
        lib.check(lib.lfdm_warp_bwd_f32(C.byref(p), _stream(lib)), "lfdm_warp_bwd_f32")
        dsrc = None
        if need[0]:
            dsrc = torch.empty(n * h * w, c, dtype=torch.float32, device=dev)
            lib.check(lib.lfdm_fix_finalize_f32(_p(acc), _p(dsrc), n * h * w, c, c, _p(st["amax"]), 4 * h * w,
                                                C.c_void_p(st["amax"].data_ptr() + 4), _stream(lib)), "lfdm_fix_finalize_f32")
            dsrc = _from_rows(dsrc, n, h, w)
        dm = _fold_dmaps(lib, dmaps, n, h, w, fh, fw, m.shape[1]) if need[2] else None
        return dsrc, (None if dprev is None else _from_rows(dprev, n, h, w)), dm


class ApplyOpticalImage(Function):
    """The same blend for an image-like tensor with few channels and any strides (generator.py:126-128: the RGB prediction blended with
    the warped source image).  src gets no gradient (it is an input image); prev and the maps do."""

    @staticmethod
    def forward(ctx, src, prev, maps):
        n, c, h, w = src.shape
        m = maps.detach().contiguous()
        fh, fw = m.shape[2], m.shape[3]
        s = src.detach().contiguous()
        pv = prev.detach()
        ld = pv.stride(3)
        if pv.stride(1) == 1 and pv.stride(2) == ld * w and pv.stride(0) == ld * h * w:      # channels of channels-last rows (a convolution's output)
            kw = dict(prev=torch.as_strided(pv, (n * h * w, c), (ld, 1)), prev_is_cl=True)
        else:
            kw = dict(prev=pv.contiguous())
        out = ops.warp_planar(s, 1, m[:, 0], m[:, 1], m[:, 2] if m.shape[1] == 3 else None, fh, fw, m.shape[1] * fh * fw, 0, **kw)
        ctx.save_for_backward(s, pv, m)
        ctx.meta = (n, c, h, w)
        return out.view(n, c, h, w)

    @staticmethod
    def backward(ctx, dout):
        s, pv, m = ctx.saved_tensors
        n, c, h, w = ctx.meta
        lib = _lib()
        dev = dout.device
        p = WarpBwdParams()
        fh, fw = _warp_bwd_common(p, m, n, h, w, c)
        p.layout_cl, p.n_div = 0, 1
        p.src, p.dout, p.prev = s.data_ptr(), dout.data_ptr(), pv.data_ptr()
        p.ss_n, p.ss_c, p.ss_h, p.ss_w = s.stride()
        p.ds_n, p.ds_c, p.ds_h, p.ds_w = dout.stride()
        p.ps_n, p.ps_c, p.ps_h, p.ps_w = pv.stride()
        dprev = torch.empty(n, c, h, w, dtype=torch.float32, device=dev)
        p.dprev = dprev.data_ptr()
        p.dps_n, p.dps_c, p.dps_h, p.dps_w = dprev.stride()
        dmaps = torch.empty(n, 3, h, w, dtype=torch.float32, device=dev)
        p.dmaps = dmaps.data_ptr()
        lib.check(lib.lfdm_warp_bwd_f32(C.byref(p), _stream(lib)), "lfdm_warp_bwd_f32")
        return None, dprev, _fold_dmaps(lib, dmaps, n, h, w, fh, fw, m.shape[1])


# ------------------------------------------------------------------------------------------------ grid_sample with an explicit grid
def _gs_params(x, grid, out_like, n_div, pad_mode):
    n, ho, wo, _ = grid.shape
    p = GridSampleParams()
    p.x, p.grid = x.data_ptr(), grid.data_ptr()
    p.xs_n, p.xs_c, p.xs_h, p.xs_w = x.stride()
    p.os_n, p.os_c, p.os_h, p.os_w = out_like.stride()
    p.n_img, p.channels, p.h, p.w, p.ho, p.wo, p.n_div, p.pad_mode = n, x.shape[1], x.shape[2], x.shape[3], ho, wo, n_div, pad_mode
    return p


class GridSample(Function):
    """F.grid_sample(x.repeat_interleave(n_div), grid, mode='bilinear', padding_mode=zeros|reflection, align_corners=False) for image-like x
    (few channels, any strides); x is shared by n_div consecutive grids.  Gradient w.r.t. the grid only (zeros padding)."""

    @staticmethod
    def forward(ctx, x, grid, n_div, reflection):
        lib = _lib()
        xd, gd = x.detach(), grid.detach().contiguous()
        _chk(lib, xd, gd)
        n, ho, wo, _ = gd.shape
        assert n == xd.shape[0] * n_div
        out = torch.empty(n, xd.shape[1], ho, wo, dtype=torch.float32, device=xd.device)
        p = _gs_params(xd, gd, out, n_div, 1 if reflection else 0)
        p.out = out.data_ptr()
        lib.check(lib.lfdm_grid_sample_fwd_f32(C.byref(p), _stream(lib)), "lfdm_grid_sample_fwd_f32")
        ctx.save_for_backward(xd, gd)
        ctx.meta = (n_div, reflection)
        return out

    @staticmethod
    def backward(ctx, dout):
        xd, gd = ctx.saved_tensors
        n_div, reflection = ctx.meta
        if not ctx.needs_input_grad[1]:
            return None, None, None, None
        if reflection:
            raise NotImplementedError("gradient of the reflection-padded warp (model.py:118-122 needs none)")
        lib = _lib()
        dgrid = torch.empty_like(gd)
        p = _gs_params(xd, gd, dout, n_div, 0)
        p.dout, p.dgrid = dout.data_ptr(), dgrid.data_ptr()
        lib.check(lib.lfdm_grid_sample_bwd_f32(C.byref(p), _stream(lib)), "lfdm_grid_sample_bwd_f32")
        return None, dgrid, None, None


# ------------------------------------------------------------------------------------------------ 2x2 SVD
class Svd2x2Sym(Function):
    """(U, S) = torch.svd(covar)[:2] for symmetric positive semi-definite (n, 2, 2) matrices with LAPACK's sign convention on the device
    (lfdm_svd2x2_sym_f32; the reference moves the matrices to the host for it, region_predictor.py:21-25), analytic backward."""

    @staticmethod
    def forward(ctx, covar):
        cv = covar.detach()
        u, s = ops.svd2x2_sym(cv[:, 0, 0], cv[:, 0, 1], cv[:, 1, 1])
        ctx.save_for_backward(u, s)
        ctx.mark_non_differentiable()
        return u, s

    @staticmethod
    def backward(ctx, gu, gs):
        u, s = ctx.saved_tensors
        lib = _lib()
        gu = None if gu is None else gu.contiguous()
        gs = None if gs is None else gs.contiguous()
        ga = torch.empty_like(u)
        lib.check(lib.lfdm_svd2x2_sym_bwd_f32(_p(u), _p(s), _p(gu), _p(gs), _p(ga), u.shape[0], _stream(lib)), "lfdm_svd2x2_sym_bwd_f32")
        return ga


# ------------------------------------------------------------------------------------------------ 2x2 pooling family, L1 mean
class Pool2(Function):
    """kind 'avg' (F.avg_pool2d(x, 2)), 'max' (F.max_pool2d(x, 2)) or 'up' (F.interpolate(x, scale_factor=2), nearest) on an NCHW tensor
    in channels-last memory with even height / width; forward and backward are lfdm_pool2_cl_f32 launches."""

    @staticmethod
    def forward(ctx, x, kind):
        n, c, h, w = x.shape
        xr = _rows(x.detach())
        if kind == "up":
            out = train_ops.pool2(xr, n, 2 * h, 2 * w, train_ops.POOL_UP)
            ctx.meta = (kind, n, 2 * h, 2 * w)
            return _from_rows(out, n, 2 * h, 2 * w)
        out = train_ops.pool2(xr, n, h, w, train_ops.POOL_AVG if kind == "avg" else train_ops.POOL_MAX)
        ctx.meta = (kind, n, h, w)
        if kind == "max":
            ctx.save_for_backward(xr)
        return _from_rows(out, n, h // 2, w // 2)

    @staticmethod
    def backward(ctx, dy):
        kind, n, h, w = ctx.meta                  # (h, w): the fine resolution
        dr = _rows(dy)
        if kind == "up":
            return _from_rows(train_ops.pool2(dr, n, h, w, train_ops.POOL_SUM), n, h // 2, w // 2), None
        if kind == "avg":
            return _from_rows(train_ops.pool2(dr, n, h, w, train_ops.POOL_UP_QUARTER), n, h, w), None
        (xr,) = ctx.saved_tensors
        return _from_rows(train_ops.pool2(xr, n, h, w, train_ops.POOL_MAX_BWD, aux=dr), n, h, w), None


def pool2(x, kind):
    """Native when the geometry allows (even size, channels % 4 == 0), else the ATen op."""
    import torch.nn.functional as F
    n, c, h, w = x.shape
    if c % 4 == 0 and (kind == "up" or (h % 2 == 0 and w % 2 == 0)):
        return Pool2.apply(x, kind)
    return {"avg": lambda: F.avg_pool2d(x, 2), "max": lambda: F.max_pool2d(x, 2), "up": lambda: F.interpolate(x, scale_factor=2)}[kind]()


class L1Mean(Function):
    """weight * |x - y|.mean() as a (1,) tensor (the perceptual loss terms, model.py:189-195); y gets no gradient."""

    @staticmethod
    def forward(ctx, x, y, weight):
        lib = _lib()
        xd, yd = x.detach(), y.detach()
        xd = xd if _dense(xd) else xd.contiguous()
        yd = yd if (_dense(yd) and yd.stride() == xd.stride()) else yd.contiguous(memory_format=_fmt(xd))
        if yd.stride() != xd.stride():
            xd, yd = xd.contiguous(), yd.contiguous()
        _chk(lib, xd, yd)
        n = xd.numel()
        assert n % 4 == 0
        st = _state(xd.device)
        out = torch.empty(1, dtype=torch.float32, device=xd.device)
        ws = torch.empty(2048, dtype=torch.float32, device=xd.device)
        lib.check(lib.lfdm_l1_mean_fwd_f32(_p(xd), _p(yd), n, float(weight), _p(out), _p(ws), 8192, C.c_void_p(st["amax"].data_ptr() + 8),
                                           _stream(lib)), "lfdm_l1_mean_fwd_f32")
        ctx.save_for_backward(xd, yd)
        ctx.weight = float(weight)
        return out

    @staticmethod
    def backward(ctx, gout):
        xd, yd = ctx.saved_tensors
        lib = _lib()
        dx = torch.empty_like(xd)
        lib.check(lib.lfdm_l1_mean_bwd_f32(_p(xd), _p(yd), xd.numel(), ctx.weight, _p(gout.contiguous()), _p(dx), _stream(lib)),
                  "lfdm_l1_mean_bwd_f32")
        return dx, None, None


def _dense(t):
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


def _fmt(t):
    return torch.contiguous_format if t.is_contiguous() else torch.channels_last
