"""torch.autograd.Function wrappers around the native forward + backward kernels (include/lfdm_hip.h), so that the
reference training scripts' `loss.backward()` (DM/modules/video_flow_diffusion_model.py:181-188) runs through
liblfdm_hip.so.  torch.autograd is used as the tape (saved tensors, gradient accumulation of shared tensors); every
activation-sized computation - forward and backward - is a HIP kernel.  Activations are channels-last rows
(N*H*W, C), frames n = b*T + t, exactly as on the sampling path.

Data gradients of convolutions reuse the forward implicit-GEMM kernels with re-packed weights:
  Conv (stride 1)        -> same-geometry conv with W^T flipped, pad k-1-p
  Conv k4 s2 p1          -> the four-parity transposed convolution (Downsample, :166-167)
  ConvTranspose k4 s2 p1 -> stride-2 conv (Upsample, :156-158)
"""
import torch
from torch.autograd import Function

from . import ops, train_ops


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def grad_out(param, like=None):
    """The tensor a native backward kernel writes the gradient of `param` into.  When `param` lives in optim.Adam's flat storage and
    this is its first gradient of the step (param.grad is None, i.e. zero_grad(set_to_none=True) - what the training scripts do), that
    is the parameter's slot of the flat gradient buffer: autograd adopts the returned tensor as param.grad without a copy and
    Adam.stage_grads / GradAllReduce find the gradient already in place (before: one staging copy per parameter per step).
    Otherwise (second use of a shared parameter, accumulation over several backward passes, any other optimizer) a new tensor."""
    slot = getattr(param, "_lfdm_grad_slot", None)
    if (slot is None or param.grad is not None or param._lfdm_grad_slot_busy or slot.shape != param.shape or
            slot.device != param.device):
        return torch.empty(param.shape, dtype=torch.float32, device=param.device if like is None else like.device)
    param._lfdm_grad_slot_busy = True
    return slot.view(param.shape)


def grad_out_pair(pa, pb):
    """(2, C) tensor for kernels that write [d pa | d pb] as one block (GroupNorm: [dgamma | dbeta]): the two slots when they are free and
    adjacent in the flat buffer (norm.weight / norm.bias are registered back to back), else a new tensor.  -> (block, grad_a, grad_b)."""
    sa, sb = getattr(pa, "_lfdm_grad_slot", None), getattr(pb, "_lfdm_grad_slot", None)
    c = pa.numel()
    if (sa is not None and sb is not None and pa.grad is None and pb.grad is None and not pa._lfdm_grad_slot_busy and
            not pb._lfdm_grad_slot_busy and pb.numel() == c and sa._base is not None and sa._base is sb._base and
            sb.storage_offset() == sa.storage_offset() + c and sa.device == pa.device):
        pa._lfdm_grad_slot_busy = pb._lfdm_grad_slot_busy = True
        blk = sa._base[sa.storage_offset():sa.storage_offset() + 2 * c].view(2, c)
        return blk, sa.view(pa.shape), sb.view(pb.shape)
    blk = torch.empty(2, c, dtype=torch.float32, device=pa.device)
    return blk, blk[0].view(pa.shape), blk[1].view(pb.shape)


def _conv(x0, w4_direct, cout, kh, kw, n_img, hi, wi, *, weight_wino=None, **kw_):
    """ops.conv2d_cl that skips the direct-form filter pack (a handful of torch kernels per call, every step) whenever the
    library confirms the Winograd schedule; w4_direct: ((O, I, kh, kw) weight, pack mode of ops.pack_conv_weight_dev) for the
    fallback - one launch."""
    if weight_wino is not None:
        try:
            return ops.conv2d_cl(x0, None, cout, kh, kw, n_img, hi, wi, weight_wino=weight_wino, **kw_)
        except ops.WinogradUnavailable:
            pass
    return ops.conv2d_cl(x0, ops.pack_conv_weight_dev(*w4_direct), cout, kh, kw, n_img, hi, wi, weight_wino=weight_wino, **kw_)


_PACKS = {}


def _view_geometry(weight, w4):
    """(storage offset, shape, strides) of w4 inside weight's storage, or the view itself when it is a copy (a non-contiguous parameter):
    the cache must not pin a deleted model's parameter storage through a strong reference to a view of it."""
    if w4.untyped_storage().data_ptr() == weight.untyped_storage().data_ptr():
        return (w4.storage_offset(), tuple(w4.shape), tuple(w4.stride()))
    return w4


def _live_view(weight, geo):
    return geo if torch.is_tensor(geo) else weight.detach().as_strided(geo[1], geo[2], geo[0])


def _purge_dead_packs():
    for k in [k for k, v in _PACKS.items() if v[0]() is None]:
        del _PACKS[k]


def _pack_wino(weight, w4, dgrad=False, part=None):
    """ops.pack_wino_weight(w4) with a cache over the life of the weights: the frozen VGG-19 of LFAE stage-1 training convolves with the
    same 13 filters 8 times forward and 4 times backward per step, the region predictor runs three times per step - each use re-packed
    (1 290 pack launches per 5 steps, 4 % of a step).  Only leaf tensors (parameters) are cached, by object identity (a weak reference:
    an address can be re-used by another tensor); an entry is valid while neither torch (`_version`) nor a raw-pointer optimizer step
    (params.weights_epoch) has written the tensor.  weight: the tensor the caller passed to the Function; w4: the view of it to pack.
    A stale entry of the same geometry is refilled IN PLACE and its tensor is never rebound: a captured hipGraph (LFAETrainer.step_graphed)
    holds the pack's raw address, and an eager / eval forward, a load_state_dict or a second batch shape between two replays must not free
    the memory it reads.  A pack that has to be allocated anew bumps params.buffers_epoch (captured graphs then re-capture)."""
    # A cache entry belongs to a PERSISTENT tensor object: a parameter, or the zero-padded buffer a frozen parameter was copied into
    # (lfae_train._PadParam hands out a fresh alias of that buffer per call and names the buffer in `_lfdm_pack_owner`).  Anything else
    # (a non-leaf, a transient alias) is packed per call.
    owner = getattr(weight, "_lfdm_pack_owner", None)
    if owner is not None and weight.is_leaf and owner.data_ptr() == weight.data_ptr():
        weight = owner
    elif not weight.is_leaf or not (weight.requires_grad or isinstance(weight, torch.nn.Parameter)):
        return ops.pack_wino_weight(w4, dgrad=dgrad)          # (a frozen non-parameter leaf: somebody's detached alias, a new object per call)
    import weakref
    from .params import bump_buffers_epoch, weights_epoch
    key = (id(weight), dgrad, part)
    # (a frozen tensor is in no optimizer: only torch writes - load_state_dict, .to() - can change it)
    tag = (weight._version, weights_epoch() if weight.requires_grad else -1, w4.data_ptr(), tuple(w4.shape))
    hit = _PACKS.get(key)
    if hit is not None and hit[0]() is weight:
        if hit[1] == tag:
            return hit[2]
        k, n = (w4.shape[0], w4.shape[1]) if dgrad else (w4.shape[1], w4.shape[0])
        packed = hit[2]
        if packed.device == w4.device and tuple(packed.shape) == (16, k // 16, (n + 31) // 32 * 32, 16):
            ops.pack_wino_weight(w4, dgrad=dgrad, out=packed)
            _PACKS[key] = (hit[0], tag, packed, _view_geometry(weight, w4))
            return packed
    if len(_PACKS) > 2048:
        _purge_dead_packs()
    packed = ops.pack_wino_weight(w4, dgrad=dgrad)
    _PACKS[key] = (weakref.ref(weight), tag, packed, _view_geometry(weight, w4))
    bump_buffers_epoch()
    return packed


def repack_stale():
    """Re-pack every cached Winograd filter of a TRAINABLE weight whose pack is older than the last optimizer step - forward and data-gradient
    forms, all in ONE launch into their existing tensors (ops.pack_wino_weights_multi).  Call at the start of a training step's forward:
    the packs a step needs are the packs the previous step used, so after the first step the ~100 (DM) / ~80 (LFAE) per-filter pack launches
    of a step become one.  Entries whose weight torch itself has rewritten in place (load_state_dict, copy_) are rebuilt here as well;
    entries of deleted weights are dropped."""
    from .params import weights_epoch
    epoch = weights_epoch()
    jobs, keys, dead = [], [], False
    for key, (ref, tag, packed, geo) in _PACKS.items():
        weight = ref()
        if weight is None:
            dead = True
            continue
        if tag[0] == weight._version and (not weight.requires_grad or tag[1] == epoch):
            continue
        # stale by the optimizer's epoch OR because torch rewrote the tensor (load_state_dict / copy_: `_version`).  The second kind used to
        # be left to the per-call path - which a REPLAYED graph never runs: a load_state_dict between two replays left the captured step
        # convolving with the filters of before (tests/test_lfae_train.py::test_graphed_step_survives_foreign_work_between_replays).
        if torch.is_tensor(geo):              # (the packed view was a COPY of a non-contiguous parameter: only the per-call path can rebuild it)
            continue
        w4 = _live_view(weight, geo)
        if w4.device != packed.device or tuple(w4.shape) != tag[3]:
            continue
        jobs.append((w4, packed, key[1]))
        keys.append((key, w4.data_ptr(), weight._version))
    if dead:
        _purge_dead_packs()
    if not jobs:
        return 0
    ops.pack_wino_weights_multi(jobs)
    for key, ptr, version in keys:
        ref, tag, packed, geo = _PACKS[key]
        _PACKS[key] = (ref, (version, epoch if ref().requires_grad else -1, ptr, tag[3]), packed, geo)
    return len(jobs)


def _wino_ok(kh, kw, stride, pad, hi, wi, *chans):
    """Geometry the Winograd F(2x2,3x3) schedule covers (lfdm_conv_params.weight_wino); chans = reduction-channel counts."""
    return (kh == 3 and kw == 3 and stride == 1 and tuple(pad) == (1, 1) and hi % 2 == 0 and wi % 2 == 0 and
            all(c % 16 == 0 for c in chans))


def _thin_wgrad_route(kind, x1, stride, pad, kh, kw, cin, cout, hi, wi, hq, wq):
    """'in' / 'out' when a k x k convolution (k >= 3, stride 1, square, symmetric padding) has <= 16 input (resp. output) channels: its weight
    gradient then runs as ONE 1x1 weight-gradient GEMM over an im2col of the thin side (train_ops.im2col_cl) instead of k*k per-tap GEMMs
    that each pad the thin side to the kernel's 64-wide tile (the generator's 7x7 RGB convolutions: 16x the work)."""
    if kind != "conv" or x1 is not None or stride != 1 or kh != kw or kh < 3 or pad[0] != pad[1]:
        return None
    if cin % 4 == 0 and cin <= 16:
        return "in"
    if cout % 4 == 0 and cout <= 16 and 2 * pad[0] == kh - 1:
        return "out"
    return None


def _thin_wgrad(route, x, dy, n_img, hi, wi, hq, wq, k, pad, cin, cout):
    """-> dw (cout, cin, k, k).  'in': dW[(tap, ci)][co] = im2col(x)^T dy.  'out': with r' = the INPUT pixel, dW[tap][ci][co] =
    sum_r' x[r'][ci] dy[r' - tap + pad][co] = x^T im2col(dy, pad' = k - 1 - pad) at the mirrored tap."""
    if route == "in":
        col = train_ops.im2col_cl(x, n_img, hi, wi, k, pad)                                   # (n*hq*wq, k*k*cin)
        dwt = train_ops.conv_wgrad(col, dy, n_img, hq, wq, hq, wq, 1, 1, pad=(0, 0))          # (1, k*k*cin, cout)
        return dwt.view(k, k, cin, cout).permute(3, 2, 0, 1)
    col = train_ops.im2col_cl(dy, n_img, hq, wq, k, k - 1 - pad)                              # (n*hi*wi, k*k*cout)
    dwt = train_ops.conv_wgrad(x, col, n_img, hi, wi, hi, wi, 1, 1, pad=(0, 0))               # (1, cin, k*k*cout)
    return dwt.view(cin, k, k, cout).flip(1, 2).permute(3, 0, 1, 2)


class ConvCL(Function):
    """y = conv(cat(x0, x1), weight) + bias (+ residual).  weight in the reference layout (Cout, Cin, [1,] kh, kw)
    (or ConvTranspose (Cin, Cout, [1,] 4, 4) when geom['kind'] == 'deconv').  geom: n_img, hi, wi, stride, pad.
    geom['fork'] (stride-1 convolutions): -> (y, x0[, x1]) - the inputs themselves as further outputs, for the second consumer of the same
    tensors (ResnetBlock: block1 and the skip / res_conv read the same x, :214-238).  The gradients of both consumers then arrive at this
    node together and the data-gradient convolution adds the other one in its epilogue (the `residual` operand of the forward kernels)
    instead of the autograd engine launching an add per block."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, residual, geom):
        kind = geom.get("kind", "conv")
        n_img, hi, wi = geom["n_img"], geom["hi"], geom["wi"]
        w4 = weight.detach().reshape(weight.shape[0], weight.shape[1], weight.shape[-2], weight.shape[-1]) \
            if weight.dim() > 2 else weight.detach()[:, :, None, None]
        kh, kw = w4.shape[2], w4.shape[3]
        b = None if bias is None else _c(bias.detach())
        res = None if residual is None else _c(residual.detach())
        if kind == "conv":
            stride = geom.get("stride", 1)
            pad = geom.get("pad", (kh // 2, kw // 2))
            ww = None
            if _wino_ok(kh, kw, stride, pad, hi, wi, x0.shape[1], 0 if x1 is None else x1.shape[1]):
                ww = _pack_wino(weight, _c(w4))
            if (w4.shape[0] <= 4 and x1 is None and res is None and stride == 1 and kh == kw and kh % 2 == 1 and 3 <= kh <= 7 and
                    tuple(pad) == (kh // 2, kh // 2) and x0.shape[1] % 16 == 0 and not geom.get("relu")):
                # <= 4 output channels (the generator's 7x7 RGB projection, generator.py:56): the 4x4x1-MFMA kernel of the decode path -
                # every multiply useful - instead of a 32-column tile that is 7/8 padding (0.98 -> 0.25 ms at 32 frames of 128x128)
                wsm, bsm = ops.pack_smalln_weight(w4, b)
                y = ops.conv2d_smalln_cl(_c(x0.detach()), wsm, bsm, 4, kh, n_img, hi, wi)
                y = y if w4.shape[0] == 4 else y[:, :w4.shape[0]].contiguous()
            else:
                y = _conv(_c(x0.detach()), (w4, 0), w4.shape[0], kh, kw, n_img, hi, wi,
                          src1=None if x1 is None else _c(x1.detach()), bias=b, residual=res, stride=stride, pad=pad,
                          weight_wino=ww, act=ops.ACT_RELU if geom.get("relu") else ops.ACT_NONE)
            hq = (hi + 2 * pad[0] - kh) // stride + 1
            wq = (wi + 2 * pad[1] - kw) // stride + 1
        else:
            assert x1 is None and residual is None and kh == 4 and kw == 4
            y = ops.deconv4x4s2_cl(_c(x0.detach()), ops.pack_conv_weight_dev(w4, 2), w4.shape[1], n_img, hi, wi, bias=b)
            stride, pad, hq, wq = 2, (1, 1), 2 * hi, 2 * wi
        relu_out = None
        if geom.get("relu"):          # ReLU in the convolution's epilogue (geom relu=True): the backward masks dy by the saved output
            assert kind == "conv" and residual is None
            relu_out = y
        ctx.save_for_backward(x0, x1, weight, relu_out)
        ctx.bias_param = bias
        ctx.meta = (kind, n_img, hi, wi, hq, wq, kh, kw, stride, pad, bias is not None, residual is not None)
        ctx.fork = bool(geom.get("fork"))
        if ctx.fork:
            assert kind == "conv" and stride == 1
            return (y, x0) if x1 is None else (y, x0, x1)
        return y

    @staticmethod
    def backward(ctx, dy, *dpass):
        x0, x1, weight, relu_out = ctx.saved_tensors
        kind, n_img, hi, wi, hq, wq, kh, kw, stride, pad, has_bias, has_res = ctx.meta
        dpass = [None if d is None else _c(d) for d in dpass] + [None, None]
        assert dy is not None, "ConvCL(fork): the convolution's own output must be used"
        dy = _c(dy)
        if relu_out is not None:
            dy = train_ops.relu_bwd(relu_out, dy)
        w4 = _c(weight.detach().reshape(weight.shape[0], weight.shape[1], kh, kw))
        need = ctx.needs_input_grad
        dx0 = dx1 = dw = db = None
        bias_p = ctx.bias_param
        taps_ok = kh * kw <= 16         # lfdm_wgrad_params.dw_layout = 1: the gradient in the weight's own layout, bias sums from the same pass
        fused_bias = (kind == "conv" and need[2] and taps_ok and
                      _thin_wgrad_route(kind, x1, stride, pad, kh, kw, x0.shape[1], w4.shape[0], hi, wi, hq, wq) is None)
        if has_bias and need[3]:
            db = grad_out(bias_p)
            if not fused_bias:
                train_ops.colsum(dy, out=db)
        if kind == "conv":
            cout = w4.shape[0]
            c0 = x0.shape[1]
            thin = _thin_wgrad_route(kind, x1, stride, pad, kh, kw, c0, cout, hi, wi, hq, wq) if need[2] else None
            if thin is not None:
                dw = _thin_wgrad(thin, _c(x0), dy, n_img, hi, wi, hq, wq, kh, pad[0], c0, cout).reshape(weight.shape)
            elif need[2] and taps_ok:
                dw = grad_out(weight)
                train_ops.conv_wgrad(_c(x0), dy, n_img, hi, wi, hq, wq, kh, kw, stride=stride, pad=pad, out=dw, ci_off=0, dbias=db)
                if x1 is not None:
                    train_ops.conv_wgrad(_c(x1), dy, n_img, hi, wi, hq, wq, kh, kw, stride=stride, pad=pad, out=dw, ci_off=c0)
            elif need[2]:
                parts = [train_ops.conv_wgrad(_c(x0), dy, n_img, hi, wi, hq, wq, kh, kw, stride=stride, pad=pad)]
                if x1 is not None:
                    parts.append(train_ops.conv_wgrad(_c(x1), dy, n_img, hi, wi, hq, wq, kh, kw, stride=stride, pad=pad))
                dwt = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)            # (taps, cin, cout)
                dw = dwt.view(kh, kw, -1, cout).permute(3, 2, 0, 1).reshape(weight.shape)
            srcs = [(x0, 0, c0, need[0], dpass[0])] + ([(x1, c0, c0 + x1.shape[1], need[1], dpass[1])] if x1 is not None else [])
            grads = []
            for xs, lo, hi_c, wanted, other in srcs:
                if not wanted:
                    grads.append(None)
                    continue
                ws = w4[:, lo:hi_c]
                if stride == 1:
                    ww = _pack_wino(weight, ws, dgrad=True, part=(lo, hi_c)) if _wino_ok(kh, kw, stride, pad, hq, wq, cout) else None
                    g = _conv(dy, (ws, 1), hi_c - lo, kh, kw, n_img, hq, wq, residual=other,
                              pad=(kh - 1 - pad[0], kw - 1 - pad[1]), weight_wino=ww)                # filter (cin, cout, kh, kw)
                else:
                    assert stride == 2 and kh == 4 and kw == 4 and pad == (1, 1)
                    g = ops.deconv4x4s2_cl(dy, ops.pack_conv_weight_dev(ws, 2), hi_c - lo, n_img, hq, wq)
                grads.append(g)
            dx0 = grads[0]
            dx1 = grads[1] if x1 is not None else None
        else:
            cin, cout = w4.shape[0], w4.shape[1]
            if need[2]:
                # roles exchanged: "input" = dy at (2h, 2w), "grad" = x at (h, w), stride 2, pad 1 -> layout 1 = (cin, cout, 4, 4)
                dw = grad_out(weight)
                train_ops.conv_wgrad(dy, _c(x0), n_img, hq, wq, hi, wi, 4, 4, stride=2, pad=(1, 1), out=dw)
            if need[0]:
                dx0 = ops.conv2d_cl(dy, ops.pack_conv_weight_dev(w4, 0), cin, 4, 4, n_img, hq, wq, stride=2, pad=(1, 1))
        dres = dy if (has_res and need[4]) else None
        return dx0, dx1, dw, db, dres, None


class GroupNormSiLU(Function):
    """Block.forward norm -> (scale+1, shift) -> SiLU (+ residual after the activation), :200-211 / :237."""

    @staticmethod
    def forward(ctx, x, gamma, beta, scale_shift, residual, batch, silu):
        xs = _c(x.detach())
        ss = None if scale_shift is None else _c(scale_shift.detach())
        y, partial, nchunk = train_ops.groupnorm_silu_train(
            xs, batch, _c(gamma.detach()), _c(beta.detach()), scale_shift=ss,
            residual=None if residual is None else _c(residual.detach()), silu=silu)
        ctx.save_for_backward(xs, gamma, beta, ss, partial)
        ctx.meta = (batch, nchunk, silu, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, ss, partial = ctx.saved_tensors
        batch, nchunk, silu, has_res = ctx.meta
        dy = _c(dy)
        dgb, dg, db = grad_out_pair(gamma, beta)
        dx, _, _, dss = train_ops.groupnorm_silu_bwd(x, dy, batch, _c(gamma.detach()), _c(beta.detach()), partial, nchunk,
                                                     scale_shift=ss, silu=silu, dgb=dgb)
        return dx, dg, db, dss, (dy if has_res else None), None, None


class LayerNormCL(Function):
    """Channel LayerNorm (gamma only), :170-179.  fork=True -> (normed, x): the second output is x itself for the residual path around the
    normalised branch (Residual(PreNorm(fn)): fn(norm(x)) + x, :118-125 / :181-188).  Both consumers' gradients then arrive at THIS node
    together and the backward kernel stores their sum - the autograd engine's separate add per block (19 per DM step) is gone."""

    @staticmethod
    def forward(ctx, x, gamma, fork=False):
        xs = _c(x.detach())
        g = _c(gamma.detach().reshape(-1))
        ctx.save_for_backward(xs, g)
        ctx.gamma_param = gamma
        ctx.fork = fork
        y = ops.layernorm_cl(xs, g)
        return (y, x) if fork else y

    @staticmethod
    def backward(ctx, dy, dpass=None):
        x, g = ctx.saved_tensors
        dg = grad_out(ctx.gamma_param)
        if dy is None:              # only the residual path was used downstream
            dg.zero_()
            return dpass, dg, None
        dx, _ = train_ops.layernorm_bwd(x, _c(dy), g, dgamma=dg, dx_add=None if dpass is None else _c(dpass))
        return dx, dg, None


class AttentionCL(Function):
    """Attention.forward core (:303-363) on qkv rows; bias = (8, L, L) relative-position table or None."""

    @staticmethod
    def forward(ctx, qkv, bias, rot_cos, rot_sin, batch, frames, hw, mode):
        q = _c(qkv.detach())
        b = None if bias is None else _c(bias.detach())
        ctx.save_for_backward(q, b, rot_cos, rot_sin)
        ctx.meta = (batch, frames, hw, mode)
        return ops.attention_cl(q, batch, frames, hw, mode, bias=b, rot_cos=rot_cos, rot_sin=rot_sin)

    @staticmethod
    def backward(ctx, dout):
        q, b, rot_cos, rot_sin = ctx.saved_tensors
        batch, frames, hw, mode = ctx.meta
        dqkv, dbias = train_ops.attention_bwd(q, _c(dout), batch, frames, hw, mode, bias=b, rot_cos=rot_cos, rot_sin=rot_sin)
        return dqkv, dbias, None, None, None, None, None, None


class LinearAttentionCL(Function):
    """SpatialLinearAttention core (:254-263) on qkv rows."""

    @staticmethod
    def forward(ctx, qkv, n_frames, hw):
        q = _c(qkv.detach())
        ctx.save_for_backward(q)
        ctx.meta = (n_frames, hw)
        return ops.linear_attention_cl(q, n_frames, hw)

    @staticmethod
    def backward(ctx, dout):
        (q,) = ctx.saved_tensors
        n_frames, hw = ctx.meta
        return train_ops.linear_attention_bwd(q, _c(dout), n_frames, hw), None, None


class MultiLinear(Function):
    """(y_0, ..., y_{n-1}) with y_j = act(x) @ w_j.T + b_j: Linear layers that share their input - every ResnetBlock.mlp of a UNet
    forward (SiLU -> Linear on cat(time_emb, cond), :230-233,240-245,562) or one layer of time_mlp (:441-447) - one native launch forward,
    three backward (weight / bias gradients into the optimizer's slots, the input gradient summed over all blocks in a fixed order).
    apply(x, act, n, w_0..w_{n-1}, b_0..b_{n-1}); a bias may be None.  The kernels take up to train_ops.ML_ROWS (16) rows per launch:
    a larger per-process batch (the reference's mhad script trains with 20 videos) runs as ceil(rows / 16) launches over row slices,
    weight / bias gradients summed over the slices in slice order."""

    @staticmethod
    def forward(ctx, x, act, n, *wb):
        ws, bs = wb[:n], wb[n:]
        xs = _c(x.detach())
        wd = [_c(w.detach()) for w in ws]
        bd = [None if b is None else _c(b.detach()) for b in bs]
        rows, step = xs.shape[0], train_ops.ML_ROWS
        if rows <= step:
            ys = train_ops.multi_linear(xs, wd, bd, act)
        else:
            parts = [train_ops.multi_linear(xs[r:r + step], wd, bd, act) for r in range(0, rows, step)]
            ys = [torch.cat([p[j] for p in parts], dim=0) for j in range(n)]
        ctx.save_for_backward(xs, *ws)
        ctx.bias_params = bs
        ctx.meta = (act, n)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        xs, *ws = ctx.saved_tensors
        act, n = ctx.meta
        bs = ctx.bias_params
        need = ctx.needs_input_grad
        dws = [grad_out(w) if need[3 + j] else None for j, w in enumerate(ws)]
        dbs = [grad_out(b) if (b is not None and need[3 + n + j]) else None for j, b in enumerate(bs)]
        wd = [_c(w.detach()) for w in ws]
        dyc = [None if d is None else _c(d) for d in dys]
        rows, step = xs.shape[0], train_ops.ML_ROWS
        if rows <= step:
            dx = train_ops.multi_linear_bwd(xs, wd, dyc, act, dws, dbs, want_dx=need[0])
        else:
            dxs = []
            for r in range(0, rows, step):
                first = r == 0
                tw = [None if d is None else (d if first else torch.empty_like(d)) for d in dws]
                tb = [None if d is None else (d if first else torch.empty_like(d)) for d in dbs]
                dxs.append(train_ops.multi_linear_bwd(xs[r:r + step], wd, [None if d is None else d[r:r + step] for d in dyc], act, tw, tb,
                                                      want_dx=need[0]))
                if not first:
                    for acc, part in zip(dws + dbs, tw + tb):
                        if acc is not None:
                            acc.add_(part)
            dx = torch.cat(dxs, dim=0) if need[0] else None
        return (dx, None, None, *dws, *dbs)


def multi_linear(x, weights, biases, act=train_ops.ACT_NONE):
    """Linear layers sharing their input (MultiLinear).  The native kernels need an input width k <= train_ops.ML_KMAX (1024) that is a
    multiple of 4 - the MUG / MHAD / NATOPS models sit at or below it (dim * 4 + 768 = 1024); a wider conditioning vector (dim = 128)
    runs the same arithmetic as torch ops on the device."""
    k = x.shape[1]
    if k > train_ops.ML_KMAX or k % 4:
        import torch.nn.functional as F
        a = {train_ops.ACT_NONE: lambda v: v, train_ops.ACT_SILU: F.silu, train_ops.ACT_GELU: F.gelu}[act](x)
        return tuple(F.linear(a, w, b) for w, b in zip(weights, biases))
    return MultiLinear.apply(x, act, len(weights), *weights, *biases)


class PlanarToCL(Function):
    """(N, C, HW) planar -> (N*HW, C) rows."""

    @staticmethod
    def forward(ctx, x):
        n, c, hw = x.shape
        ctx.meta = (n, c, hw)
        return ops.planar_to_cl(_c(x.detach()), n, c, hw)

    @staticmethod
    def backward(ctx, dy):
        n, c, hw = ctx.meta
        return ops.cl_to_planar(_c(dy), n, c, hw).view(n, c, hw)


class CLToPlanar(Function):
    """(N*HW, C) rows -> (N, C, HW) planar."""

    @staticmethod
    def forward(ctx, x, n, hw):
        c = x.shape[1]
        ctx.meta = (n, c, hw)
        return ops.cl_to_planar(_c(x.detach()), n, c, hw).view(n, c, hw)

    @staticmethod
    def backward(ctx, dy):
        n, c, hw = ctx.meta
        return ops.planar_to_cl(_c(dy), n, c, hw), None, None


class Upsample2Pad(Function):
    """nearest x2 + pad (zeros / reflect) of CL rows: the input side of the use_deconv=False Upsample (:160-163)."""

    @staticmethod
    def forward(ctx, x, n_img, h, w, pad, reflect):
        ctx.meta = (n_img, h, w, pad, reflect)
        return train_ops.upsample2_pad(_c(x.detach()), n_img, h, w, pad, reflect)

    @staticmethod
    def backward(ctx, dy):
        n_img, h, w, pad, reflect = ctx.meta
        return train_ops.upsample2_pad(_c(dy), n_img, h, w, pad, reflect, backward=True), None, None, None, None, None


def conv_cl(x0, weight, bias, *, x1=None, residual=None, **geom):
    return ConvCL.apply(x0, x1, weight, bias, residual, geom)
