"""Differentiable forward of Unet3D for the DM training step (reference Unet3D.forward :528-588 as used by
GaussianDiffusion.p_losses :856-895).  Same parameters (ParamTree keys), same channels-last dataflow as the
sampling executor in unet.py, but every block is a torch.autograd.Function from autograd.py whose forward AND
backward are liblfdm_hip.so kernels.  What stays in torch: the (B, <=1024)-sized conditioning vectors (time MLP,
ResnetBlock.mlp, null-cond merge), weight re-layouts, and the gradient accumulation autograd performs.

Differences from the sampling executor, all exact: no LayerNorm->qkv folding, no conv->GroupNorm statistics fusion
and no in-place buffers (the inputs of every op are saved for its backward); the init_conv keeps the
linearity split [3 step-dependent channels | 256 fea channels] because the fea term and its weight gradient are
40x cheaper on the T-reduced tensors.
"""
import math

import torch
import torch.nn.functional as F

from . import autograd as A
from . import ops, train_ops
from .unet import prob_mask_like, rel_pos_bias_table


def _time_embedding(unet, time):
    half = unet.dim // 2
    freq = torch.exp(torch.arange(half, device=time.device) * -(math.log(10000) / (half - 1)))
    e = time[:, None].float() * freq[None, :]
    e = torch.cat((e.sin(), e.cos()), dim=-1)
    # time_mlp (:441-447): Linear -> GELU -> Linear; the GELU is the input activation of the second native launch
    (e,) = A.multi_linear(e, [unet.get("time_mlp.1.weight")], [unet.get("time_mlp.1.bias")])
    (e,) = A.multi_linear(e, [unet.get("time_mlp.3.weight")], [unet.get("time_mlp.3.bias")], act=train_ops.ACT_GELU)
    return e


def _resblock_prefixes(unet):
    """Every ResnetBlock of the forward, in execution order (the two output heads take no conditioning)."""
    nl = len(unet.levels)
    ps = ["downs.%d.%d." % (lvl, i) for lvl in range(nl) for i in (0, 1)] + ["mid_block1.", "mid_block2."]
    return ps + ["ups.%d.%d." % (lvl, i) for lvl in range(nl) for i in (0, 1)]


def _block_scale_shifts(unet, tc):
    """{prefix: (B, 2*dim_out)} for every ResnetBlock with an mlp (:230-233, :240-245): all of them read SiLU(tc), so they run as one
    native launch (and one backward) instead of a SiLU + Linear pair of vendor kernels per block."""
    ps = [p for p in _resblock_prefixes(unet) if unet.has(p + "mlp.1.weight")]
    out = {}
    for i in range(0, len(ps), train_ops.ML_MAX):
        chunk = ps[i:i + train_ops.ML_MAX]
        ys = A.multi_linear(tc, [unet.get(p + "mlp.1.weight") for p in chunk], [unet.get(p + "mlp.1.bias") for p in chunk],
                            act=train_ops.ACT_SILU)
        out.update(zip(chunk, ys))
    return out


class _Ctx:
    pass


def _resblock(unet, c, prefix, x, skip, res, cout, ss_of):
    g = unet.get
    n_img = c.batch * c.frames
    # fork: x (and skip) come back as outputs for the block's second reader (res_conv / the identity skip), so that both gradients meet
    # in block1's data-gradient convolution (autograd.ConvCL)
    forked = A.conv_cl(x, g(prefix + "block1.proj.weight"), g(prefix + "block1.proj.bias"), x1=skip,
                       n_img=n_img, hi=res, wi=res, fork=True)
    h, x = forked[0], forked[1]
    skip = forked[2] if skip is not None else None
    ss = None if ss_of is None else ss_of.get(prefix)                                                  # (B, 2*cout)
    h = A.GroupNormSiLU.apply(h, g(prefix + "block1.norm.weight"), g(prefix + "block1.norm.bias"), ss, None, c.batch, True)
    h = A.conv_cl(h, g(prefix + "block2.proj.weight"), g(prefix + "block2.proj.bias"), n_img=n_img, hi=res, wi=res)
    if unet.has(prefix + "res_conv.weight"):
        h = A.GroupNormSiLU.apply(h, g(prefix + "block2.norm.weight"), g(prefix + "block2.norm.bias"), None, None, c.batch, True)
        return A.conv_cl(x, g(prefix + "res_conv.weight"), g(prefix + "res_conv.bias"), x1=skip, residual=h,
                         n_img=n_img, hi=res, wi=res)
    assert skip is None
    return A.GroupNormSiLU.apply(h, g(prefix + "block2.norm.weight"), g(prefix + "block2.norm.bias"), None, x, c.batch, True)


def _temporal_attn(unet, c, prefix, x, res, focus=None):
    g = unet.get
    n_img = c.batch * c.frames
    normed, x = A.LayerNormCL.apply(x, g(prefix + "fn.norm.gamma"), True)       # x again: the residual path's gradient meets the norm's
    qkv = A.conv_cl(normed, g(prefix + "fn.fn.fn.to_qkv.weight"), None, n_img=n_img, hi=res, wi=res)
    if focus is not None and all(focus):          # Attention.forward :313-317: the values pass straight through to_out
        att = qkv[:, 512:768].contiguous()
    else:
        att = A.AttentionCL.apply(qkv, c.bias, c.cos, c.sin, c.batch, c.frames, res * res, 0)
        if focus is not None and any(focus):      # :342-352: a focused sample's softmax rows are exactly one-hot -> its value rows
            rowmask = torch.tensor(focus, device=x.device).repeat_interleave(c.frames * res * res).view(-1, 1)
            att = torch.where(rowmask, qkv[:, 512:768], att)
    return A.conv_cl(att, g(prefix + "fn.fn.fn.to_out.weight"), None, residual=x, n_img=n_img, hi=res, wi=res)


def _mid_spatial_attn(unet, c, prefix, x, res):
    g = unet.get
    n_img = c.batch * c.frames
    normed, x = A.LayerNormCL.apply(x, g(prefix + "fn.norm.gamma"), True)       # x again: the residual path's gradient meets the norm's
    qkv = A.conv_cl(normed, g(prefix + "fn.fn.fn.to_qkv.weight"), None, n_img=n_img, hi=res, wi=res)
    att = A.AttentionCL.apply(qkv, None, None, None, c.batch, c.frames, res * res, 1)
    return A.conv_cl(att, g(prefix + "fn.fn.fn.to_out.weight"), None, residual=x, n_img=n_img, hi=res, wi=res)


def _linear_attn(unet, c, prefix, x, res):
    g = unet.get
    n_img = c.batch * c.frames
    normed, x = A.LayerNormCL.apply(x, g(prefix + "fn.norm.gamma"), True)       # x again: the residual path's gradient meets the norm's
    qkv = A.conv_cl(normed, g(prefix + "fn.fn.to_qkv.weight"), None, n_img=n_img, hi=res, wi=res)
    att = A.LinearAttentionCL.apply(qkv, n_img, res * res)
    return A.conv_cl(att, g(prefix + "fn.fn.to_out.weight"), g(prefix + "fn.fn.to_out.bias"), residual=x,
                     n_img=n_img, hi=res, wi=res)


def unet_train_forward(unet, x_dyn, fea, time, cond, null_cond_prob=0., none_cond_mask=None, rank_shard=None, focus=None,
                       prob_focus_present=0.):
    """x_dyn (B, 3, T, S, S) noisy flow/occlusion, fea (B, 256, S, S) reference-image features (constant over T),
    time (B,) long, cond (B, 768)  ->  eps_hat (B, 3, T, S, S) with grad to every UNet parameter."""
    g = unet.get
    c = _Ctx()
    A.repack_stale()          # every Winograd filter the previous step used (forward + data-gradient forms): one launch
    b, n_dyn, t, s, _ = x_dyn.shape
    c.batch, c.frames = b, t
    dev = x_dyn.device
    n_img = b * t
    dim = unet.dim

    # --- focus_present_mask (:542-543; drawn before the null-condition mask, like the reference): a list of bools or None
    if focus is None and prob_focus_present == 1:
        focus = [True] * b
    elif focus is None and prob_focus_present != 0:
        if rank_shard is not None:
            from .diffusion import shard_bounds
            lo, hi, total = shard_bounds(rank_shard, b)
            focus = prob_mask_like((total,), prob_focus_present, device=dev)[lo:hi].tolist()
        else:
            focus = prob_mask_like((b,), prob_focus_present, device=dev).tolist()
    if focus is not None:
        focus = [bool(v) for v in focus]
        focus = focus if any(focus) else None

    # --- conditioning (B x 1024 vectors: torch) (:549-562)
    if rank_shard is not None:       # sharded data parallelism: the draw of the global batch, this rank's slice
        from .diffusion import shard_bounds
        lo, hi, total = shard_bounds(rank_shard, b)
        unet.null_cond_mask = prob_mask_like((total,), null_cond_prob, device=dev)[lo:hi]
    else:
        unet.null_cond_mask = prob_mask_like((b,), null_cond_prob, device=dev)
    if none_cond_mask is not None:
        unet.null_cond_mask = torch.logical_or(unet.null_cond_mask, torch.as_tensor(none_cond_mask, device=dev))
    temb = _time_embedding(unet, time)
    if unet.has_cond:
        null = unet.null_cond_emb.to(dev)
        cond = torch.where(unet.null_cond_mask.view(b, 1), null, cond.float())
        tc = torch.cat((temb, cond), dim=-1)
    else:
        tc = temb
    tc = _block_scale_shifts(unet, tc.contiguous())

    # --- relative position bias / rotary tables (:545, :395)
    c.bias = rel_pos_bias_table(g("time_rel_pos_bias.relative_attention_bias.weight"), t)
    freqs = g("init_temporal_attn.fn.fn.fn.rotary_emb.freqs").detach()
    ang = torch.arange(t, device=dev).float()[:, None] * freqs[None, :]
    c.cos, c.sin = ang.cos().contiguous(), ang.sin().contiguous()

    # --- init_conv split by linearity (:410, :547): [n_dyn dynamic channels | fea channels]
    w0, b0 = g("init_conv.weight"), g("init_conv.bias")
    # The n_dyn = 3 dynamic channels as a 1x1 convolution over their unfolded 7x7 patches (n_dyn*49 = 147 -> 160 columns):
    # as a 7x7 convolution with 3 (4) input channels both the forward (generic gather path) and above all the weight
    # gradient (49 taps x a 64-channel tile holding 4 channels: 1.7 ms per B = 8 step) waste the matrix pipe; the patches
    # cost one 200 MB tensor, the input needs no gradient.
    # (three torch launches: pad, zero fill, one strided copy of the window view - F.unfold launches an im2col per image)
    kk = w0.shape[-1]
    kcols = n_dyn * kk * kk
    kpad = (kcols + 31) // 32 * 32
    xp = F.pad(x_dyn.detach().permute(0, 2, 1, 3, 4).reshape(n_img, n_dyn, s, s).float(), (kk // 2,) * 4)
    win = xp.unfold(2, kk, 1).unfold(3, kk, 1)                                       # (n, c, s, s, ky, kx) view
    x_cl = torch.zeros(n_img * s * s, kpad, dtype=torch.float32, device=dev)
    x_cl.view(n_img, s, s, kpad)[..., :kcols].unflatten(-1, (n_dyn, kk, kk)).copy_(win.permute(0, 2, 3, 1, 4, 5))
    w_dyn = F.pad(w0[:, :n_dyn].reshape(dim, kcols), (0, kpad - kcols)).view(dim, kpad, 1, 1)     # (c, ky, kx) order = unfold's
    r = A.conv_cl(x_cl, w_dyn, None, n_img=n_img, hi=s, wi=s, pad=(0, 0))
    fea_cl = ops.planar_to_cl(fea.detach().reshape(b, fea.shape[1], s * s).contiguous(), b, fea.shape[1], s * s)
    term = A.conv_cl(fea_cl, w0[:, n_dyn:], b0, n_img=b, hi=s, wi=s, pad=(3, 3))                      # (B*S*S, dim)
    r = (r.view(b, t, s * s, dim) + term.view(b, 1, s * s, dim)).reshape(n_img * s * s, dim)

    x = _temporal_attn(unet, c, "init_temporal_attn.", r, s)
    skips = []
    res = s
    levels = unet.levels
    nl = len(levels)
    for lvl, (ci, co) in enumerate(levels):
        p = "downs.%d." % lvl
        x = _resblock(unet, c, p + "0.", x, None, res, co, tc)
        x = _resblock(unet, c, p + "1.", x, None, res, co, tc)
        x = _linear_attn(unet, c, p + "2.", x, res)
        x = _temporal_attn(unet, c, p + "3.", x, res, focus)
        skips.append(x)
        if lvl < nl - 1:
            x = A.conv_cl(x, g(p + "4.weight"), g(p + "4.bias"), n_img=n_img, hi=res, wi=res, stride=2, pad=(1, 1))
            res //= 2
    mid = levels[-1][1]
    x = _resblock(unet, c, "mid_block1.", x, None, res, mid, tc)
    x = _mid_spatial_attn(unet, c, "mid_spatial_attn.", x, res)
    x = _temporal_attn(unet, c, "mid_temporal_attn.", x, res, focus)
    x = _resblock(unet, c, "mid_block2.", x, None, res, mid, tc)
    for lvl, (ci, co) in enumerate(reversed(levels)):
        p = "ups.%d." % lvl
        x = _resblock(unet, c, p + "0.", x, skips.pop(), res, ci, tc)
        x = _resblock(unet, c, p + "1.", x, None, res, ci, tc)
        x = _linear_attn(unet, c, p + "2.", x, res)
        x = _temporal_attn(unet, c, p + "3.", x, res, focus)
        if lvl < nl - 1:
            if unet.use_deconv:
                x = A.conv_cl(x, g(p + "4.weight"), g(p + "4.bias"), n_img=n_img, hi=res, wi=res, kind="deconv")
            else:   # Upsample(nearest x2) -> Conv3d k=(1,3,3) with padding_mode (:160-163): padded input materialised
                xp = A.Upsample2Pad.apply(x, n_img, res, res, 1, unet.padding_mode == "reflect")
                x = A.conv_cl(xp, g(p + "4.1.weight"), g(p + "4.1.bias"), n_img=n_img, hi=2 * res + 2, wi=2 * res + 2, pad=(0, 0))
            res *= 2

    # --- output heads (:493-509, :587-588): two ResnetBlocks on cat(x, r), then the 1x1 convs as ONE 4-column
    # block-diagonal projection [flow(2) | occlusion(1) | zero pad] over cat(y_flow, y_occ)
    yf = _resblock(unet, c, "final_conv.0.", x, r, res, dim, None)
    yo = _resblock(unet, c, "occlusion_map.0.", x, r, res, dim, None)
    wf = g("final_conv.1.weight").reshape(unet.out_grid_dim, dim)
    wo = g("occlusion_map.1.weight").reshape(unet.out_conf_dim, dim)
    nout = unet.out_grid_dim + unet.out_conf_dim
    npad = (4 - nout % 4) % 4
    wc = torch.cat((torch.cat((wf, torch.zeros_like(wf)), dim=1), torch.cat((torch.zeros_like(wo), wo), dim=1),
                    wf.new_zeros(npad, 2 * dim)), dim=0)
    bc = torch.cat((g("final_conv.1.bias"), g("occlusion_map.1.bias"), wf.new_zeros(npad)))
    out = A.conv_cl(yf, wc, bc, x1=yo, n_img=n_img, hi=res, wi=res)                                   # (rows, 4)
    planar = A.CLToPlanar.apply(out, n_img, res * res)                                               # (B*T, 4, HW)
    return planar.view(b, t, nout + npad, res, res)[:, :, :nout].permute(0, 2, 1, 3, 4)
