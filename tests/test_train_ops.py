"""Backward kernels (include/lfdm_hip.h, training section) against torch autograd of the ATen op the
reference calls.  Same dual-backend scheme as test_ops_parity.py: "emu" (x86 emulation of the same kernel
sources, CPU) and "hip" (MI355X, marked gpu)."""
import pytest
import torch
import torch.nn.functional as F

from cvpr23_lfdm_amd import ops, train_ops
from util import assert_close, from_cl, rnd, to_cl

TOL = 2e-4


def big(dev):
    return dev == "cuda"


@pytest.mark.parametrize("cin,cout,k,stride,res", [(8, 12, 3, 1, 6), (68, 72, 3, 1, 5), (132, 8, 1, 1, 4),
                                                   (8, 8, 4, 2, 8), (4, 64, 7, 1, 8), (36, 20, 3, 1, 3), (12, 16, 3, 1, 16)])
def test_conv_wgrad(backend, cin, cout, k, stride, res):
    dev = backend
    n = 3
    if big(dev):
        cin, cout, res, n = {8: (64, 64), 68: (256, 128), 132: (512, 768), 4: (4, 64), 36: (36, 20), 12: (128, 64)}[cin] + (16, 10)
        if cin == 36:
            res, n = 3, 1                     # M = 9 rows: less than one 32-row chunk
        if k == 4:
            cin, cout = 128, 128
    pad = {3: 1, 1: 0, 4: 1, 7: 3}[k]
    x = rnd(n, cin, res, res, seed=1).requires_grad_(False)
    w = (rnd(cout, cin, k, k, seed=2) * 0.1).requires_grad_(True)
    y = F.conv2d(x, w, stride=stride, padding=pad)
    dy = rnd(*y.shape, seed=3)
    y.backward(dy)
    hq = y.shape[2]
    dw = train_ops.conv_wgrad(to_cl(x).to(dev), to_cl(dy).to(dev), n, res, res, hq, hq, k, k, stride=stride, pad=(pad, pad))
    got = dw.cpu().view(k, k, cin, cout).permute(3, 2, 0, 1)
    scale = float(w.grad.abs().max())
    assert_close(got / scale, w.grad / scale, TOL, "conv wgrad")
    db = train_ops.colsum(to_cl(dy).to(dev))
    ref_db = dy.sum(dim=(0, 2, 3))
    assert_close(db.cpu() / float(ref_db.abs().max()), ref_db / float(ref_db.abs().max()), TOL, "bias grad")
    if k * k <= 16:
        # lfdm_wgrad_params.dw_layout = 1: the reference (cout, cin_total, k, k) layout written in two input-channel halves (a convolution
        # over cat(x0, x1)), bias gradient from the same pass - bit-identical to the tap-major result and to colsum's order-independent parts
        c0 = (cin // 8) * 4 or cin
        xcl = to_cl(x).to(dev)
        out = torch.full((cout, cin, k, k), float("nan"), device=dev)
        db2 = torch.full((cout,), float("nan"), device=dev)
        train_ops.conv_wgrad(xcl[:, :c0], to_cl(dy).to(dev), n, res, res, hq, hq, k, k, stride=stride, pad=(pad, pad), out=out, ci_off=0, dbias=db2)
        if c0 < cin:
            train_ops.conv_wgrad(xcl[:, c0:], to_cl(dy).to(dev), n, res, res, hq, hq, k, k, stride=stride, pad=(pad, pad), out=out, ci_off=c0)
        assert_close(out.cpu() / scale, w.grad / scale, TOL, "conv wgrad, reference layout")
        assert_close(db2.cpu() / float(ref_db.abs().max()), ref_db / float(ref_db.abs().max()), TOL, "bias grad from the wgrad pass")
        if c0 == cin:
            # (the two layouts fold their row-slice slabs in different - each fixed - orders since round 5: equal up to fp32 summation order)
            assert_close(out.cpu() / scale, got.contiguous() / scale, 1e-5, "layout 1 vs layout 0")


@pytest.mark.parametrize("c,with_ss,silu", [(32, True, True), (64, False, True), (32, False, False), (512, True, True)])
def test_groupnorm_bwd(backend, c, with_ss, silu):
    dev = backend
    b, t, s = 2, 3, 4
    if big(dev):
        b, t, s, c = 2, 40, 16, {32: 128, 64: 512, 512: 1024}[c]
    elif c == 512:
        b, t, s = 1, 1, 3                     # 9 pixels: fewer than one statistics chunk, widest supported rows
    x = rnd(b, c, t, s, s, seed=1).requires_grad_(True)
    gamma = (1 + 0.2 * rnd(c, seed=2)).requires_grad_(True)
    beta = (0.1 * rnd(c, seed=3)).requires_grad_(True)
    ss = (0.3 * rnd(b, 2 * c, seed=4)).requires_grad_(True) if with_ss else None
    y = F.group_norm(x, 8, gamma, beta, eps=1e-5)
    if with_ss:
        y = y * (ss[:, :c].view(b, c, 1, 1, 1) + 1) + ss[:, c:].view(b, c, 1, 1, 1)
    if silu:
        y = F.silu(y)
    dy = rnd(*y.shape, seed=5)
    y.backward(dy)
    to_rows = lambda v: v.detach().permute(0, 2, 3, 4, 1).reshape(-1, c).contiguous().to(dev)
    xd = to_rows(x)
    yk, partial, nchunk = train_ops.groupnorm_silu_train(xd, b, gamma.detach().to(dev), beta.detach().to(dev),
                                                         scale_shift=ss.detach().to(dev) if with_ss else None, silu=silu)
    assert_close(yk.cpu(), to_rows(y).cpu(), TOL, "gn fwd")
    dx, dg, db, dss = train_ops.groupnorm_silu_bwd(xd, to_rows(dy), b, gamma.detach().to(dev), beta.detach().to(dev), partial,
                                                   nchunk, scale_shift=ss.detach().to(dev) if with_ss else None, silu=silu)
    assert_close(dx.cpu(), to_rows(x.grad).cpu(), TOL, "gn dx")
    sc = float(gamma.grad.abs().max())
    assert_close(dg.cpu() / sc, gamma.grad / sc, TOL, "gn dgamma")
    sc = float(beta.grad.abs().max())
    assert_close(db.cpu() / sc, beta.grad / sc, TOL, "gn dbeta")
    if with_ss:
        sc = float(ss.grad.abs().max())
        assert_close(dss.cpu() / sc, ss.grad / sc, TOL, "gn dscale_shift")


@pytest.mark.parametrize("c", [32, 72, 1024, 64, 128])        # 64 / 128: the several-rows-per-wavefront form (ragged last group)
def test_layernorm_bwd(backend, c):
    dev = backend
    rows = 37 if c != 1024 else 5
    if big(dev):
        rows, c = 40 * 1024 + 3, {32: 64, 72: 512, 1024: 1024, 64: 64, 128: 128}[c]
    x = rnd(rows, c, seed=1).requires_grad_(True)
    gamma = (1 + 0.2 * rnd(c, seed=2)).requires_grad_(True)
    mean = x.mean(dim=1, keepdim=True)
    var = x.var(dim=1, unbiased=False, keepdim=True)
    y = (x - mean) / (var + 1e-5).sqrt() * gamma          # reference LayerNorm :176-179
    dy = rnd(rows, c, seed=3)
    y.backward(dy)
    dx, dg = train_ops.layernorm_bwd(x.detach().to(dev), dy.to(dev), gamma.detach().to(dev))
    assert_close(dx.cpu(), x.grad, TOL, "ln dx")
    sc = float(gamma.grad.abs().max())
    assert_close(dg.cpu() / sc, gamma.grad / sc, TOL, "ln dgamma")


def _attention_ref(qkv_tokens, bias, rotary):
    """Attention.forward semantics (video_flow_diffusion.py:303-363) on (..., n, 768) tokens."""
    import lfdm_oracle as O
    q, k, v = qkv_tokens.chunk(3, dim=-1)
    heads = lambda z: z.reshape(*z.shape[:-1], 8, 32).transpose(-2, -3)
    q, k, v = heads(q), heads(k), heads(v)
    q = q * (32 ** -0.5)
    if rotary is not None:
        q, k = O.apply_rotary(q, *rotary), O.apply_rotary(k, *rotary)
    sim = q @ k.transpose(-1, -2)
    if bias is not None:
        sim = sim + bias
    sim = sim - sim.amax(dim=-1, keepdim=True).detach()
    out = sim.softmax(dim=-1) @ v
    return out.transpose(-2, -3).reshape(*qkv_tokens.shape[:-1], 256)


@pytest.mark.parametrize("frames", [4, 36, 40, 44, 64])      # 33..40 frames: the 40-row / five-wave form (36: zero rows inside the tile)
def test_attention_temporal_bwd(backend, frames):
    import lfdm_oracle as O
    dev = backend
    b, s = (1, 16) if (big(dev) and frames == 40) else (2, 2)
    if not big(dev) and frames >= 36:
        b, s = 1, (3 if frames == 40 else 1)     # few sequences under the emulator; 64 frames = the 2-wave LP=64 variant
    hw = s * s
    qkv = rnd(b, frames, hw, 768, seed=1).requires_grad_(True)
    emb = rnd(32, 8, seed=2)
    bias = O.rel_pos_bias(emb, frames).clone().requires_grad_(True)
    freqs = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
    cos, sin = O.rotary_tables(freqs, frames)
    out = _attention_ref(qkv.permute(0, 2, 1, 3), bias, (cos, sin)).permute(0, 2, 1, 3).reshape(-1, 256)
    dout = rnd(*out.shape, seed=3)
    out.backward(dout)
    dqkv, dbias = train_ops.attention_bwd(qkv.detach().reshape(-1, 768).to(dev), dout.to(dev), b, frames, hw, 0,
                                          bias=bias.detach().contiguous().to(dev),
                                          rot_cos=cos[:, 0::2].contiguous().to(dev), rot_sin=sin[:, 0::2].contiguous().to(dev))
    assert_close(dqkv.cpu(), qkv.grad.reshape(-1, 768), TOL, "temporal attention dqkv")
    sc = float(bias.grad.abs().max())
    assert_close(dbias.cpu() / sc, bias.grad / sc, TOL, "temporal attention dbias")


@pytest.mark.parametrize("hw", [16, 64])
def test_attention_spatial_bwd(backend, hw):
    dev = backend
    b, frames = 1, 3
    qkv = rnd(b, frames, hw, 768, seed=3).requires_grad_(True)
    out = _attention_ref(qkv, None, None).reshape(-1, 256)
    dout = rnd(*out.shape, seed=4)
    out.backward(dout)
    dqkv, _ = train_ops.attention_bwd(qkv.detach().reshape(-1, 768).to(dev), dout.to(dev), b, frames, hw, 1)
    assert_close(dqkv.cpu(), qkv.grad.reshape(-1, 768), TOL, "spatial attention dqkv")


@pytest.mark.parametrize("hw", [16, 80])
def test_linear_attention_bwd(backend, hw):
    dev = backend
    nf = 3
    if big(dev) and hw == 80:
        nf, hw = 40, 1024
    qkv = rnd(nf, hw, 768, seed=4).requires_grad_(True)
    q, k, v = [z.reshape(nf, hw, 8, 32).permute(0, 2, 3, 1) for z in qkv.chunk(3, dim=-1)]  # b h d n
    q = q.softmax(dim=-2) * (32 ** -0.5)
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).permute(0, 3, 1, 2).reshape(nf * hw, 256)
    dout = rnd(*out.shape, seed=5)
    out.backward(dout)
    dqkv = train_ops.linear_attention_bwd(qkv.detach().reshape(-1, 768).to(dev), dout.to(dev), nf, hw)
    sc = float(qkv.grad.abs().max())
    assert_close(dqkv.cpu() / sc, qkv.grad.reshape(-1, 768) / sc, TOL, "linear attention dqkv")


def test_depthwise_down(backend):
    """AntiAliasInterpolation2d (util.py:217-264): pad (ka, kb) -> depthwise Gaussian -> [::4]."""
    from cvpr23_lfdm_amd.params import antialias_kernel
    dev = backend
    n, c, h = (3, 3, 20) if not big(dev) else (40, 3, 128)
    x = rnd(n, c, h, h, seed=1)
    w = antialias_kernel(c, 0.25)
    ks = w.shape[-1]
    ka = ks // 2
    ref = F.conv2d(F.pad(x, (ka, ka, ka, ka)), w, groups=c)[:, :, ::4, ::4]
    out = ops.depthwise_down_planar(x.to(dev), w.to(dev), 4, ka, ka)
    assert_close(out.cpu(), ref, 1e-5, "antialias down")


@pytest.mark.parametrize("rows,k,ns,act", [(8, 1024, (128, 256, 2048, 128), 1), (2, 64, (256,), 0), (3, 256, (256,), 2),
                                            (16, 32, (8, 24, 4), 1), (8, 8, tuple([8] * 32), 2),
                                            (20, 64, (16, 24, 8, 4), 1), (37, 32, (12,), 2), (5, 1030, (8, 4), 1)])
def test_multi_linear(backend, rows, k, ns, act):
    """lfdm_multi_linear_f32 / _bwd_f32 against torch: every ResnetBlock.mlp of a forward (SiLU -> Linear on the shared cat(time_emb,
    cond), video_flow_diffusion.py:230-233,562) and the two time_mlp layers (:441-447), incl. a missing bias, an unused output (dy = None)
    and a weight whose gradient is not wanted."""
    from cvpr23_lfdm_amd import autograd as A
    dev = backend
    x = rnd(rows, k, seed=1).requires_grad_(True)
    ws = [(rnd(n, k, seed=10 + j) * (k ** -0.5)).requires_grad_(j != 1 or len(ns) == 1) for j, n in enumerate(ns)]
    bs = [None if j == 2 else rnd(n, seed=50 + j).requires_grad_(True) for j, n in enumerate(ns)]
    fa = {0: lambda v: v, 1: F.silu, 2: F.gelu}[act]
    refs = [F.linear(fa(x), w, b) for w, b in zip(ws, bs)]
    gs = [rnd(rows, n, seed=90 + j) for j, n in enumerate(ns)]
    used = [j for j in range(len(ns)) if j != 3]
    sum((refs[j] * gs[j]).sum() for j in used).backward()
    xd = x.detach().to(dev).requires_grad_(True)
    wd = [w.detach().to(dev).requires_grad_(w.requires_grad) for w in ws]
    bd = [None if b is None else b.detach().to(dev).requires_grad_(True) for b in bs]
    ys = A.multi_linear(xd, wd, bd, act=act)
    for j, (y, r) in enumerate(zip(ys, refs)):
        sc = float(r.detach().abs().max())
        assert_close(y.detach().cpu() / sc, r.detach() / sc, TOL, "multi_linear y[%d]" % j)
    sum((ys[j] * gs[j].to(dev)).sum() for j in used).backward()
    sc = float(x.grad.abs().max())
    assert_close(xd.grad.cpu() / sc, x.grad / sc, TOL, "multi_linear dx")
    for j in used:
        if ws[j].requires_grad:
            sc = float(ws[j].grad.abs().max())
            assert_close(wd[j].grad.cpu() / sc, ws[j].grad / sc, TOL, "multi_linear dw[%d]" % j)
        else:
            assert wd[j].grad is None
        if bs[j] is not None:
            sc = float(bs[j].grad.abs().max())
            assert_close(bd[j].grad.cpu() / sc, bs[j].grad / sc, TOL, "multi_linear dbias[%d]" % j)
    if len(ns) > 3 and ws[3].requires_grad:        # the unused block: zero gradients, like autograd's
        assert float(wd[3].grad.abs().max()) == 0.0


@pytest.mark.parametrize("cin,cout,n,res", [(64, 64, 16, 8), (96, 40, 64, 4), (32, 128, 4, 16), (160, 64, 2, 12), (128, 136, 16, 8), (192, 64, 64, 4)])
def test_conv_wgrad_nine_tap_tiles(backend, cin, cout, n, res):
    """conv_wgrad3_kernel (3x3 / stride 1 / pad 1: all nine taps of a pixel tile per workgroup - or one filter row per workgroup where a
    convolution has one or two channel blocks; TW = 8 and the TW = 4 two-image form, partial channel tiles, ragged split counts) against autograd, in both output layouts, with the fused bias sums - and bit-identical to the per-tap
    kernel's summation order is NOT required (different tiling): tolerance as for the other weight-gradient tests."""
    import os
    dev = backend
    if big(dev):
        n, res = n * 4, res * 2
    x = rnd(n, cin, res, res, seed=1)
    w = (rnd(cout, cin, 3, 3, seed=2) * 0.1).requires_grad_(True)
    y = F.conv2d(x, w, padding=1)
    dy = rnd(*y.shape, seed=3)
    y.backward(dy)
    scale = float(w.grad.abs().max())
    ref_db = dy.sum(dim=(0, 2, 3))
    xd, dyd = to_cl(x).to(dev), to_cl(dy).to(dev)
    dw = train_ops.conv_wgrad(xd, dyd, n, res, res, res, res, 3, 3, stride=1, pad=(1, 1))
    got = dw.cpu().view(3, 3, cin, cout).permute(3, 2, 0, 1)
    assert_close(got / scale, w.grad / scale, TOL, "nine-tap wgrad, tap-major")
    out = torch.full((cout, cin, 3, 3), float("nan"), device=dev)
    db = torch.full((cout,), float("nan"), device=dev)
    c0 = (cin // 8) * 4
    train_ops.conv_wgrad(xd[:, :c0], dyd, n, res, res, res, res, 3, 3, stride=1, pad=(1, 1), out=out, ci_off=0, dbias=db)
    train_ops.conv_wgrad(xd[:, c0:], dyd, n, res, res, res, res, 3, 3, stride=1, pad=(1, 1), out=out, ci_off=c0)
    assert_close(out.cpu() / scale, w.grad / scale, TOL, "nine-tap wgrad, reference layout in two channel halves")
    assert_close(db.cpu() / float(ref_db.abs().max()), ref_db / float(ref_db.abs().max()), TOL, "bias sums from the nine-tap pass")
    os.environ["LFDM_WGRAD3"] = "0"
    try:
        old = train_ops.conv_wgrad(xd, dyd, n, res, res, res, res, 3, 3, stride=1, pad=(1, 1))
    finally:
        del os.environ["LFDM_WGRAD3"]
    assert_close(old.cpu() / scale, dw.cpu() / scale, TOL, "per-tap kernel vs nine-tap kernel")
