"""LFAE stage-1 training glue kernels (csrc/train_lfae.hip through cvpr23_lfdm_amd/lfae_ops.py) against the ATen ops the reference's
nn.Modules dispatch to (LFAE/modules/util.py:70-150, 217-264; generator.py:59-88; pixelwise_flow_predictor.py:95-102; model.py:118-122;
region_predictor.py:16-26), forward and backward through torch.autograd.  Dual backend like test_ops_parity.py: "emu" = the same kernel
sources on the x86 emulator (CPU), "hip" = MI355X."""
import pytest
import torch
import torch.nn.functional as F

from cvpr23_lfdm_amd import lfae_ops as L
from cvpr23_lfdm_amd import params as P
from util import assert_close, rnd

TOL = 2e-4


def _cl(x):
    return x.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("n,c,h,w,relu", [(2, 32, 5, 7, True), (3, 64, 16, 16, True), (2, 12, 9, 9, False), (2, 256, 4, 4, True),
                                          (2, 8, 48, 48, True), (1, 1024, 1, 2, True), (5, 96, 3, 3, False)])
def test_batchnorm_relu(backend, n, c, h, w, relu):
    dev = backend
    if dev == "cuda" and c == 8:
        n, c, h, w = 16, 64, 128, 128          # 262144 rows: 1024 chunks, 32 ticket groups
    x = (rnd(n, c, h, w, seed=1) * 1.7 + 0.3).requires_grad_(True)
    g = (rnd(c, seed=2) * 0.3 + 1.0).requires_grad_(True)
    b = (rnd(c, seed=3) * 0.2).requires_grad_(True)
    rm, rv = rnd(c, seed=4) * 0.1, rnd(c, seed=5).abs() + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm_ref, rv_ref, g, b, True, 0.1, 1e-5)
    y = F.relu(y) if relu else y
    dy = rnd(*y.shape, seed=6)
    y.backward(dy)
    xd = _cl(x.detach().to(dev)).requires_grad_(True)
    gd, bd = g.detach().to(dev).requires_grad_(True), b.detach().to(dev).requires_grad_(True)
    rmd, rvd = rm.to(dev), rv.to(dev)
    yd = L.BatchNormReLU.apply(xd, gd, bd, rmd, rvd, 0.1, 1e-5, relu)
    assert_close(yd.detach().cpu(), y.detach(), TOL, "bn y")
    assert_close(rmd.cpu(), rm_ref, TOL, "running_mean")
    assert_close(rvd.cpu(), rv_ref, TOL, "running_var")
    yd.backward(_cl(dy.to(dev)))
    sc = float(x.grad.abs().max())
    assert_close(xd.grad.cpu() / sc, x.grad / sc, TOL, "bn dx")
    sc = float(g.grad.abs().max())
    assert_close(gd.grad.cpu() / sc, g.grad / sc, TOL, "bn dgamma")
    sc = float(b.grad.abs().max())
    assert_close(bd.grad.cpu() / sc, b.grad / sc, TOL, "bn dbeta")
    # run-to-run identical (ticket folds in a fixed order) and the ticket words are left zeroed
    xd2 = xd.detach().clone().requires_grad_(True)
    yd2 = L.BatchNormReLU.apply(xd2, gd.detach(), bd.detach(), rm.to(dev), rv.to(dev), 0.1, 1e-5, relu)
    assert torch.equal(yd2.detach(), yd.detach())
    assert int(L._state(xd.device)["tickets"].abs().max()) == 0


@pytest.mark.parametrize("segments,n,c,h,w", [(3, 2, 32, 6, 5), (2, 3, 12, 9, 9), (3, 1, 256, 2, 2), (4, 2, 64, 16, 16)])
def test_batchnorm_relu_segments(backend, segments, n, c, h, w):
    """segments = S is S calls of the module on the S sub-batches: own batch statistics each, the running statistics updated S times in
    order, the parameter gradients summed (the region predictor on source / driving / transformed frames, model.py:157-160, :190-191)."""
    dev = backend
    if dev == "cuda" and c == 64:
        n, h, w = 16, 64, 64                      # 65536 rows per segment
    x = (rnd(segments * n, c, h, w, seed=1) * 1.7 + 0.3)
    x = (x + torch.arange(segments).repeat_interleave(n).view(-1, 1, 1, 1).float()).requires_grad_(True)      # different means per segment
    g = (rnd(c, seed=2) * 0.3 + 1.0).requires_grad_(True)
    b = (rnd(c, seed=3) * 0.2).requires_grad_(True)
    rm, rv = rnd(c, seed=4) * 0.1, rnd(c, seed=5).abs() + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = torch.cat([F.relu(F.batch_norm(xs, rm_ref, rv_ref, g, b, True, 0.1, 1e-5)) for xs in x.chunk(segments)])
    dy = rnd(*y.shape, seed=6)
    y.backward(dy)
    xd = _cl(x.detach().to(dev)).requires_grad_(True)
    gd, bd = g.detach().to(dev).requires_grad_(True), b.detach().to(dev).requires_grad_(True)
    rmd, rvd = rm.to(dev), rv.to(dev)
    yd = L.BatchNormReLU.apply(xd, gd, bd, rmd, rvd, 0.1, 1e-5, True, segments)
    assert_close(yd.detach().cpu(), y.detach(), TOL, "bn y")
    assert_close(rmd.cpu(), rm_ref, TOL, "running_mean")
    assert_close(rvd.cpu(), rv_ref, TOL, "running_var")
    yd.backward(_cl(dy.to(dev)))
    for name, got, want in (("dx", xd.grad, x.grad), ("dgamma", gd.grad, g.grad), ("dbeta", bd.grad, b.grad)):
        sc = float(want.abs().max())
        assert_close(got.cpu() / sc, want / sc, TOL, "bn " + name)
    # bit-identical to the separate calls
    rm1, rv1 = rm.to(dev), rv.to(dev)
    parts = [L.BatchNormReLU.apply(_cl(xs), gd.detach(), bd.detach(), rm1, rv1, 0.1, 1e-5, True) for xs in xd.detach().chunk(segments)]
    assert torch.equal(torch.cat(parts), yd.detach())
    assert torch.equal(rm1, rmd) and torch.equal(rv1, rvd)
    assert int(L._state(xd.device)["tickets"].abs().max()) == 0


def test_batchnorm_fork_and_channel_slices(backend):
    """fork=True: x comes back for the block's skip path and the skip's gradient is summed inside the backward kernel; inputs / gradients that
    are channel slices of a wider channels-last tensor (the halves of a torch.cat) are read where they lie (row stride > C)."""
    dev = backend
    n, c, h, w = 3, 16, 6, 5
    wide = _cl(rnd(n, c + 8, h, w, seed=1).to(dev))
    g = (rnd(c, seed=2) * 0.3 + 1.0).to(dev).requires_grad_(True)
    b = (rnd(c, seed=3) * 0.2).to(dev).requires_grad_(True)
    dyw = _cl(rnd(n, c + 8, h, w, seed=6).to(dev))
    grads = []
    for fork in (False, True):
        x = wide.clone().requires_grad_(True)
        g.grad = b.grad = None
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        xs = x[:, 8:]                                   # a channel slice: row stride c + 8
        assert L._rows_ld(xs.detach()).stride(0) == c + 8
        if fork:
            y, xa = L.BatchNormReLU.apply(xs, g, b, rm, rv, 0.1, 1e-5, True, 1, True)
        else:
            y, xa = L.BatchNormReLU.apply(xs, g, b, rm, rv, 0.1, 1e-5, True), xs
        out = y * 0.7 + xa * 1.3
        out.backward(dyw[:, 8:])
        grads.append((x.grad.clone(), g.grad.clone(), b.grad.clone(), y.detach().clone()))
    ref = F.relu(F.batch_norm(wide[:, 8:].cpu(), None, None, g.detach().cpu(), b.detach().cpu(), True, 0.1, 1e-5))
    assert_close(grads[0][3].cpu(), ref, TOL, "bn y of a channel slice")
    for a, f, what in zip(grads[0], grads[1], ("dx", "dgamma", "dbeta", "y")):
        sc = float(a.abs().max())
        assert_close(f.cpu() / sc, a.cpu() / sc, TOL, "bn fork " + what)


def _antialias_ref(x, weight, scale):
    ks = weight.shape[-1]
    ka = ks // 2
    kb = ka - 1 if ks % 2 == 0 else ka
    out = F.conv2d(F.pad(x, (ka, kb, ka, kb)), weight=weight, groups=x.shape[1])
    s = int(1 / scale)
    return out[:, :, ::s, ::s]


@pytest.mark.parametrize("scale,h,w,rows4,affine", [(0.25, 32, 32, False, False), (0.5, 20, 28, True, True), (0.125, 32, 32, True, False),
                                                    (0.25, 17, 23, False, True)])
def test_blur_down(backend, scale, h, w, rows4, affine):
    dev = backend
    n, c = (2, 3) if dev == "cpu" else (8, 3)
    if dev == "cuda":
        h, w = h * 4, w * 4
    x = rnd(n, c, h, w, seed=1).requires_grad_(True)
    weight = P.antialias_kernel(c, scale)
    sc = (rnd(c, seed=2).abs() + 0.5) if affine else None
    bi = rnd(c, seed=3) if affine else None
    ref = _antialias_ref(x, weight, scale)
    if affine:
        ref = ref * sc.view(1, c, 1, 1) + bi.view(1, c, 1, 1)
    dy = rnd(*ref.shape, seed=4)
    ref.backward(dy)
    # a strided input: the 3 real channels of 4-channel rows (what a padded convolution output looks like)
    x4 = torch.zeros(n, h, w, 4)
    x4[..., :c] = x.detach().permute(0, 2, 3, 1)
    xd = x4.to(dev).permute(0, 3, 1, 2)[:, :c].requires_grad_(True)
    out = L.BlurDown.apply(xd, weight.to(dev), int(1 / scale), rows4, None if sc is None else sc.to(dev), None if bi is None else bi.to(dev))
    if rows4:
        assert out.shape[1] == 4 and float(out[:, c:].abs().max()) == 0.0 and out.permute(0, 2, 3, 1).is_contiguous()
    assert_close(out[:, :c].detach().cpu(), ref.detach(), 1e-5, "blur out")
    dyd = torch.zeros_like(out)
    dyd[:, :c] = dy.to(dev)
    out.backward(dyd)
    s = float(x.grad.abs().max())
    assert_close(xd.grad.cpu() / s, x.grad / s, 1e-5, "blur dx")


def _apply_optical_ref(src, prev, flow, occ):
    h, w = src.shape[2:]
    if flow.shape[1] != h or flow.shape[2] != w:
        flow = F.interpolate(flow.permute(0, 3, 1, 2), size=(h, w), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    out = F.grid_sample(src, flow, align_corners=False)
    if occ is not None:
        if occ.shape[2:] != out.shape[2:]:
            occ = F.interpolate(occ, size=out.shape[2:], mode="bilinear", align_corners=False)
        out = out * occ + prev * (1 - occ) if prev is not None else out * occ
    return out


def _flow(n, fh, fw, seed, amp=0.25):
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, fh), torch.linspace(-1, 1, fw), indexing="ij")
    ident = torch.stack((xs, ys), -1).view(1, fh, fw, 2)
    return (ident + amp * rnd(n, fh, fw, 2, seed=seed)).contiguous()


@pytest.mark.parametrize("c,h,w,fh,fw,with_prev,with_occ", [(8, 8, 8, 8, 8, False, True), (16, 16, 12, 8, 6, True, True),
                                                            (64, 8, 8, 4, 4, True, True), (256, 4, 6, 4, 6, False, True),
                                                            (32, 8, 8, 8, 8, False, False), (128, 16, 16, 4, 4, True, True)])
def test_apply_optical_cl(backend, c, h, w, fh, fw, with_prev, with_occ):
    dev = backend
    n = 2
    if dev == "cuda":
        n, h, w, fh, fw = 4, h * 4, w * 4, fh * 4, fw * 4
    src = rnd(n, c, h, w, seed=1).requires_grad_(True)
    prev = rnd(n, c, h, w, seed=2).requires_grad_(True) if with_prev else None
    flow = _flow(n, fh, fw, 3, amp=0.6).requires_grad_(True)          # large enough that taps leave the image
    occ = torch.sigmoid(rnd(n, 1, fh, fw, seed=4)).requires_grad_(True) if with_occ else None
    ref = _apply_optical_ref(src, prev, flow, occ)
    dy = rnd(*ref.shape, seed=5)
    ref.backward(dy)
    sd = _cl(src.detach().to(dev)).requires_grad_(True)
    pd = None if prev is None else _cl(prev.detach().to(dev)).requires_grad_(True)
    fd = flow.detach().to(dev).requires_grad_(True)
    od = None if occ is None else occ.detach().to(dev).requires_grad_(True)
    out = L.ApplyOpticalCL.apply(sd, pd, L._maps_planar(fd, od))
    assert_close(out.detach().cpu(), ref.detach(), TOL, "apply_optical out")
    out.backward(_cl(dy.to(dev)))
    for name, got, want in (("dsrc", sd.grad, src.grad), ("dprev", None if pd is None else pd.grad, None if prev is None else prev.grad),
                            ("dflow", fd.grad, flow.grad), ("docc", None if od is None else od.grad, None if occ is None else occ.grad)):
        if want is None:
            continue
        s = float(want.abs().max())
        assert_close(got.cpu() / s, want / s, TOL, "apply_optical " + name)
    # the fixed-point scatter: bit-identical from run to run, workspace handed back zeroed
    sd2 = sd.detach().clone().requires_grad_(True)
    out2 = L.ApplyOpticalCL.apply(sd2, None if pd is None else pd.detach(), L._maps_planar(fd.detach(), None if od is None else od.detach()))
    out2.backward(_cl(dy.to(dev)))
    assert torch.equal(sd2.grad, sd.grad)
    st = L._state(sd.device)
    assert int(st["fix"].abs().max()) == 0 and int(st["amax"].abs().max()) == 0


def test_apply_optical_image(backend):
    dev = backend
    n, c, h, w, fh, fw = (2, 3, 16, 16, 4, 4) if dev == "cpu" else (8, 3, 128, 128, 32, 32)
    src = rnd(n, c, h, w, seed=1).abs()
    prev = torch.sigmoid(rnd(n, c, h, w, seed=2)).requires_grad_(True)
    flow = _flow(n, fh, fw, 3, amp=0.5).requires_grad_(True)
    occ = torch.sigmoid(rnd(n, 1, fh, fw, seed=4)).requires_grad_(True)
    ref = _apply_optical_ref(src, prev, flow, occ)
    dy = rnd(*ref.shape, seed=5)
    ref.backward(dy)
    # prev as the convolution hands it over: 3 of 4 channels of channels-last rows
    p4 = torch.zeros(n, h, w, 4)
    p4[..., :c] = prev.detach().permute(0, 2, 3, 1)
    pd = p4.to(dev).permute(0, 3, 1, 2)[:, :c].requires_grad_(True)
    fd, od = flow.detach().to(dev).requires_grad_(True), occ.detach().to(dev).requires_grad_(True)
    out = L.ApplyOpticalImage.apply(src.to(dev), pd, L._maps_planar(fd, od))
    assert_close(out.detach().cpu(), ref.detach(), TOL, "apply_optical_image out")
    out.backward(dy.to(dev))
    for name, got, want in (("dprev", pd.grad, prev.grad), ("dflow", fd.grad, flow.grad), ("docc", od.grad, occ.grad)):
        s = float(want.abs().max())
        assert_close(got.cpu() / s, want / s, TOL, "apply_optical_image " + name)


@pytest.mark.parametrize("reflection,n_div", [(False, 1), (False, 3), (True, 1)])
def test_grid_sample(backend, reflection, n_div):
    dev = backend
    nb, c, h, w, ho, wo = (2, 3, 9, 11, 7, 8) if dev == "cpu" else (8, 3, 128, 128, 128, 128)
    x = rnd(nb, c, h, w, seed=1)
    n = nb * n_div
    grid = _flow(n, ho, wo, 2, amp=0.9 if reflection else 0.5).requires_grad_(not reflection)
    xr = x.unsqueeze(1).repeat(1, n_div, 1, 1, 1).view(n, c, h, w)
    ref = F.grid_sample(xr, grid, padding_mode="reflection" if reflection else "zeros", align_corners=False)
    gd = grid.detach().to(dev).requires_grad_(not reflection)
    out = L.GridSample.apply(x.to(dev), gd, n_div, reflection)
    assert_close(out.detach().cpu(), ref.detach(), TOL, "grid_sample out")
    if not reflection:
        dy = rnd(*ref.shape, seed=3)
        ref.backward(dy)
        out.backward(dy.to(dev))
        s = float(grid.grad.abs().max())
        assert_close(gd.grad.cpu() / s, grid.grad / s, TOL, "grid_sample dgrid")


def test_svd2x2_sym_autograd(backend):
    dev = backend
    n = 500
    l = rnd(n, 2, 2, seed=1)
    a = (l @ l.transpose(1, 2) + 0.05 * torch.eye(2)).requires_grad_(True)
    u, s, _ = torch.svd(a)
    gu, gs = rnd(n, 2, 2, seed=2), rnd(n, 2, seed=3)
    (u * gu).sum().add((s * gs).sum()).backward()
    ad = a.detach().to(dev).requires_grad_(True)
    ud, sd = L.Svd2x2Sym.apply(ad)
    assert_close(ud.detach().cpu(), u.detach(), 1e-4, "svd u")
    assert_close(sd.detach().cpu(), s.detach(), 1e-4, "svd s")
    (ud * gu.to(dev)).sum().add((sd * gs.to(dev)).sum()).backward()
    # ill-conditioned pairs (s0 ~ s1) amplify fp32 rounding in both implementations: compare where the gap is healthy
    ok = ((s.detach()[:, 0] - s.detach()[:, 1]) / s.detach()[:, 0]) > 0.05
    sc = float(a.grad[ok].abs().max())
    assert_close(ad.grad.cpu()[ok] / sc, a.grad[ok] / sc, 5e-4, "svd backward")


@pytest.mark.parametrize("kind", ["avg", "max", "up"])
def test_pool2(backend, kind):
    dev = backend
    n, c, h, w = (2, 8, 6, 10) if dev == "cpu" else (8, 64, 128, 128)
    x = rnd(n, c, h, w, seed=1)
    if kind == "max":
        x[:, :, ::2, ::2] = x[:, :, 1::2, 1::2]          # ties inside every window: the FIRST maximum takes the gradient
    x.requires_grad_(True)
    ref = {"avg": lambda: F.avg_pool2d(x, 2), "max": lambda: F.max_pool2d(x, 2), "up": lambda: F.interpolate(x, scale_factor=2)}[kind]()
    dy = rnd(*ref.shape, seed=2)
    ref.backward(dy)
    xd = _cl(x.detach().to(dev)).requires_grad_(True)
    out = L.pool2(xd, kind)
    assert isinstance(out.grad_fn, torch.autograd.function.BackwardCFunction)
    assert_close(out.detach().cpu(), ref.detach(), 1e-6, "pool2 %s" % kind)
    out.backward(_cl(dy.to(dev)))
    assert_close(xd.grad.cpu(), x.grad, 1e-6, "pool2 %s backward" % kind)


def test_l1_mean_and_conv_relu(backend):
    from cvpr23_lfdm_amd import lfae_train
    dev = backend
    n, c, h, w = (2, 8, 6, 6) if dev == "cpu" else (8, 64, 64, 64)
    x = rnd(n, c, h, w, seed=1).requires_grad_(True)
    y = rnd(n, c, h, w, seed=2)
    wgt = (rnd(c, c, 3, 3, seed=3) * 0.2).requires_grad_(True)
    b = rnd(c, seed=4).requires_grad_(True)
    ref = 10.0 * torch.abs(F.relu(F.conv2d(x, wgt, b, padding=1)) - y).mean()
    ref.backward()
    xd = _cl(x.detach().to(dev)).requires_grad_(True)
    wd, bd = wgt.detach().to(dev).requires_grad_(True), b.detach().to(dev).requires_grad_(True)
    feat = lfae_train.conv2d(xd, wd, bd, 1, relu=True)
    loss = L.L1Mean.apply(feat, _cl(y.to(dev)), 10.0)
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    loss.sum().backward()
    for name, got, want in (("dx", xd.grad, x.grad), ("dw", wd.grad, wgt.grad), ("db", bd.grad, b.grad)):
        s = float(want.abs().max())
        assert_close(got.cpu() / s, want / s, TOL, "conv+relu+l1 " + name)
    # run-to-run identical, ticket word left zeroed
    assert float(L.L1Mean.apply(feat.detach(), _cl(y.to(dev)), 10.0)) == float(loss)
    assert int(L._state(xd.device)["amax"].abs().max()) == 0


@pytest.mark.parametrize("cin,cout,k", [(4, 64, 7), (64, 4, 7), (8, 32, 3), (12, 12, 5), (16, 4, 3)])
def test_thin_channel_weight_gradient(backend, cin, cout, k):
    """ConvCL's im2col route for convolutions with <= 16 input or output channels (autograd._thin_wgrad) against F.conv2d's weight gradient."""
    from cvpr23_lfdm_amd import autograd as A
    from util import to_cl
    dev = backend
    n, h, w = (2, 9, 11) if dev == "cpu" else (8, 128, 128)
    pad = k // 2
    x = rnd(n, cin, h, w, seed=1).requires_grad_(True)
    wgt = (rnd(cout, cin, k, k, seed=2) * 0.1).requires_grad_(True)
    b = rnd(cout, seed=3).requires_grad_(True)
    y = F.conv2d(x, wgt, b, padding=pad)
    dy = rnd(*y.shape, seed=4)
    y.backward(dy)
    assert A._thin_wgrad_route("conv", None, 1, (pad, pad), k, k, cin, cout, h, w, h, w) is not None
    xd = to_cl(x.detach()).to(dev).requires_grad_(True)
    wd, bd = wgt.detach().to(dev).requires_grad_(True), b.detach().to(dev).requires_grad_(True)
    yd = A.conv_cl(xd, wd, bd, n_img=n, hi=h, wi=w, pad=(pad, pad))
    assert_close(yd.detach().cpu(), to_cl(y.detach()), TOL, "thin conv y")
    yd.backward(to_cl(dy).to(dev))
    for name, got, want in (("dw", wd.grad, wgt.grad), ("db", bd.grad, b.grad), ("dx", xd.grad, to_cl(x.grad))):
        s = float(want.abs().max())
        assert_close(got.cpu() / s, want / s, TOL, "thin conv " + name)
