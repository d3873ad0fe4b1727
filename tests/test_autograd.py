"""torch.autograd.Function wrappers (cvpr23_lfdm_amd/autograd.py) against torch autograd of the ATen graph the
reference builds.  backend = emu (CPU) / hip (gpu)."""
import pytest
import torch
import torch.nn.functional as F

from cvpr23_lfdm_amd import autograd as A
from util import assert_close, from_cl, rnd, to_cl

TOL = 3e-4


def _cmp(got, ref, what):
    sc = float(ref.abs().max()) + 1e-12
    assert_close(got.detach().cpu() / sc, ref.detach() / sc, TOL, what)


@pytest.mark.parametrize("case", ["conv3_cat_res", "linear", "down", "deconv", "stem4"])
def test_conv_function(backend, case):
    dev = backend
    n, h = 2, 6
    if case == "conv3_cat_res":
        c0, c1, co, k = 8, 8, 12, 3
        x0, x1 = rnd(n, c0, h, h, seed=1).requires_grad_(True), rnd(n, c1, h, h, seed=2).requires_grad_(True)
        w = (rnd(co, c0 + c1, 1, k, k, seed=3) * 0.1).requires_grad_(True)
        b = rnd(co, seed=4).requires_grad_(True)
        res = rnd(n, co, h, h, seed=5).requires_grad_(True)
        ref = F.conv2d(torch.cat((x0, x1), 1), w[:, :, 0], b, padding=1) + res
        args = dict(x1=x1, residual=res)
    elif case == "linear":
        c0, co, k = 16, 24, 1
        x0, x1, res = rnd(n, c0, h, h, seed=1).requires_grad_(True), None, None
        w = (rnd(co, c0, seed=3) * 0.1).requires_grad_(True)
        b = None
        ref = F.conv2d(x0, w[:, :, None, None])
        args = {}
    elif case == "down":
        c0, co, k = 8, 8, 4
        x0, x1, res = rnd(n, c0, h, h, seed=1).requires_grad_(True), None, None
        w = (rnd(co, c0, 1, 4, 4, seed=3) * 0.1).requires_grad_(True)
        b = rnd(co, seed=4).requires_grad_(True)
        ref = F.conv2d(x0, w[:, :, 0], b, stride=2, padding=1)
        args = dict(stride=2, pad=(1, 1))
    elif case == "deconv":
        c0, co, k = 8, 12, 4
        x0, x1, res = rnd(n, c0, h, h, seed=1).requires_grad_(True), None, None
        w = (rnd(c0, co, 1, 4, 4, seed=3) * 0.1).requires_grad_(True)
        b = rnd(co, seed=4).requires_grad_(True)
        ref = F.conv_transpose2d(x0, w[:, :, 0], b, stride=2, padding=1)
        args = dict(kind="deconv")
    else:
        c0, co, k = 4, 16, 7
        x0, x1, res = rnd(n, c0, h, h, seed=1), None, None
        w = (rnd(co, c0, 1, 7, 7, seed=3) * 0.1).requires_grad_(True)
        b = rnd(co, seed=4).requires_grad_(True)
        ref = F.conv2d(x0, w[:, :, 0], b, padding=3)
        args = dict(pad=(3, 3))
    dy = rnd(*ref.shape, seed=9)
    ref.backward(dy)
    leaf = lambda t: None if t is None else t.detach().clone().to(dev).requires_grad_(t.requires_grad)
    cl = lambda t: None if t is None else to_cl(t.detach()).to(dev).requires_grad_(t.requires_grad)
    kx0, kx1, kres, kw, kb = cl(x0), cl(x1), cl(res), leaf(w), leaf(b)
    kargs = dict(args)
    if "x1" in kargs:
        kargs["x1"], kargs["residual"] = kx1, kres
    y = A.conv_cl(kx0, kw, kb, n_img=n, hi=h, wi=h, **kargs)
    ho = ref.shape[2]
    _cmp(from_cl(y, n, ho, ho), ref, case + " fwd")
    y.backward(to_cl(dy).to(dev))
    _cmp(kw.grad, w.grad, case + " dW")
    if b is not None:
        _cmp(kb.grad, b.grad, case + " db")
    if x0.requires_grad:
        _cmp(from_cl(kx0.grad, n, h, h), x0.grad, case + " dx0")
    if x1 is not None:
        _cmp(from_cl(kx1.grad, n, h, h), x1.grad, case + " dx1")
        _cmp(from_cl(kres.grad, n, h, h), res.grad, case + " dres")


@pytest.mark.parametrize("reflect", [True, False])
def test_upsample2_pad_function(backend, reflect):
    dev = backend
    n, c, h, w = 2, 8, 3, 5
    x = rnd(n, c, h, w, seed=1).requires_grad_(True)
    up = F.interpolate(x, scale_factor=2, mode="nearest")
    ref = F.pad(up, (1, 1, 1, 1), mode="reflect" if reflect else "constant")
    dy = rnd(*ref.shape, seed=2)
    ref.backward(dy)
    kx = to_cl(x.detach()).to(dev).requires_grad_(True)
    y = A.Upsample2Pad.apply(kx, n, h, w, 1, reflect)
    _cmp(from_cl(y, n, 2 * h + 2, 2 * w + 2), ref, "upsample+pad fwd")
    y.backward(to_cl(dy).to(dev))
    _cmp(from_cl(kx.grad, n, h, w), x.grad, "upsample+pad bwd")


def test_filter_pack_cache_follows_the_weights(backend):
    """autograd._pack_wino caches the Winograd packs of LEAF weights (LFAE stage-1 training: the frozen VGG-19 convolves 12 times per step with
    the same filters).  The cache must notice every way a weight can change: an in-place torch write (`_version`), a raw-pointer optimizer
    step (params.weights_epoch) - and must never serve a pack to a different tensor that happens to live at the same address."""
    from cvpr23_lfdm_amd import params as P
    dev = backend
    n, h, c = 1, 4, 16
    x = to_cl(rnd(n, c, h, h, seed=1)).to(dev)

    def run(w):
        return A.conv_cl(x, w, None, n_img=n, hi=h, wi=h).detach().cpu()

    w = torch.nn.Parameter((rnd(c, c, 3, 3, seed=2) * 0.1).to(dev), requires_grad=False)       # frozen leaf: cached
    y0 = run(w)
    assert torch.equal(run(w), y0)
    with torch.no_grad():
        w.mul_(2.0)                                        # torch write: _version moves
    assert_close(run(w), 2 * y0, 1e-5, "pack rebuilt after an in-place write")
    wt = torch.nn.Parameter((rnd(c, c, 3, 3, seed=3) * 0.1).to(dev))                            # trainable leaf
    y1 = run(wt)
    with torch.no_grad():
        wt.copy_((rnd(c, c, 3, 3, seed=4) * 0.1).to(dev))                                     # what load_state_dict does (an in-place write THROUGH
    y2 = run(wt)                                                                              # `.data` is invisible to torch's counter: INTEGRATION.md)
    ref2 = F.conv2d(from_cl(x.cpu(), n, h, h), wt.detach().cpu(), padding=1)
    assert_close(from_cl(y2, n, h, h), ref2, TOL, "pack rebuilt after copy_")
    before = P.weights_epoch()
    P.bump_weights_epoch()                                 # what FlatAdam.step does after its raw-pointer update
    assert P.weights_epoch() == before + 1
    assert_close(from_cl(run(wt), n, h, h), ref2, TOL, "same weights, new epoch: same result")
    del w, wt                                               # freed tensors: a new parameter at a recycled address must not hit their packs
    for seed in range(5, 9):
        wn = torch.nn.Parameter((rnd(c, c, 3, 3, seed=seed) * 0.1).to(dev), requires_grad=False)
        ref = F.conv2d(from_cl(x.cpu(), n, h, h), wn.detach().cpu(), padding=1)
        assert_close(from_cl(run(wn), n, h, h), ref, TOL, "fresh parameter %d" % seed)
        del wn
    assert not torch.equal(y1, y2)


def test_repack_stale_batches_the_winograd_packs(backend):
    """autograd.repack_stale: after an optimizer step every cached Winograd pack of a trainable weight (forward and data-gradient forms, incl. the
    input-channel slices of a convolution over cat(x, skip)) is rebuilt by ONE multi-filter launch into its existing tensor - the results equal
    fresh single packs, and a step that follows needs no per-filter pack launch."""
    import torch
    from cvpr23_lfdm_amd import autograd as A
    from cvpr23_lfdm_amd import ops, params as P
    from util import rnd, to_cl
    dev = backend
    A._PACKS.clear()
    w1 = (rnd(32, 48, 3, 3, seed=1) * 0.1).to(dev).requires_grad_(True)          # conv over cat(x0 (32 ch), x1 (16 ch))
    w2 = (rnd(16, 32, 3, 3, seed=2) * 0.1).to(dev).requires_grad_(True)
    x0, x1 = to_cl(rnd(2, 32, 8, 8, seed=3)).to(dev).requires_grad_(True), to_cl(rnd(2, 16, 8, 8, seed=4)).to(dev).requires_grad_(True)

    def step():
        h = A.conv_cl(x0, w1, None, x1=x1, n_img=2, hi=8, wi=8)
        y = A.conv_cl(h, w2, None, n_img=2, hi=8, wi=8)
        for t in (w1, w2, x0, x1):
            t.grad = None
        y.sum().backward()
        return y.detach().clone(), w1.grad.clone(), x0.grad.clone(), x1.grad.clone()

    ref = step()
    n_entries = len(A._PACKS)
    assert n_entries == 5                                   # w1 fwd, w1 dgrad x 2 parts, w2 fwd, w2 dgrad
    with torch.no_grad():                                    # what FlatAdam.step does: raw writes + an epoch bump
        w1.data.mul_(1.5)
        w2.data.add_(0.01)
    w1._version, w2._version                                 # (data writes do not touch the version counters)
    P.bump_weights_epoch()
    packs_before = {k: v[2].data_ptr() for k, v in A._PACKS.items()}
    assert A.repack_stale() == 5
    assert {k: v[2].data_ptr() for k, v in A._PACKS.items()} == packs_before       # rebuilt in place
    calls = []
    orig = ops.pack_wino_weight
    ops.pack_wino_weight = lambda *a, **k: calls.append(1) or orig(*a, **k)
    try:
        got = step()
    finally:
        ops.pack_wino_weight = orig
    assert not calls, "a step after repack_stale() still packed %d filters one by one" % len(calls)
    A._PACKS.clear()
    want = step()                                            # the same step on fresh single packs
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert not torch.equal(got[0], ref[0])
    assert A.repack_stale() == 0


@pytest.mark.parametrize("two_sources,wino", [(False, False), (True, False), (False, True), (True, True)])
def test_conv_fork_sums_both_gradients_in_the_dgrad(backend, two_sources, wino):
    """geom fork=True: the inputs come back as outputs for their second reader; the gradient equals the unforked graph's (where the
    autograd engine adds the two contributions)."""
    dev = backend
    n, h = 2, 8
    c0, c1, co = (16, 16, 32) if wino else (8, 12, 12)
    mk = lambda *s, seed: rnd(*s, seed=seed).to(dev)
    w = (mk(co, c0 + (c1 if two_sources else 0), 1, 3, 3, seed=3) * 0.1).requires_grad_(True)
    wr = (mk(co, c0 + (c1 if two_sources else 0), seed=4) * 0.1).requires_grad_(True)        # the block's 1x1 res_conv
    b = mk(co, seed=5).requires_grad_(True)
    dy = to_cl(mk(n, co, h, h, seed=9))
    grads = []
    for fork in (False, True):
        x0 = to_cl(mk(n, c0, h, h, seed=1)).requires_grad_(True)
        x1 = to_cl(mk(n, c1, h, h, seed=2)).requires_grad_(True) if two_sources else None
        for p in (w, wr, b):
            p.grad = None
        if fork:
            out = A.conv_cl(x0, w, b, x1=x1, n_img=n, hi=h, wi=h, fork=True)
            hmid, xa, xb = out[0], out[1], (out[2] if two_sources else None)
        else:
            hmid, xa, xb = A.conv_cl(x0, w, b, x1=x1, n_img=n, hi=h, wi=h), x0, x1
        y = A.conv_cl(xa, wr, None, x1=xb, residual=hmid, n_img=n, hi=h, wi=h)
        y.backward(dy)
        grads.append([x0.grad.clone(), None if x1 is None else x1.grad.clone(), w.grad.clone(), wr.grad.clone(), b.grad.clone()])
    for a, f, what in zip(grads[0], grads[1], ("dx0", "dx1", "dw", "dw_res", "db")):
        if a is not None:
            _cmp(f, a.cpu(), "fork " + what)


@pytest.mark.parametrize("c", [64, 128, 96])
def test_layernorm_fork_sums_the_residual_gradient(backend, c):
    dev = backend
    rows = 70
    gamma = (rnd(c, seed=2).to(dev) * 0.2 + 1.0).requires_grad_(True)
    dy = rnd(rows, c, seed=9).to(dev)
    grads = []
    for fork in (False, True):
        x = rnd(rows, c, seed=1).to(dev).requires_grad_(True)
        gamma.grad = None
        if fork:
            normed, xa = A.LayerNormCL.apply(x, gamma, True)
        else:
            normed, xa = A.LayerNormCL.apply(x, gamma), x
        y = normed * 0.7 + xa * 1.3
        y.backward(dy)
        grads.append((x.grad.clone(), gamma.grad.clone()))
    _cmp(grads[1][0], grads[0][0].cpu(), "fork dx")
    _cmp(grads[1][1], grads[0][1].cpu(), "fork dgamma")
