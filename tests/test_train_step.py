"""One full DM training step (FlowDiffusion.optimize_parameters: batched frozen-LFAE pseudo ground truth -> native UNet
forward/backward -> fused Adam) against the golden fixture recorded from the UNMODIFIED reference
(tests/golden/train_step_128.npz, oracle/make_golden.py --train).  GPU only: the step is ~3 TFLOP."""
import os

import numpy as np
import pytest
import torch

import synth
from util import assert_close, record_margin

GOLD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLD = os.path.join(GOLD_DIR, "train_step_128.npz")


def _build(dev, b, t, hw, null_cond_prob=0.0, **variant):
    from cvpr23_lfdm_amd import FlowDiffusion
    m = FlowDiffusion(img_size=hw // 4, num_frames=t, sampling_timesteps=5, timesteps=1000, null_cond_prob=null_cond_prob,
                      is_train=True, lr=1e-3, config_pth=synth.CONFIG, pretrained_pth="", **variant)
    m.unet.load_state_dict(synth.unet_state())
    m.generator.load_state_dict(synth.generator_state())
    m.region_predictor.load_state_dict(synth.region_state())
    m.bg_predictor.load_state_dict(synth.bg_state())
    for net in (m.generator, m.region_predictor, m.bg_predictor):
        net.eval()
        m.set_requires_grad(net, False)
    return m.to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["train_step_128", "train_step_128_resflow_p05"])
def test_training_step_matches_reference(monkeypatch, fixture):
    """(_resflow_p05: the reference's step with use_residual_flow=True and null_cond_prob=0.5 - the uniform draw of
    prob_mask_like (:55-61) replayed from tests/synth.py::null_uniform, B = 4.)"""
    g = np.load(os.path.join(GOLD_DIR, fixture + ".npz"))
    b, t, hw = int(g["b"]), int(g["t"]), int(g["hw"])
    dev = "cuda"
    variant = dict(null_cond_prob=0.5, use_residual_flow=True) if fixture.endswith("_resflow_p05") else {}
    m = _build(dev, b, t, hw, **variant)
    ref_img, real_vid, cond, tt, noise = synth.train_inputs(b, t, hw)
    m.diffusion.text_encoder = lambda texts: cond
    monkeypatch.setattr(torch, "randint", lambda *a, **k: tt.clone().to(k.get("device", "cpu")))
    monkeypatch.setattr(torch, "randn_like", lambda x, **k: noise.clone().to(x.device))
    if variant:
        u, orig_uniform = synth.null_uniform(b), torch.Tensor.uniform_
        monkeypatch.setattr(torch.Tensor, "uniform_", lambda self, *a, **k: self.copy_(u.to(self.device)) if tuple(self.shape) == (b,)
                            else orig_uniform(self, *a, **k))
    m.set_train_input(ref_img=ref_img.to(dev), real_vid=real_vid.to(dev), ref_text=[str(s) for s in g["labels"]])
    m.optimize_parameters()
    monkeypatch.undo()

    T = lambda k: torch.from_numpy(g[k])
    assert_close(m.real_vid_grid, T("real_vid_grid"), 1e-3, "pseudo-GT flow")
    assert_close(m.real_vid_conf, T("real_vid_conf"), 1e-3, "pseudo-GT occlusion")
    assert_close(m.real_out_vid[:, :, -1], T("real_out_vid"), 1e-3, "real_out_vid")
    assert_close(m.real_warped_vid[:, :, -1], T("real_warped_vid"), 1e-3, "real_warped_vid")
    assert_close(m.ref_img_fea[:, ::32, ::4, ::4], T("ref_img_fea_slice"), 1e-3, "ref_img_fea")
    assert bool((m.unet.null_cond_mask.cpu() == T("null_cond_mask")).all())
    assert_close(m.diffusion.pred_x0, T("pred_x0"), 1e-3, "pred_x0")
    assert_close(m.fake_out_vid[:, :, -1], T("fake_out_vid"), 1e-3, "fake_out_vid")
    assert_close(m.fake_warped_vid[:, :, -1], T("fake_warped_vid"), 1e-3, "fake_warped_vid")
    for k in ("loss", "rec_loss", "rec_warp_loss"):
        got, want = float(getattr(m, k)), float(g[k])
        record_margin(k, abs(got - want), abs(want), 1e-3)
        assert abs(got - want) <= 1e-3 * max(1.0, abs(want)), (k, got, want)

    names = [str(n) for n in g["names"]]
    params = dict(m.diffusion.named_parameters())
    assert set(names) == set(params)
    rng = np.random.Generator(np.random.PCG64(77))
    worst = ("", 0.0)
    for i, k in enumerate(names):
        p = params[k]
        gr = p.grad.detach().double().cpu()
        probe = torch.from_numpy(rng.standard_normal(p.numel())).view_as(gr)
        gn, want = float(gr.norm()), float(g["grad_norm"][i])
        rel = abs(gn - want) / (want + 1e-12)
        # the probe is a random projection: compare on the scale of the gradient norm
        relp = abs(float((gr * probe).sum()) - float(g["grad_probe"][i])) / (want * np.sqrt(p.numel()) + 1e-12)
        pn = abs(float(p.detach().double().norm()) - float(g["param_norm_after"][i])) / (float(g["param_norm_after"][i]) + 1e-12)
        for tag, e in (("grad norm", rel), ("grad probe", relp), ("updated weight norm", pn)):
            if e > worst[1]:
                worst = ("%s of %s" % (tag, k), e)
    record_margin("largest relative deviation over all parameter gradients / updated weights: " + worst[0], worst[1], 1.0, 1e-3)
    assert worst[1] < 1e-3, "largest deviation %.3e: %s" % (worst[1], worst[0])
    for key in g.files:
        if key.startswith("grad/"):
            want = T(key)
            assert_close(params[key[5:]].grad / (float(want.abs().max()) + 1e-12), want / (float(want.abs().max()) + 1e-12),
                         1e-3, key)
    # the optimizer is a real torch Optimizer: state_dict round trip + a second step run
    sd = m.optimizer_diff.state_dict()
    assert len(sd["state"]) == len(names) and float(sd["state"][0]["step"]) == 1.0
    m.optimizer_diff.load_state_dict(sd)
    m.optimize_parameters()
    assert float(m.optimizer_diff.state_dict()["state"][0]["step"]) == 2.0
    assert torch.isfinite(m.loss)


@pytest.mark.gpu
def test_functional_multigpu_flavour(monkeypatch):
    """DM/modules/video_flow_diffusion_model_multiGPU.py API: forward(real_vid, ref_img, ref_text) -> dict with an
    un-reduced loss whose mean equals the single-GPU class's loss; functional sample_one_video -> dict."""
    from DM.modules.video_flow_diffusion_model_multiGPU import FlowDiffusion as FD
    g = np.load(GOLD)
    b, t, hw = int(g["b"]), int(g["t"]), int(g["hw"])
    dev = "cuda"
    m = FD(img_size=hw // 4, num_frames=t, sampling_timesteps=3, timesteps=1000, null_cond_prob=0.0, is_train=True,
           config_pth=synth.CONFIG, pretrained_pth="")
    m.unet.load_state_dict(synth.unet_state())
    m.generator.load_state_dict(synth.generator_state())
    m.region_predictor.load_state_dict(synth.region_state())
    m.bg_predictor.load_state_dict(synth.bg_state())
    m.to(dev)
    ref_img, real_vid, cond, tt, noise = synth.train_inputs(b, t, hw)
    monkeypatch.setattr(torch, "randint", lambda *a, **k: tt.clone().to(k.get("device", "cpu")))
    monkeypatch.setattr(torch, "randn_like", lambda x, **k: noise.clone().to(x.device))
    out = m.forward(real_vid=real_vid.to(dev), ref_img=ref_img.to(dev), ref_text=cond.to(dev))
    monkeypatch.undo()
    assert out["loss"].shape == (b, 3, t, hw // 4, hw // 4) and out["null_cond_mask"].shape == (b,)
    assert out["rec_loss"].shape == real_vid.shape
    # labels of the fixture: second sample is "None" -> null cond there; here nothing is nulled, so only sample 0 matches
    assert_close(out["real_vid_grid"], torch.from_numpy(g["real_vid_grid"]), 1e-3, "pseudo-GT flow")
    out["loss"].mean().backward()
    assert all(p.grad is not None for p in m.diffusion.parameters())
    s = m.sample_one_video(sample_img=ref_img[:1].to(dev), sample_text=cond[:1].to(dev), cond_scale=1.0)
    assert s["sample_out_vid"].shape == (1, 3, t, hw, hw) and bool(torch.isfinite(s["sample_out_vid"]).all())


@pytest.mark.gpu
def test_lazy_real_decode_is_the_same_video():
    """FlowDiffusion.lazy_real_decode: real_out_vid / real_warped_vid (no loss reads them) are decoded when first read and
    equal what forward() computes eagerly; everything else of the step is untouched."""
    dev = "cuda"
    vids = {}
    for lazy in (False, True):
        m = _build(dev, 2, 4, 128)
        m.lazy_real_decode = lazy
        ref_img, real_vid, cond, tt, noise = synth.train_inputs(2, 4, 128)
        torch.manual_seed(5)
        m.set_train_input(ref_img=ref_img.to(dev), real_vid=real_vid.to(dev), ref_text=cond.to(dev))
        m.optimize_parameters()
        if lazy:
            assert m._real_out_vid is None and m._real_decode is not None          # nothing decoded yet
        vids[lazy] = (m.real_out_vid.clone(), m.real_warped_vid.clone(), m.fake_out_vid.clone(), float(m.loss))
        assert m._real_decode is None
    for a, b in zip(vids[False][:3], vids[True][:3]):
        assert torch.equal(a, b)
    assert vids[False][3] == vids[True][3]
