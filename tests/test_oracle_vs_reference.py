"""Live check of the oracle against the unmodified reference (only where /root/reference exists - i.e. in
the build container; skipped on the GPU box, where the committed fixtures of tests/golden/ stand in)."""
import pytest
import torch

import reference_loader

pytestmark = pytest.mark.skipif(not reference_loader.reference_available(), reason="reference tree not present")


def test_unet_and_sampler_match_live_reference():
    import lfdm_oracle as O
    import synth
    ref = reference_loader.load_reference()
    b, t, s = 1, 2, 8
    m = ref.vfdm.FlowDiffusion(img_size=s, num_frames=t, sampling_timesteps=3, is_train=False,
                               config_pth=synth.CONFIG, pretrained_pth="")
    m.unet.load_state_dict(synth.unet_state(seed=99))
    m.generator.load_state_dict(synth.generator_state(seed=98))
    m.eval()
    dsd = {"denoise_fn." + k: v for k, v in synth.unet_state(seed=99).items()}
    x, time, cond = synth.unet_inputs(b, t, s, seed=12)
    with torch.no_grad():
        want = m.unet(x, time, cond=cond, null_cond_prob=0.)
    got = O.unet_forward(dsd, x, time, cond)
    assert float((got - want).abs().max()) < 1e-5
    # drop-in import path resolves to the reference here (reference root precedes the repo root)
    import DM.modules.video_flow_diffusion_model as mod
    assert mod.__file__.startswith(reference_loader.REFERENCE_ROOT)


@pytest.mark.parametrize("learn_null_cond", [False, True])
def test_optimizer_state_of_the_reference_loads_into_flat_adam(learn_null_cond):
    """torch.optim state dicts index parameters by POSITION: with the reference's registration order mirrored
    (params.unet_spec) an `optimizer_diff` saved by the reference's training script restores into FlatAdam with every
    moment on the right parameter (train_video_flow_diffusion_mug.py:181 `model.optimizer_diff.load_state_dict`)."""
    import synth
    from cvpr23_lfdm_amd import FlowDiffusion
    ref = reference_loader.load_reference()
    # (learn_null_cond: the root-level null_cond_emb parameter comes FIRST in the reference's named_parameters())
    kw = dict(img_size=8, num_frames=2, sampling_timesteps=5, is_train=True, config_pth=synth.CONFIG, pretrained_pth="",
              learn_null_cond=learn_null_cond)
    rm, om = ref.vfdm.FlowDiffusion(**kw), FlowDiffusion(**kw)
    rnames = [k for k, _ in rm.diffusion.named_parameters()]
    assert rnames == [k for k, _ in om.diffusion.named_parameters()]
    for i, p in enumerate(rm.diffusion.parameters()):            # a recognisable moment per parameter
        rm.optimizer_diff.state[p] = {"step": torch.tensor(3.0), "exp_avg": torch.full_like(p, float(i + 1)),
                                      "exp_avg_sq": torch.full_like(p, 0.5 * (i + 1))}
    sd = rm.optimizer_diff.state_dict()
    om.optimizer_diff.load_state_dict(sd)
    got = om.optimizer_diff.state_dict()
    assert got["param_groups"][0]["betas"] == sd["param_groups"][0]["betas"] and len(got["state"]) == len(rnames)
    for i, p in enumerate(om.diffusion.parameters()):
        st = om.optimizer_diff.state[p]
        assert st["exp_avg"].shape == p.shape and float(st["exp_avg"].flatten()[0]) == float(i + 1) and float(st["step"]) == 3.0
    # the moments are folded into the flat buffers at the first use: every one must have its parameter's size
    om.optimizer_diff.ensure_flat()
    # ... and a state saved for another parameter order is refused with a clear message instead of being mis-assigned
    bad = rm.optimizer_diff.state_dict()
    bad["state"][0], bad["state"][1] = bad["state"][1], bad["state"][0]
    om2 = FlowDiffusion(**kw)
    om2.optimizer_diff.load_state_dict(bad)
    with pytest.raises(ValueError, match="parameter ORDER"):
        om2.optimizer_diff.ensure_flat()
