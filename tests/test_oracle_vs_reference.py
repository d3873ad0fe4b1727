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
