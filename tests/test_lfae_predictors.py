"""The frozen LFAE region predictor as the DM training step runs it (cvpr23_lfdm_amd.lfae_predictors.RegionPredictorExec:
batched frames, native convolutions, the device-side LAPACK-convention 2x2 SVD) against the CPU oracle's restatement of
LFAE/modules/region_predictor.py:52-117 (which calls torch.svd on the host like the reference)."""
import pytest
import torch

import lfdm_oracle as O
import synth
from util import assert_close


@pytest.mark.parametrize("pad", [3, 0], ids=["mug", "mhad_natops"])
def test_region_predictor(backend, pad):
    """pad = 3: config/mug128.yaml; pad = 0: mhad128 / natops128 (unpadded `regions` head, 26x26 heat-maps)."""
    dev = backend
    if pad == 0 and dev != "cuda":
        pytest.skip("second config runs on the GPU only (the x86 emulator needs ~45 s per predictor pass)")
    from cvpr23_lfdm_amd import FlowDiffusion
    cfg = synth.CONFIG if pad == 3 else synth.CONFIG.replace("lfae_128.yaml", "lfae_128_pad0.yaml")
    m = FlowDiffusion(img_size=32, num_frames=2, sampling_timesteps=5, timesteps=1000, null_cond_prob=0.0,
                      is_train=False, config_pth=cfg, pretrained_pth="")
    rsd = synth.region_state()
    m.region_predictor.load_state_dict(rsd)
    net = m.region_predictor.to(dev).eval()
    g = torch.Generator().manual_seed(77)
    n = 3 if dev == "cuda" else 1                       # the x86 emulator runs the five-level hourglass at ~1 frame / 10 s
    x = torch.rand(n, 3, 128, 128, generator=g)
    with torch.no_grad():
        ref = O.region_predictor({k: v.float() for k, v in rsd.items()}, x, pad=pad)
        got = net(x.to(dev))
    assert_close(got["shift"].cpu(), ref["shift"], 1e-3, "region centres")
    assert_close(got["covar"].cpu(), ref["covar"], 1e-3, "region covariances")
    assert_close(got["affine"].cpu(), ref["affine"], 1e-3, "U sqrt(S) (LAPACK sign convention)")
    # and LAPACK on the device path's own covariances gives the same factors
    u, sv, _ = torch.svd(got["covar"].cpu().reshape(-1, 2, 2))
    host = (u @ torch.diag_embed(sv.sqrt())).view(*got["affine"].shape)
    assert_close(got["affine"].cpu(), host, 1e-4, "device closed form vs host LAPACK")
