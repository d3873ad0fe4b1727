"""The frozen LFAE region predictor as the DM training step runs it (cvpr23_lfdm_amd.lfae_predictors.RegionPredictorExec:
batched frames, native convolutions, the device-side LAPACK-convention 2x2 SVD) against the CPU oracle's restatement of
LFAE/modules/region_predictor.py:52-117 (which calls torch.svd on the host like the reference)."""
import os

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


def test_bg_motion_predictor(backend):
    """BGMotionPredictorExec (bg_motion_predictor.py:42-57: encoder-only hourglass over cat(source, driving) -> global
    average -> fc -> 3x3 affine) directly against the oracle - not only through the training-step fixture."""
    dev = backend
    from cvpr23_lfdm_amd import FlowDiffusion
    m = FlowDiffusion(img_size=32, num_frames=2, sampling_timesteps=5, is_train=False, config_pth=synth.CONFIG, pretrained_pth="")
    bsd = synth.bg_state()
    m.bg_predictor.load_state_dict(bsd)
    net = m.bg_predictor.to(dev).eval()
    g = torch.Generator().manual_seed(78)
    n = 3 if dev == "cuda" else 1
    src, drv = torch.rand(n, 3, 128, 128, generator=g), torch.rand(n, 3, 128, 128, generator=g)
    with torch.no_grad():
        ref = O.bg_predictor({k: v.float() for k, v in bsd.items()}, src, drv)
        got = net(src.to(dev), drv.to(dev))
    assert got.shape == (n, 3, 3)
    assert_close(got.cpu(), ref, 1e-3, "background affine")
    assert float((ref[:, :2] - torch.eye(3)[:2]).abs().max()) > 1e-3          # not the trivial identity


def test_pixelwise_flow_predictor(backend):
    """PixelwiseFlowPredictorExec (pixelwise_flow_predictor.py:48-137: heat-maps, sparse motions through inv(affine), deformed
    sources, hourglass, softmax mask -> flow + occlusion) directly against the oracle, fed with oracle region parameters."""
    dev = backend
    from cvpr23_lfdm_amd import FlowDiffusion
    from cvpr23_lfdm_amd.lfae_predictors import PixelwiseFlowPredictorExec
    m = FlowDiffusion(img_size=32, num_frames=2, sampling_timesteps=5, is_train=False, config_pth=synth.CONFIG, pretrained_pth="")
    gsd, rsd, bsd = synth.generator_state(), synth.region_state(), synth.bg_state()
    m.generator.load_state_dict(gsd)
    m.generator.to(dev).eval()
    g = torch.Generator().manual_seed(79)
    n = 2
    src = torch.rand(n, 3, 128, 128, generator=g)
    drv_img = (0.8 * torch.roll(src, shifts=(5, -3), dims=(2, 3)) + 0.2 * torch.rand(n, 3, 128, 128, generator=g)).clamp(0, 1)
    f32 = lambda sd: {k: v.float() for k, v in sd.items()}
    with torch.no_grad():
        s_par, d_par = O.region_predictor(f32(rsd), src), O.region_predictor(f32(rsd), drv_img)
        bg = O.bg_predictor(f32(bsd), src, drv_img)
        ref = O.pixelwise_flow_predictor(f32(gsd), src, d_par, s_par, bg)
        ex = PixelwiseFlowPredictorExec(m.generator, num_regions=10)
        to = lambda d: {k: v.to(dev) for k, v in d.items() if k != "heatmap"}
        got = ex(src.to(dev), to(d_par), to(s_par), bg.to(dev))
    assert got["optical_flow"].shape == (n, 32, 32, 2) and got["occlusion_map"].shape == (n, 1, 32, 32)
    assert_close(got["optical_flow"].cpu(), ref["optical_flow"], 1e-3, "pixelwise flow")
    assert_close(got["occlusion_map"].cpu(), ref["occlusion_map"], 1e-3, "occlusion map")
    # the per-video form (frames=T: one source image / source regions per video, nothing repeated) = the per-frame form on
    # the repeated source
    with torch.no_grad():
        one = lambda d: {k: v[:1] for k, v in to(d).items()}
        rep2 = lambda d: {k: v[:1].repeat(2, *([1] * (v.dim() - 1))) for k, v in to(d).items()}
        a = ex(src[:1].to(dev), to(d_par), one(s_par), bg.to(dev), frames=2)
        b = ex(src[:1].repeat(2, 1, 1, 1).to(dev), to(d_par), rep2(s_par), bg.to(dev))
    assert torch.equal(a["optical_flow"], b["optical_flow"]) and torch.equal(a["occlusion_map"], b["occlusion_map"])


@pytest.mark.parametrize("bg_type", ["zero", "shift", "perspective"])
def test_bg_motion_predictor_other_types(backend, bg_type):
    """The bg_type values no LFDM yaml selects but the reference accepts (bg_motion_predictor.py:19, 27-57): identity, translation and
    perspective - the holder class and its executor against the oracle's restatement (itself checked against the live reference in
    tests/test_oracle_vs_reference.py), on a small encoder."""
    from cvpr23_lfdm_amd import params as P
    from cvpr23_lfdm_amd.flow_diffusion import BGMotionPredictor
    dev = backend
    kw = dict(block_expansion=8, max_features=32, num_blocks=2, bg_type=bg_type)
    net = BGMotionPredictor(num_channels=3, **kw)
    sd = P.synthetic_state_dict(P.bg_predictor_spec(num_channels=3, **kw), 6262)
    if bg_type != "zero":
        sd["fc.weight"] = sd["fc.weight"] * 0.05
        sd["fc.bias"] = torch.tensor(P.BG_FC_BIAS[bg_type], dtype=torch.float32) + 0.2 * sd["fc.bias"]
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    g = torch.Generator().manual_seed(79)
    n = 2
    src, drv = torch.rand(n, 3, 32, 32, generator=g), torch.rand(n, 3, 32, 32, generator=g)
    with torch.no_grad():
        ref = O.bg_predictor({k: v.float() for k, v in sd.items()}, src, drv, bg_type=bg_type, num_blocks=2)
        got = net(src.to(dev), drv.to(dev))
    assert got.shape == (n, 3, 3)
    assert_close(got.cpu(), ref, 1e-3, "background transform, bg_type %s" % bg_type)
    if bg_type != "zero":
        assert float((ref - torch.eye(3)).abs().max()) > 1e-3


@pytest.mark.parametrize("estimate_affine", [True, False], ids=["regression", "centres_only"])
def test_region_predictor_without_pca(backend, estimate_affine):
    """pca_based: false (region_predictor.py:43-49, 98-108): the FOMM-like `jacobian` regression head, or centres + heat-maps only -
    holder + executor against the oracle (which tests/test_oracle_vs_reference.py checks against the reference module), small hourglass."""
    from cvpr23_lfdm_amd import params as P
    from cvpr23_lfdm_amd.flow_diffusion import RegionPredictor
    dev = backend
    kw = dict(block_expansion=8, max_features=32, num_blocks=2, temperature=0.1, scale_factor=0.25, pca_based=False, pad=3)
    net = RegionPredictor(num_regions=4, num_channels=3, estimate_affine=estimate_affine, **kw)
    assert net.has("jacobian.weight") == estimate_affine
    sd = P.synthetic_state_dict(P.region_predictor_spec(num_regions=4, num_channels=3, estimate_affine=estimate_affine, **kw), 7171)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(80))
    with torch.no_grad():
        ref = O.region_predictor({k: v.float() for k, v in sd.items()}, x, pca_based=False, num_blocks=2)
        got = net(x.to(dev))
    assert set(got) == set(ref)
    for key in ref:
        assert_close(got[key].cpu(), ref[key], 1e-3, "region predictor (no pca) " + key)


def _small_generator(use_deformed_source, k=4):
    from cvpr23_lfdm_amd import params as P
    from cvpr23_lfdm_amd.generator import Generator
    fp = dict(block_expansion=8, max_features=32, num_blocks=2, scale_factor=0.25, use_covar_heatmap=True,
              use_deformed_source=use_deformed_source, estimate_occlusion_map=True)
    gen = Generator(num_channels=3, num_regions=k, block_expansion=8, max_features=32, num_down_blocks=1, num_bottleneck_blocks=1,
                    pixelwise_flow_predictor_params=fp)
    spec = P.generator_spec(num_channels=3, block_expansion=8, max_features=32, num_down_blocks=1, num_bottleneck_blocks=1, num_regions=k,
                            fp_block_expansion=8, fp_max_features=32, fp_num_blocks=2, use_deformed_source=use_deformed_source)
    gsd = P.synthetic_state_dict(spec, 4242)
    gen.load_state_dict(gsd)
    return gen, gsd, fp


def _region_params(n, k, g):
    mk = lambda: {"shift": torch.rand(n, k, 2, generator=g) * 1.2 - 0.6,
                  "covar": torch.eye(2).view(1, 1, 2, 2) * 0.02 + 0.004 * torch.rand(n, k, 1, 1, generator=g),
                  "affine": torch.eye(2).view(1, 1, 2, 2) * 0.15 + 0.03 * torch.rand(n, k, 2, 2, generator=g)}
    return mk(), mk()


@pytest.mark.parametrize("use_deformed_source", [False, True])
def test_pixelwise_flow_predictor_heatmaps_only(backend, use_deformed_source):
    """use_deformed_source=False (pixelwise_flow_predictor.py:28, :116-119; no LFDM yaml): the hourglass reads the K+1 heat-maps only - the
    frozen executor and the trainer's forward (eval statistics) against the oracle (pinned on the live reference in test_oracle_vs_reference)."""
    dev = backend
    from cvpr23_lfdm_amd import lfae_train
    from cvpr23_lfdm_amd.lfae_predictors import PixelwiseFlowPredictorExec
    k, n = 4, 2
    gen, gsd, fp = _small_generator(use_deformed_source, k)
    gen = gen.to(dev).eval()
    g = torch.Generator().manual_seed(81)
    src = torch.rand(n, 3, 64, 64, generator=g)
    drv, sr = _region_params(n, k, g)
    bg = torch.eye(3).view(1, 3, 3).repeat(n, 1, 1) + 0.05 * torch.rand(n, 3, 3, generator=g)
    to = lambda d: {key: v.to(dev) for key, v in d.items()}
    with torch.no_grad():
        ref = O.pixelwise_flow_predictor({key: v.float() for key, v in gsd.items()}, src, drv, sr, bg, num_regions=k,
                                         use_deformed_source=use_deformed_source, num_blocks=2)
        ex = PixelwiseFlowPredictorExec(gen, num_regions=k, num_blocks=2, scale_factor=0.25, use_covar_heatmap=True,
                                        use_deformed_source=use_deformed_source)
        got = ex(src.to(dev), to(drv), to(sr), bg.to(dev))
        trn = lfae_train.pixelwise_flow_forward(gen, src.to(dev), to(drv), to(sr), bg.to(dev), fp, k, True, training=False)
    for name, out in (("executor", got), ("trainer forward", trn)):
        assert_close(out["optical_flow"].cpu(), ref["optical_flow"], 1e-3, name + ": pixelwise flow")
        assert_close(out["occlusion_map"].cpu(), ref["occlusion_map"], 1e-3, name + ": occlusion map")
    # calls above the launches' frame limit run in slices of whole videos: same result (limit lowered to one frame per slice here)
    ex.MAX_FRAMES = 1
    with torch.no_grad():
        sliced = ex(src.to(dev), to(drv), to(sr), bg.to(dev))
    assert torch.equal(sliced["optical_flow"], got["optical_flow"]) and torch.equal(sliced["occlusion_map"], got["occlusion_map"])


def test_region_predictor_large_heatmaps(backend):
    """Heat-maps above 4096 pixels (frames above 256x256 at scale 0.25): the fused statistics launch does not take them; the executor computes
    the statistics in tensor ops and the 2x2 SVD on the native kernel - against the oracle."""
    dev = backend
    from cvpr23_lfdm_amd import params as P
    from cvpr23_lfdm_amd.flow_diffusion import RegionPredictor
    kw = dict(block_expansion=8, max_features=32, num_blocks=2, temperature=0.1, scale_factor=0.25, pca_based=True, pad=3)
    net = RegionPredictor(num_regions=4, num_channels=3, estimate_affine=True, **kw)
    sd = P.synthetic_state_dict(P.region_predictor_spec(num_regions=4, num_channels=3, estimate_affine=True, **kw), 7171)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    x = torch.rand(1, 3, 272, 288, generator=torch.Generator().manual_seed(80))            # 68 x 72 = 4896-pixel maps
    with torch.no_grad():
        ref = O.region_predictor({key: v.float() for key, v in sd.items()}, x, num_blocks=2)
        got = net(x.to(dev))
    assert got["heatmap"].shape[-2:] == (68, 72)
    for key in ("shift", "covar", "affine", "heatmap"):
        assert_close(got[key].cpu(), ref[key], 1e-3, "large-map region predictor: " + key)


def test_constructor_defaults_are_the_reference_ones():
    """A yaml without estimate_affine / pca_based / bg_type builds the reference's module tree (region_predictor.py:33-35: estimate_affine
    False, pca_based False; bg_motion_predictor.py:20: bg_type 'zero' = no parameters), the same in every place that reads the keys - the
    state dict then loads into a reference model built from the same yaml."""
    from cvpr23_lfdm_amd import params as P
    from cvpr23_lfdm_amd.flow_diffusion import BGMotionPredictor, RegionPredictor
    kw = dict(block_expansion=8, max_features=32, num_blocks=2, temperature=0.1, scale_factor=0.25)
    net = RegionPredictor(num_regions=4, num_channels=3, **kw)
    assert not net.has("jacobian.weight") and not net._exec.regression and not net._exec.pca_based
    assert [k for k, *_ in P.region_predictor_spec(num_regions=4, num_channels=3, **kw)] == list(net.state_dict().keys())
    bg = BGMotionPredictor(num_channels=3, block_expansion=8, max_features=32, num_blocks=2)
    assert bg.bg_type == "zero" and bg._exec.bg_type == "zero" and len(bg.state_dict()) == 0
    assert P.bg_predictor_spec(num_channels=3) == []
