"""End-to-end parity of the drop-in classes against the CPU oracle on a synthetic checkpoint:
Unet3D.forward, Generator.compute_fea / forward_with_flow, FlowDiffusion.sample_one_video
(DDIM and DDPM, noise replayed from a tape so CPU oracle and HIP path see identical draws).
Tolerance: north-star 1e-3 (fp32) for the sampled latent and the decoded frames."""
import os

import pytest
import torch

import lfdm_oracle as O
import synth
from util import assert_close


def _skip_slow_emu(dev):
    """A whole UNet forward under the fiber emulator takes minutes: opt in with LFDM_EMU_E2E=1."""
    if dev == "cpu" and os.environ.get("LFDM_EMU_E2E", "0") != "1":
        pytest.skip("end-to-end under the emulator is opt-in (LFDM_EMU_E2E=1); it runs on the GPU")


def _tiny(dev):
    return dict(b=2, t=4, s=8) if dev == "cpu" else dict(b=2, t=8, s=16)


@pytest.mark.parametrize("variant", ["deconv", "upconv_lnc"])
def test_unet_forward(backend, variant):
    dev = backend
    _skip_slow_emu(dev)
    if dev == "cpu" and variant == "upconv_lnc":
        pytest.skip("second UNet variant is exercised on the GPU (emulator time)")
    kw = {} if variant == "deconv" else dict(learn_null_cond=True, use_deconv=False, padding_mode="reflect")
    z = _tiny(dev)
    m, dsd, _ = synth.build_flow_diffusion(dev, img_size=z["s"], num_frames=z["t"], sampling_timesteps=5, **kw)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(z["b"], 259, z["t"], z["s"], z["s"], generator=g)
    x[:, 3:] = x[:, 3:, :1]                               # fea is constant over frames on the real path
    time = torch.tensor([999, 17][: z["b"]])
    cond = torch.randn(z["b"], 768, generator=g)
    for null_prob in (0.0, 1.0):
        mask = torch.full((z["b"],), bool(null_prob), dtype=torch.bool)
        ref = O.unet_forward(dsd, x, time, cond, mask)
        with torch.no_grad():
            out = m.unet.forward(x.to(dev), time.to(dev), cond=cond.to(dev), null_cond_prob=null_prob)
        assert_close(out.cpu(), ref, 2e-4, "unet forward (%s, null=%s)" % (variant, null_prob))


def test_generator(backend):
    dev = backend
    _skip_slow_emu(dev)
    hw = 32 if dev == "cpu" else 128
    m, _, gsd = synth.build_flow_diffusion(dev, img_size=hw // 4, num_frames=2, sampling_timesteps=5)
    img, _ = synth.inputs(2, hw)
    g = torch.Generator().manual_seed(5)
    s = hw // 4
    ident = torch.nn.functional.affine_grid(torch.eye(2, 3).unsqueeze(0), (1, 1, s, s), align_corners=True)
    flow = (ident.repeat(2, 1, 1, 1) + 0.3 * torch.randn(2, s, s, 2, generator=g)).clamp(-1.3, 1.3)
    occ = torch.rand(2, 1, s, s, generator=g)
    fea = m.generator.compute_fea(img.to(dev))
    assert_close(fea.cpu(), O.generator_compute_fea(gsd, img), 2e-4, "compute_fea")
    ref = O.generator_forward_with_flow(gsd, img, flow, occ)
    out = m.generator.forward_with_flow(img.to(dev), flow.to(dev), occ.to(dev))
    assert_close(out["deformed"].cpu(), ref["deformed"], 2e-4, "deformed")
    assert_close(out["prediction"].cpu(), ref["prediction"], 5e-4, "prediction")


@pytest.mark.parametrize("sampler", ["ddim", "ddpm"])
def test_sample_one_video(backend, sampler, tmp_path):
    dev = backend
    _skip_slow_emu(dev)
    z = dict(b=1, t=2, s=8, hw=32) if dev == "cpu" else dict(b=2, t=8, s=16, hw=64)
    steps, total = (3, 1000) if sampler == "ddim" else (6, 6)
    m, dsd, gsd = synth.build_flow_diffusion(dev, img_size=z["s"], num_frames=z["t"], sampling_timesteps=steps,
                                             timesteps=total)
    img, cond = synth.inputs(z["b"], z["hw"])
    sd = dict(dsd)
    sd.update(O.make_schedule(total))
    ref = O.sample_one_video(sd, gsd, img, cond, z["t"], z["s"], steps, timesteps=total,
                             noise_fn=synth.NoiseTape(11))
    m.diffusion.noise_source = synth.NoiseTape(11)
    m.set_sample_input(sample_img=img.to(dev), sample_text=cond.to(dev))
    m.sample_one_video(cond_scale=1.0)
    assert_close(m.sample_vid_grid.cpu(), ref["sample_vid_grid"], 1e-3, "sample_vid_grid")
    assert_close(m.sample_vid_conf.cpu(), ref["sample_vid_conf"], 1e-3, "sample_vid_conf")
    assert_close(m.sample_warped_vid.cpu(), ref["sample_warped_vid"], 1e-3, "sample_warped_vid")
    assert_close(m.sample_out_vid.cpu(), ref["sample_out_vid"], 1e-3, "sample_out_vid")
    if sampler == "ddim":      # the caller side of the demo scripts (tools/demo.py): five-panel frames -> GIF
        from PIL import Image
        from cvpr23_lfdm_amd import io_compat
        frames = io_compat.video_strip(m, img.to(dev), grid_size=z["s"])
        io_compat.mimsave(str(tmp_path / "demo.gif"), frames)
        with Image.open(str(tmp_path / "demo.gif")) as gif:
            assert gif.n_frames == z["t"] and gif.size == (5 * z["hw"], z["hw"])


@pytest.mark.parametrize("mode", ["cfg2", "cfg0", "residual_flow"])
def test_sample_variants(backend, mode):
    """Classifier-free guidance (cond_scale 2 = two UNet passes + combine, cond_scale 0 = null only,
    reference :511-526) and use_residual_flow (:195-198) against the oracle."""
    dev = backend
    _skip_slow_emu(dev)
    z = dict(b=1, t=2, s=8, hw=32) if dev == "cpu" else dict(b=2, t=4, s=8, hw=32)
    kw = dict(use_residual_flow=True) if mode == "residual_flow" else {}
    m, dsd, gsd = synth.build_flow_diffusion(dev, img_size=z["s"], num_frames=z["t"], sampling_timesteps=3, **kw)
    img, cond = synth.inputs(z["b"], z["hw"])
    scale = {"cfg2": 2.0, "cfg0": 0.0}.get(mode, 1.0)
    sd = dict(dsd)
    sd.update(O.make_schedule(1000))
    ref = O.sample_one_video(sd, gsd, img, cond, z["t"], z["s"], 3, cond_scale=scale, noise_fn=synth.NoiseTape(5),
                             use_residual_flow=(mode == "residual_flow"))
    m.diffusion.noise_source = synth.NoiseTape(5)
    m.set_sample_input(sample_img=img.to(dev), sample_text=cond.to(dev))
    m.sample_one_video(cond_scale=scale)
    for k in ("sample_vid_grid", "sample_vid_conf", "sample_warped_vid", "sample_out_vid"):
        assert_close(getattr(m, k).cpu(), ref[k], 1e-3, "%s (%s)" % (k, mode))


def test_second_call_reuses_graph(backend):
    """A second sample() with the same shapes replays the captured hipGraph with refreshed tables: the result
    must follow the new inputs (new image / cond / noise), not the first call's."""
    dev = backend
    _skip_slow_emu(dev)
    m, dsd, gsd = synth.build_flow_diffusion(dev, img_size=8, num_frames=4, sampling_timesteps=3)
    sd = dict(dsd)
    sd.update(O.make_schedule(1000))
    for seed in (7, 8):
        img, cond = synth.inputs(1, 32, seed=seed)
        ref = O.sample_one_video(sd, gsd, img, cond, 4, 8, 3, noise_fn=synth.NoiseTape(seed))
        m.diffusion.noise_source = synth.NoiseTape(seed)
        m.set_sample_input(sample_img=img.to(dev), sample_text=cond.to(dev))
        m.sample_one_video(cond_scale=1.0)
        assert_close(m.sample_out_vid.cpu(), ref["sample_out_vid"], 1e-3, "call with seed %d" % seed)


def test_graph_survives_arena_reallocation(backend):
    """the captured hipGraph holds raw pointers into the executor's scratch arenas.  An eager call that needs LARGER
    arenas in between (Unet3D.forward on a bigger batch) frees the old storage; the next sample() with the same shapes must
    re-capture on the new arenas (generation counter) instead of replaying pointers into freed memory - and a training-style
    weight update through FlatAdam's raw-pointer kernel must reach the sampler's packed weights."""
    dev = backend
    _skip_slow_emu(dev)
    m, dsd, gsd = synth.build_flow_diffusion(dev, img_size=8, num_frames=4, sampling_timesteps=3)
    sd = dict(dsd)
    sd.update(O.make_schedule(1000))
    img, cond = synth.inputs(1, 32, seed=7)
    ref = O.sample_one_video(sd, gsd, img, cond, 4, 8, 3, noise_fn=synth.NoiseTape(7))

    def sample():
        m.diffusion.noise_source = synth.NoiseTape(7)
        m.set_sample_input(sample_img=img.to(dev), sample_text=cond.to(dev))
        m.sample_one_video(cond_scale=1.0)
        return m.sample_out_vid.cpu()

    assert_close(sample(), ref["sample_out_vid"], 1e-3, "first call")
    gen = m.unet._buf_gen
    xb, tb, cb = synth.unet_inputs(3, 4, 8)                      # batch 3 > 1: every arena has to grow
    with torch.no_grad():
        m.unet(xb.to(dev), tb.to(dev), cond=cb.to(dev))
    assert m.unet._buf_gen > gen
    for _ in range(3):                                           # fill the freed blocks with something else
        torch.full((1 << 20,), float("nan"), device=dev)
    assert_close(sample(), ref["sample_out_vid"], 1e-3, "after the arenas were re-allocated")
    # ... and the sampler follows parameter writes torch cannot see
    from cvpr23_lfdm_amd.optim import FlatAdam
    opt = FlatAdam(m.unet.parameters(), lr=1e-2)
    for p in m.unet.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    moved = sample()
    assert float((moved - ref["sample_out_vid"]).abs().max()) > 1e-3


def test_c5_shape_natops_variant(backend):
    """BASELINE.json configs[4] geometry (64x64 latent = 256x256 frames, nearest-upsample + reflect-pad Upsample,
    learned null condition; 64-token mid spatial attention, 4x the spatial work per frame) at a reduced frame /
    step count, against the oracle."""
    dev = backend
    if dev == "cpu":
        pytest.skip("GPU-only size")
    t, s, hw, steps = 6, 64, 256, 2
    m, dsd, gsd = synth.build_flow_diffusion(dev, img_size=s, num_frames=t, sampling_timesteps=steps, learn_null_cond=True,
                                             use_deconv=False, padding_mode="reflect")
    img, cond = synth.inputs(1, hw, seed=21)
    sd = dict(dsd)
    sd.update(O.make_schedule(1000))
    ref = O.sample_one_video(sd, gsd, img, cond, t, s, steps, noise_fn=synth.NoiseTape(21))
    m.diffusion.noise_source = synth.NoiseTape(21)
    m.set_sample_input(sample_img=img.to(dev), sample_text=cond.to(dev))
    m.sample_one_video(cond_scale=1.0)
    for k in ("sample_vid_grid", "sample_vid_conf", "sample_warped_vid", "sample_out_vid"):
        assert_close(getattr(m, k).cpu(), ref[k], 1e-3, "%s (C5 shape)" % k)


@pytest.mark.gpu
def test_multi_step_graphs_draw_the_same_noise(monkeypatch):
    """Several sampler steps per graph launch with the noise draws captured inside (LFDM_GRAPH_STEPS): for the same seed the
    videos equal the one-step-per-replay loop's bit for bit (torch's graph-safe philox state makes the captured normal_() calls draw what
    the eager calls would), over two consecutive videos, incl. the chunk whose last step draws nothing and a ragged last chunk."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cvpr23_lfdm_amd import _native
    _native._set_library_for_tests(None)
    outs = {}
    for k in ("1", "3", "4"):
        monkeypatch.setenv("LFDM_GRAPH_STEPS", k)
        m, _, _ = synth.build_flow_diffusion("cuda", img_size=8, num_frames=4, sampling_timesteps=7)
        img, cond = synth.inputs(1, 32, seed=5)
        m.set_sample_input(sample_img=img.cuda(), sample_text=cond.cuda())
        torch.manual_seed(321)
        vids = []
        for _ in range(2):
            m.sample_one_video(cond_scale=1.0)
            vids.append(m.sample_out_vid.clone())
        outs[k] = vids
    for k in ("3", "4"):
        for a, b in zip(outs["1"], outs[k]):
            assert torch.equal(a, b), "LFDM_GRAPH_STEPS=%s draws different noise" % k
    assert not torch.equal(outs["1"][0], outs["1"][1])
