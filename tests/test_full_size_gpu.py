"""FULL-SIZE parity for the other BASELINE.json configurations (configs[2..4]), against compact fixtures minted from the
unmodified reference at those sizes (oracle/make_golden.py --full c3|c4|c5: strided sub-tensors + per-sample statistics +
random projections of every result, so batch-16 / 40-frame / 256x256 tensors are pinned without storing them).

  c3  MHAD shape class, DDPM-1000, batch 16, 40 frames: teacher-forced sampler steps at t = 999, 500, 0 through the REAL
      sampling path (fea term, step tables, stem, trunk, radix-select threshold, update; one captured hipGraph per call)
  c4  one DM training step at B = 4 and at the per-GPU B = 8, T = 40, 128x128 (pseudo ground truth of 160 / 320 frames, loss, pred_x0,
      every gradient norm)
  c5  NATOPS variant (learned null condition, upsample + reflect), 64x64 latent, 256x256 frames, 40 frames, DDIM-10
"""
import os

import numpy as np
import pytest
import torch

import synth
from util import assert_close, record_margin

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
pytestmark = pytest.mark.gpu


def gold(name):
    path = os.path.join(GOLD, name + ".npz")
    if not os.path.exists(path):
        pytest.skip("fixture %s not generated" % name)
    return {k: np.asarray(v) for k, v in np.load(path).items()}


def probes(x, n=64, seed=5):
    """oracle/make_golden.py::probes on the device tensor."""
    x = x.detach().double().reshape(x.shape[0], -1).cpu()
    d = torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal((n, x.shape[1])))
    return torch.stack((x.mean(1), x.abs().mean(1), x.std(1)), dim=1), (x @ d.t()) / np.sqrt(x.shape[1])


@pytest.fixture(autouse=True)
def _hip():
    from cvpr23_lfdm_amd import _native
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _native._set_library_for_tests(None)


def test_c3_ddpm_steps_batch16():
    g = gold("c3_ddpm_steps_b16")
    b, t, s = int(g["b"]), int(g["t"]), int(g["s"])
    m, _, _ = synth.build_flow_diffusion("cuda", img_size=s, num_frames=t, sampling_timesteps=1000)
    dif = m.diffusion
    rng = np.random.Generator(np.random.PCG64(int(g["input_seed"])))
    fea = torch.from_numpy(rng.standard_normal((b, 256, s, s)).astype(np.float32)).cuda()
    cond = torch.from_numpy(rng.standard_normal((b, 768)).astype(np.float32)).cuda()
    times, coef, _ = dif._step_tables(False)
    for i, step in enumerate(g["steps"].tolist()):
        x_t = torch.from_numpy(rng.standard_normal((b, 3, t, s, s)).astype(np.float32))
        noise = torch.from_numpy(rng.standard_normal((b, 3, t, s, s)).astype(np.float32))
        k = times.index(step)
        dif._step_tables = lambda ddim, k=k: ([times[k]], coef[k:k + 1].contiguous(), [True])        # ONE teacher-forced step
        tape = iter([x_t, noise])
        dif.noise_source = lambda shape: next(tape)                     # first draw = "x_T" (our x_t), second = the step's noise
        y = dif.sample(fea, cond=cond, cond_scale=1.0)
        assert y.shape == (b, 3, t, s, s) and bool(torch.isfinite(y).all())
        assert_close(y[:, :, ::8, ::4, ::4].cpu(), torch.from_numpy(g["x_prev_%d" % i]), 1e-3, "x_{t-1} sub-tensor, t=%d" % step)
        st, pr = probes(y)
        assert_close(st.float(), torch.from_numpy(g["stats_%d" % i]).float(), 1e-3, "per-sample statistics, t=%d" % step)
        assert_close(pr.float(), torch.from_numpy(g["probes_%d" % i]).float(), 2e-3, "random projections, t=%d" % step)


@pytest.mark.parametrize("fixture", ["train_step_c4_b4_t40", "train_step_c4_b8_t40"])
def test_c4_training_step_t40(monkeypatch, fixture):
    """(_b8_: the per-GPU batch of BASELINE.json configs[3], 64 videos over 8 GPUs - 320 pseudo-ground-truth frames.)"""
    g = gold(fixture)
    b, t, hw = int(g["b"]), int(g["t"]), int(g["hw"])
    from cvpr23_lfdm_amd import FlowDiffusion
    m = FlowDiffusion(img_size=hw // 4, num_frames=t, sampling_timesteps=5, timesteps=1000, null_cond_prob=0.0, is_train=True, lr=1e-3,
                      config_pth=synth.CONFIG, pretrained_pth="")
    m.unet.load_state_dict(synth.unet_state())
    m.generator.load_state_dict(synth.generator_state())
    m.region_predictor.load_state_dict(synth.region_state())
    m.bg_predictor.load_state_dict(synth.bg_state())
    for net in (m.generator, m.region_predictor, m.bg_predictor):
        net.eval()
        m.set_requires_grad(net, False)
    m.to("cuda")
    ref_img, real_vid, cond, tt, noise = synth.train_inputs(b, t, hw)
    m.diffusion.text_encoder = lambda texts: cond
    monkeypatch.setattr(torch, "randint", lambda *a, **k: tt.clone().to(k.get("device", "cpu")))
    monkeypatch.setattr(torch, "randn_like", lambda x, **k: noise.clone().to(x.device))
    m.set_train_input(ref_img=ref_img.cuda(), real_vid=real_vid.cuda(), ref_text=[str(x) for x in g["labels"]])
    m.optimize_parameters()
    monkeypatch.undo()
    T = lambda k: torch.from_numpy(g[k])
    sub = lambda v: v[:, :, ::8, ::2, ::2].cpu()
    assert_close(sub(m.real_vid_grid), T("real_vid_grid"), 1e-3, "pseudo-GT flow (160 frames)")
    assert_close(sub(m.real_vid_conf), T("real_vid_conf"), 1e-3, "pseudo-GT occlusion")
    st, pr = probes(m.real_vid_grid)
    assert_close(st.float(), T("grid_stats").float(), 1e-3, "flow statistics")
    assert_close(pr.float(), T("grid_probes").float(), 1e-3, "flow projections")
    assert bool((m.unet.null_cond_mask.cpu() == T("null_cond_mask")).all())
    assert_close(sub(m.diffusion.pred_x0), T("pred_x0"), 1e-3, "pred_x0")
    st, pr = probes(m.diffusion.pred_x0)
    assert_close(st.float(), T("pred_x0_stats").float(), 1e-3, "pred_x0 statistics")
    assert_close(pr.float(), T("pred_x0_probes").float(), 1e-3, "pred_x0 projections")
    assert_close(m.real_out_vid[:, :, -1, ::2, ::2].cpu(), T("real_out_vid"), 1e-3, "real_out_vid")
    assert_close(m.fake_out_vid[:, :, -1, ::2, ::2].cpu(), T("fake_out_vid"), 1e-3, "fake_out_vid")
    for k in ("loss", "rec_loss", "rec_warp_loss"):
        got, want = float(getattr(m, k)), float(g[k])
        record_margin(k, abs(got - want), abs(want), 1e-3)
        assert abs(got - want) <= 1e-3 * max(1.0, abs(want)), (k, got, want)
    names = [str(n) for n in g["names"]]
    params = dict(m.diffusion.named_parameters())
    rng = np.random.Generator(np.random.PCG64(77))
    worst = ("", 0.0)
    for i, k in enumerate(names):
        p = params[k]
        gr = p.grad.detach().double().cpu()
        probe = torch.from_numpy(rng.standard_normal(p.numel())).view_as(gr)
        want = float(g["grad_norm"][i])
        for tag, e in (("grad norm", abs(float(gr.norm()) - want) / (want + 1e-12)),
                       ("grad probe", abs(float((gr * probe).sum()) - float(g["grad_probe"][i])) / (want * np.sqrt(p.numel()) + 1e-12)),
                       ("updated weight norm", abs(float(p.detach().double().norm()) - float(g["param_norm_after"][i])) / (float(g["param_norm_after"][i]) + 1e-12))):
            if e > worst[1]:
                worst = ("%s of %s" % (tag, k), e)
    record_margin("largest relative deviation over all parameter gradients / updated weights: " + worst[0], worst[1], 1.0, 1e-3)
    assert worst[1] < 1e-3, "largest deviation %.3e: %s" % (worst[1], worst[0])


@pytest.mark.parametrize("fixture", ["sample_ddim10_c5_256", "sample_ddim50_c5_256", "sample_ddim50_c5_256_b4"])      # 10 steps, the configuration's own 50, and
def test_c5_natops_256_t40(fixture):                                                                                   # its per-GPU batch (32 videos / 8 GPUs = B 4)
    g = gold(fixture)
    q = int(g["stride"]) if "stride" in g else 2
    b, t, s, hw = int(g["b"]), int(g["t"]), int(g["s"]), int(g["hw"])
    m, _, _ = synth.build_flow_diffusion("cuda", img_size=s, num_frames=t, sampling_timesteps=int(g["steps"]), timesteps=int(g["timesteps"]),
                                         learn_null_cond=True, use_deconv=False, padding_mode="reflect")
    img, cond = synth.inputs(b, hw)
    m.diffusion.noise_source = synth.NoiseTape(int(g["noise_seed"]))
    m.set_sample_input(sample_img=img.cuda(), sample_text=cond.cuda())
    m.sample_one_video(cond_scale=1.0)
    vf = torch.from_numpy(g["video_frames"]).long()
    T = lambda k: torch.from_numpy(g[k])
    assert m.sample_out_vid.shape == (b, 3, t, hw, hw)
    assert_close(m.sample_vid_grid[:, :, :, ::q, ::q].cpu(), T("sample_vid_grid"), 1e-3, "flow")
    assert_close(m.sample_vid_conf[:, :, :, ::q, ::q].cpu(), T("sample_vid_conf"), 1e-3, "occlusion")
    assert_close(m.sample_out_vid.cpu()[:, :, vf][..., ::q, ::q], T("sample_out_vid"), 1e-3, "frames")
    assert_close(m.sample_warped_vid.cpu()[:, :, vf][..., ::q, ::q], T("sample_warped_vid"), 1e-3, "warped frames")
    st, pr = probes(m.sample_out_vid)
    assert_close(st.float(), T("out_stats").float(), 1e-3, "video statistics")
    assert_close(pr.float(), T("out_probes").float(), 2e-3, "video projections")


def test_c3_ddpm1000_full_schedule():
    """BASELINE.json configs[2] END TO END: `p_sample_loop` (DM/modules/video_flow_diffusion.py:712-759) over ALL 1000 timesteps - the one
    BASELINE sampler whose whole schedule had only been pinned by teacher-forced steps (test_c3_ddpm_steps_batch16) - at B = 2, T = 40,
    32x32 latent, on the recorded noise tape the reference fixture was minted with (oracle/make_golden.py --full c3full), then the
    40-frame LFAE decode.  Besides the final tensors the fixture holds the state x_t ENTERING t = 750 / 500 / 250 and statistics / random
    projections of x_t every 50 steps: the error-vs-step curve goes to the parity log (profiles/*_parity_margins.json, and
    profiles/r06_*_c3_drift.txt), so a miss would show where the drift enters.  Bar: 1e-3 (north star) on everything."""
    g = gold("sample_ddpm1000_c3_b2")
    b, t, s, hw, steps = int(g["b"]), int(g["t"]), int(g["s"]), int(g["hw"]), int(g["timesteps"])
    m, _, _ = synth.build_flow_diffusion("cuda", img_size=s, num_frames=t, sampling_timesteps=steps, timesteps=steps)
    assert not m.diffusion.is_ddim_sampling and steps == 1000
    img, cond = synth.inputs(b, hw)
    tape = synth.NoiseTape(int(g["noise_seed"]))
    every, keep = int(g["every"]), [int(k) for k in g["keep"]]
    seen, calls = {}, [0]

    def source(shape):
        # call 0 draws x_T; call k >= 1 draws the noise of the step at timestep steps - k, and the sampler's state buffer still holds the
        # x_t ENTERING that step (one step per graph replay on a noise tape): what the reference's p_sample received
        k = calls[0]
        calls[0] += 1
        step = steps - k
        if k >= 1 and (step % every == 0 or step in keep):
            x = next(iter(m.diffusion._plans.values()))["x"]
            st, pr = probes(x)
            seen[step] = (st, pr, x[:, :, ::8, ::4, ::4].cpu().clone() if step in keep else None)
        return tape(shape)

    m.diffusion.noise_source = source
    m.set_sample_input(sample_img=img.cuda(), sample_text=cond.cuda())
    m.sample_one_video(cond_scale=1.0)
    assert calls[0] == steps + 1
    T = lambda k: torch.from_numpy(g[k])
    # ---- the drift curve first (recorded even if a later assertion fails): probes / statistics of x_t every `every` steps
    curve = []
    for step in sorted(seen, reverse=True):
        st, pr, sub = seen[step]
        e_st = float((st.float() - T("xt_stats_%d" % step).float()).abs().max())
        e_pr = float((pr.float() - T("xt_probes_%d" % step).float()).abs().max())
        scale = max(1.0, float(T("xt_probes_%d" % step).abs().max()))
        record_margin("x_t projections entering t=%d" % step, e_pr, scale, 1e-3)
        curve.append((step, e_st, e_pr, scale))
    out = os.environ.get("LFDM_C3_DRIFT_OUT")
    if out:
        with open(out, "w") as f:
            f.write("# DDPM-1000 (configs[2], B = 2, T = 40, 32x32): max |HIP - reference| of the per-sample statistics (mean, mean |x|, std) and of\n"
                    "# 64 random unit-variance projections per sample of the sampler state x_t ENTERING timestep t; bar = 1e-3 * max(1, scale)\n"
                    "# %6s %14s %14s %10s\n" % ("t", "err(stats)", "err(probes)", "scale"))
            for step, e_st, e_pr, scale in curve:
                f.write("  %6d %14.3e %14.3e %10.3f\n" % (step, e_st, e_pr, scale))
    for step, e_st, e_pr, scale in curve:
        assert e_st <= 1e-3 and e_pr <= 1e-3 * scale, "drift at t=%d: stats %.3e, projections %.3e (scale %.2f)" % (step, e_st, e_pr, scale)
    for step in keep:
        assert_close(seen[step][2], T("xt_%d" % step), 1e-3, "x_t entering t=%d (sub-tensor)" % step)
    # ---- the end of the schedule and the decode
    vf = torch.from_numpy(g["video_frames"]).long()
    pred = torch.cat((m.sample_vid_grid, m.sample_vid_conf * 2 - 1), dim=1)
    st, pr = probes(pred)
    assert_close(st.float(), T("pred_stats").float(), 1e-3, "x_0 statistics after 1000 steps")
    assert_close(pr.float(), T("pred_probes").float(), 1e-3, "x_0 projections after 1000 steps")
    assert_close(m.sample_vid_grid[:, :, :, ::2, ::2].cpu(), T("sample_vid_grid"), 1e-3, "flow after 1000 steps")
    assert_close(m.sample_vid_conf[:, :, :, ::2, ::2].cpu(), T("sample_vid_conf"), 1e-3, "occlusion after 1000 steps")
    assert_close(m.sample_out_vid.cpu()[:, :, vf][..., ::2, ::2], T("sample_out_vid"), 1e-3, "frames")
    assert_close(m.sample_warped_vid.cpu()[:, :, vf][..., ::2, ::2], T("sample_warped_vid"), 1e-3, "warped frames")
    st, pr = probes(m.sample_out_vid)
    assert_close(st.float(), T("out_stats").float(), 1e-3, "video statistics")
    assert_close(pr.float(), T("out_probes").float(), 1e-3, "video projections")
