"""Pins the CPU oracle (oracle/lfdm_oracle.py) against fixtures produced by the UNMODIFIED reference
(oracle/make_golden.py ran /root/reference on CPU).  Runs anywhere (no GPU, no reference tree)."""
import os

import numpy as np
import pytest
import torch

import lfdm_oracle as O
import synth
from util import assert_close

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    path = os.path.join(GOLD, name + ".npz")
    if not os.path.exists(path):
        pytest.skip("fixture %s not generated" % name)
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(path).items()}


def test_ops_against_reference_fixtures():
    g = gold("ops")
    assert torch.equal(O.rel_pos_bias(g["emb"], 40), g["bias40"])
    cos, sin = O.rotary_tables(1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32)), 40)
    assert_close(O.apply_rotary(g["rot_q"], cos, sin), g["rot_out"], 1e-6, "rotary")
    assert torch.equal(O.abs_quantile(g["quant_in"], 0.9), g["quant_out"])
    sched = O.make_schedule(1000)
    for k in O.SCHEDULE_KEYS:
        assert torch.equal(sched[k], g[k]), k
    pairs = O.ddim_time_pairs(1000, 100)
    assert [p[0] for p in pairs] == list(reversed(g["ddim100_times"].tolist()))[:-1]
    assert pairs[0] == (990, 980) and pairs[-1] == (9, 0) and len(pairs) == 100


@pytest.mark.parametrize("name,variant", [("unet_tiny_deconv", {}),
                                          ("unet_tiny_upconv_lnc", dict(learn_null_cond=True, use_deconv=False))])
def test_unet_forward(name, variant):
    g = gold(name)
    b, t, s = int(g["b"]), int(g["t"]), int(g["s"])
    dsd = {"denoise_fn." + k: v for k, v in synth.unet_state(**variant).items()}
    x, time, cond = synth.unet_inputs(b, t, s)
    ones = torch.ones(b, dtype=torch.bool)
    assert_close(O.unet_forward(dsd, x, time, cond, ~ones), g["cond"], 1e-5, "cond")
    assert_close(O.unet_forward(dsd, x, time, cond, ones), g["null"], 1e-5, "null")
    assert_close(O.unet_forward_with_cond_scale(dsd, x, time, cond, 2.0), g["scale2"], 1e-5, "scale2")


def test_unet_forward_focus_present_mask():
    """Attention.forward's focus_present_mask branches (:313-317 all focused, :342-352 mixed) through Unet3D.forward, pinned by the reference."""
    g = gold("unet_tiny_focus")
    b, t, s = int(g["b"]), int(g["t"]), int(g["s"])
    dsd = {"denoise_fn." + k: v for k, v in synth.unet_state().items()}
    x, time, cond = synth.unet_inputs(b, t, s)
    zeros = torch.zeros(b, dtype=torch.bool)
    assert_close(O.unet_forward(dsd, x, time, cond, zeros, focus_mask=g["mask_mixed"]), g["focus_mixed"], 1e-5, "mixed focus mask")
    assert_close(O.unet_forward(dsd, x, time, cond, zeros, focus_mask=~zeros), g["focus_all"], 1e-5, "all focused")
    assert torch.equal(g["focus_all"], g["focus_p1"])          # prob_focus_present = 1 is the all-True mask


def test_generator():
    g = gold("generator_32")
    b, hw = int(g["b"]), int(g["hw"])
    gsd = synth.generator_state()
    img, _ = synth.inputs(b, hw)
    flow, occ = synth.flow_inputs(b, hw // 4)
    assert_close(O.generator_compute_fea(gsd, img), g["fea"], 1e-5, "fea")
    out = O.generator_forward_with_flow(gsd, img, flow, occ)
    assert_close(out["deformed"], g["deformed"], 1e-5, "deformed")
    assert_close(out["prediction"], g["prediction"], 1e-5, "prediction")


def test_generator_without_skips():
    """Generator(skips=False) (generator.py:153-161), minted from the reference class with the flag cleared."""
    g = gold("generator_32_noskips")
    b, hw = int(g["b"]), int(g["hw"])
    img, _ = synth.inputs(b, hw)
    flow, occ = synth.flow_inputs(b, hw // 4)
    out = O.generator_forward_with_flow(synth.generator_state(), img, flow, occ, use_skips=False)
    assert_close(out["deformed"], g["deformed"], 1e-5, "deformed")
    assert_close(out["prediction"], g["prediction"], 1e-5, "prediction")


@pytest.mark.parametrize("name", ["sample_ddim5_tiny", "sample_ddpm8_tiny", "sample_ddim5_tiny_static", "sample_ddim5_tiny_resflow"])
def test_sample_one_video(name):
    """(_static: use_dynamic_thres=False, _resflow: use_residual_flow=True - both minted from the unmodified reference.)"""
    g = gold(name)
    b, t, s, hw = int(g["b"]), int(g["t"]), int(g["s"]), int(g["hw"])
    steps, total = int(g["steps"]), int(g["timesteps"])
    sd = {"denoise_fn." + k: v for k, v in synth.unet_state().items()}
    sd.update(O.make_schedule(total))
    img, cond = synth.inputs(b, hw)
    out = O.sample_one_video(sd, synth.generator_state(), img, cond, t, s, steps, timesteps=total,
                             noise_fn=synth.NoiseTape(int(g["noise_seed"])), dynamic=not name.endswith("_static"),
                             use_residual_flow=name.endswith("_resflow"))
    vf = g["video_frames"].long() if "video_frames" in g else torch.arange(t)
    tol = 5e-5 if name.endswith("_static") else 2e-5      # (measured 2.7e-5 on the warped frames of the static-clamp case: two CPU
    for k in ("sample_vid_grid", "sample_vid_conf"):       #  formulations of the same fp32 network, five steps)
        assert_close(out[k], g[k], tol, k)
    for k in ("sample_out_vid", "sample_warped_vid"):
        assert_close(out[k][:, :, vf], g[k], tol, k)


def test_training_rows_of_the_oracle_match_the_reference_fixture():
    """oracle pseudo ground truth (region / bg / pixelwise-flow predictors, Generator.forward) and the diffusion loss
    against the reference's own training step (tests/golden/train_step_128.npz, first sample only: CPU time)."""
    import numpy as np
    g = np.load(os.path.join(GOLD, "train_step_128.npz"))
    b, t, hw = int(g["b"]), int(g["t"]), int(g["hw"])
    ref_img, real_vid, cond, tt, noise = synth.train_inputs(b, t, hw)
    gsd, rsd, bsd = synth.generator_state(), synth.region_state(), synth.bg_state()
    with torch.no_grad():
        pg = O.pseudo_ground_truth(gsd, rsd, bsd, ref_img[:1], real_vid[:1])
    T = lambda k: torch.from_numpy(g[k])
    assert_close(pg["real_vid_grid"], T("real_vid_grid")[:1], 2e-4, "pseudo-GT flow")
    assert_close(pg["real_vid_conf"], T("real_vid_conf")[:1], 2e-4, "pseudo-GT occlusion")
    assert_close(pg["real_out_vid"][:, :, -1], T("real_out_vid")[:1], 2e-4, "real_out_vid")
    assert_close(pg["real_warped_vid"][:, :, -1], T("real_warped_vid")[:1], 2e-4, "real_warped_vid")
    # the diffusion loss needs both samples (mean over the batch): use the fixture's own pseudo ground truth
    dsd = {"denoise_fn." + k: v for k, v in synth.unet_state().items()}
    dsd.update(O.make_schedule(1000))
    x0 = torch.cat((T("real_vid_grid"), T("real_vid_conf") * 2 - 1), dim=1)
    with torch.no_grad():
        pg_all = O.pseudo_ground_truth(gsd, rsd, bsd, ref_img, real_vid[:, :, :1])       # ref_img_fea of both samples
        loss, pred_x0 = O.p_losses(dsd, x0, tt, pg_all["ref_img_fea"], cond, noise,
                                   null_mask=torch.from_numpy(g["null_cond_mask"]))
    assert abs(float(loss) - float(g["loss"])) <= 2e-4 * max(1.0, abs(float(g["loss"])))
    assert_close(pred_x0, T("pred_x0"), 5e-4, "pred_x0")
