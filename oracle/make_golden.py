#!/usr/bin/env python
"""TEST INFRASTRUCTURE - generates tests/golden/*.npz by running the UNMODIFIED reference
(/root/reference, imported on CPU through oracle/ref_shims) on synthetic checkpoints and inputs
that every consumer can regenerate bit-for-bit from seeds (tests/synth.py: numpy PCG64).

The reference has no tests / golden vectors of its own, so these fixtures are
what pins the oracle (tests/test_oracle_golden.py) and the HIP path (tests/test_golden_gpu.py).
Only OUTPUTS are stored; inputs and weights are re-derived from the seeds recorded in each file.

    python oracle/make_golden.py            # small cases (seconds)
    python oracle/make_golden.py --c2       # + one C2-shape UNet forward and a full C2 DDIM-100 video (minutes)
    python oracle/make_golden.py --full c3|c4|c5    # full-size cases of BASELINE.json configs[2..4] (compact fixtures)
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import reference_loader  # noqa: E402

ref = reference_loader.load_reference()
sys.path.append(REPO)
sys.path.append(os.path.join(REPO, "tests"))
import synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


class patched_noise:
    """Routes torch.randn / randn_like of the reference's samplers to a NoiseTape."""

    def __init__(self, tape):
        self.tape = tape

    def __enter__(self):
        self.randn, self.randn_like = torch.randn, torch.randn_like
        torch.randn = lambda *shape, **kw: self.tape(tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape))
        torch.randn_like = lambda t, **kw: self.tape(tuple(t.shape))

    def __exit__(self, *a):
        torch.randn, torch.randn_like = self.randn, self.randn_like


def reference_model(img_size, num_frames, sampling_timesteps, timesteps=1000, **variant):
    m = ref.vfdm.FlowDiffusion(img_size=img_size, num_frames=num_frames, sampling_timesteps=sampling_timesteps,
                               timesteps=timesteps, is_train=False, config_pth=synth.CONFIG, pretrained_pth="",
                               **variant)
    spec_kw = dict(learn_null_cond=variant.get("learn_null_cond", False), use_deconv=variant.get("use_deconv", True))
    m.unet.load_state_dict(synth.unet_state(**spec_kw))
    m.generator.load_state_dict(synth.generator_state())
    m.eval()
    return m


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def unet_case(name, b, t, s, **variant):
    m = reference_model(s, t, 5, **variant)
    x, time, cond = synth.unet_inputs(b, t, s)
    outs = {}
    with torch.no_grad():
        for tag, prob in (("cond", 0.0), ("null", 1.0)):
            outs[tag] = m.unet(x, time, cond=cond, null_cond_prob=prob)
        outs["scale2"] = m.unet.forward_with_cond_scale(x, time, cond=cond, cond_scale=2.0)
    save(name, b=b, t=t, s=s, **outs)


def unet_focus_case(name, b=3, t=4, s=8):
    """Unet3D.forward with focus_present_mask (:542-543, Attention.forward :313-317 / :342-352): a mixed mask and the all-True shortcut."""
    m = reference_model(s, t, 5)
    x, time, cond = synth.unet_inputs(b, t, s)
    mixed = torch.tensor([True, False, True][:b])
    with torch.no_grad():
        out_mixed = m.unet(x, time, cond=cond, null_cond_prob=0.0, focus_present_mask=mixed)
        out_all = m.unet(x, time, cond=cond, null_cond_prob=0.0, focus_present_mask=torch.ones(b, dtype=torch.bool))
        out_p1 = m.unet(x, time, cond=cond, null_cond_prob=0.0, prob_focus_present=1.0)
    save(name, b=b, t=t, s=s, mask_mixed=mixed, focus_mixed=out_mixed, focus_all=out_all, focus_p1=out_p1)


def sampler_case(name, b, t, s, hw, steps, timesteps, video_frames=None, static_clip=False, **variant):
    """video_frames: keep only these frame indices of the image-resolution outputs (fixture size).  static_clip: the
    GaussianDiffusion default use_dynamic_thres=False (x0.clamp(-1, 1), :729-732) instead of the wrapper's dynamic threshold."""
    m = reference_model(s, t, steps, timesteps, **variant)
    if static_clip:
        m.diffusion.use_dynamic_thres = False
    img, cond = synth.inputs(b, hw)
    m.set_sample_input(sample_img=img, sample_text=cond)
    with patched_noise(synth.NoiseTape(11)), torch.no_grad():
        m.sample_one_video(cond_scale=1.0)
    vf = np.arange(t) if video_frames is None else np.asarray(video_frames)
    save(name, b=b, t=t, s=s, hw=hw, steps=steps, timesteps=timesteps, noise_seed=11, video_frames=vf,
         sample_vid_grid=m.sample_vid_grid, sample_vid_conf=m.sample_vid_conf,
         sample_out_vid=m.sample_out_vid[:, :, vf], sample_warped_vid=m.sample_warped_vid[:, :, vf])


def generator_case(name, b, hw):
    m = reference_model(hw // 4, 2, 5)
    img, _ = synth.inputs(b, hw)
    flow, occ = synth.flow_inputs(b, hw // 4)
    with torch.no_grad():
        fea = m.generator.compute_fea(img)
        out = m.generator.forward_with_flow(img, flow, occ)
    save(name, b=b, hw=hw, fea=fea, prediction=out["prediction"], deformed=out["deformed"])


def generator_noskips_case(name, b=2, hw=32):
    """Generator(skips=False) (generator.py:24,57,153-161): the reference class itself with the constructor flag cleared."""
    m = reference_model(hw // 4, 2, 5)
    m.generator.skips = False
    img, _ = synth.inputs(b, hw)
    flow, occ = synth.flow_inputs(b, hw // 4)
    with torch.no_grad():
        out = m.generator.forward_with_flow(img, flow, occ)
    save(name, b=b, hw=hw, prediction=out["prediction"], deformed=out["deformed"])


def train_case(name, b, t, hw, labels, compact=False, null_cond_prob=0.0, **variant):
    """One full DM training step of the reference (FlowDiffusion.optimize_parameters, single-GPU class) on synthetic
    frozen-LFAE + UNet checkpoints: pseudo ground truth, losses, per-parameter gradient / updated-weight statistics.
    Randomness (t, noise) is replaced by recorded tensors; null_cond_prob = 0 (labels 'None' exercise none_cond_mask);
    text -> embedding is a fixed tensor (the reference's BERT download is unavailable)."""
    m = ref.vfdm.FlowDiffusion(img_size=hw // 4, num_frames=t, sampling_timesteps=5, timesteps=1000, null_cond_prob=null_cond_prob,
                               is_train=True, lr=1e-3, config_pth=synth.CONFIG, pretrained_pth="", **variant)
    m.unet.load_state_dict(synth.unet_state())
    m.generator.load_state_dict(synth.generator_state())
    m.region_predictor.load_state_dict(synth.region_state())
    m.bg_predictor.load_state_dict(synth.bg_state())
    for net in (m.generator, m.region_predictor, m.bg_predictor):     # what pretrained_pth != "" does (:42-61)
        net.eval()
        m.set_requires_grad(net, False)
    ref_img, real_vid, cond, tt, noise = synth.train_inputs(b, t, hw)
    ref.vfd.tokenize = lambda texts: texts
    ref.vfd.bert_embed = lambda tokens, return_cls_repr=False: cond
    randint, randn_like, uniform_ = torch.randint, torch.randn_like, torch.Tensor.uniform_
    torch.randint = lambda *a, **k: tt.clone()
    torch.randn_like = lambda x, **k: noise.clone()
    if 0 < null_cond_prob < 1:      # prob_mask_like (:55-61) draws zeros(B).uniform_(0, 1) < prob: replay a recorded draw
        u = synth.null_uniform(b)
        torch.Tensor.uniform_ = lambda self, *a, **k: self.copy_(u.to(self.device)) if tuple(self.shape) == (b,) else uniform_(self, *a, **k)
    try:
        m.set_train_input(ref_img=ref_img, real_vid=real_vid, ref_text=labels)
        m.optimize_parameters()
    finally:
        torch.randint, torch.randn_like, torch.Tensor.uniform_ = randint, randn_like, uniform_
    rng = np.random.Generator(np.random.PCG64(77))
    names, gnorm, pnorm, gprobe = [], [], [], []
    small = {}
    for k, p in m.diffusion.named_parameters():
        names.append(k)
        g = p.grad.detach().double()
        probe = torch.from_numpy(rng.standard_normal(p.numel())).view_as(g)
        gnorm.append(float(g.norm()))
        gprobe.append(float((g * probe).sum()))
        pnorm.append(float(p.detach().double().norm()))
        if p.numel() <= 128:
            small["grad/" + k] = p.grad.detach()
    if compact:          # full-T fixture: strided sub-tensors + statistics / projections instead of whole videos
        sub = lambda v: v[:, :, ::8, ::2, ::2].clone()
        st_g, pr_g = probes(m.real_vid_grid)
        st_x, pr_x = probes(m.diffusion.pred_x0)
        save(name, b=b, t=t, hw=hw, labels=np.array(labels), names=np.array(names), grad_norm=np.array(gnorm),
             grad_probe=np.array(gprobe), param_norm_after=np.array(pnorm), real_vid_grid=sub(m.real_vid_grid),
             real_vid_conf=sub(m.real_vid_conf), grid_stats=st_g, grid_probes=pr_g, pred_x0=sub(m.diffusion.pred_x0),
             pred_x0_stats=st_x, pred_x0_probes=pr_x, real_out_vid=m.real_out_vid[:, :, -1, ::2, ::2], fake_out_vid=m.fake_out_vid[:, :, -1, ::2, ::2],
             loss=m.loss.detach(), rec_loss=m.rec_loss, rec_warp_loss=m.rec_warp_loss,
             null_cond_mask=m.diffusion.denoise_fn.null_cond_mask, **small)
        return
    save(name, b=b, t=t, hw=hw, labels=np.array(labels), names=np.array(names), grad_norm=np.array(gnorm),
         grad_probe=np.array(gprobe), param_norm_after=np.array(pnorm),
         real_vid_grid=m.real_vid_grid, real_vid_conf=m.real_vid_conf, real_out_vid=m.real_out_vid[:, :, -1],
         real_warped_vid=m.real_warped_vid[:, :, -1], fake_out_vid=m.fake_out_vid[:, :, -1],
         fake_warped_vid=m.fake_warped_vid[:, :, -1], ref_img_fea_sum=m.ref_img_fea.double().sum(),
         ref_img_fea_slice=m.ref_img_fea[:, ::32, ::4, ::4], pred_x0=m.diffusion.pred_x0, loss=m.loss.detach(),
         rec_loss=m.rec_loss, rec_warp_loss=m.rec_warp_loss, null_cond_mask=m.diffusion.denoise_fn.null_cond_mask, **small)


def train_flops(b=1, t=40, hw=128):
    """FLOPs of ONE training step of the unmodified reference (FlowDiffusion.optimize_parameters: per-frame frozen-LFAE
    pseudo ground truth incl. the encoder it re-runs per frame, UNet forward + backward, per-frame decode of the denoised
    flow) counted by torch.utils.flop_counter on the CPU run: the "reference dataflow" work bench.py prices the training
    throughput against (2 FLOP per multiply-add; convolutions / matmuls only, as the counter defines it)."""
    from torch.utils.flop_counter import FlopCounterMode
    m = ref.vfdm.FlowDiffusion(img_size=hw // 4, num_frames=t, sampling_timesteps=5, timesteps=1000, null_cond_prob=0.1,
                               is_train=True, lr=1e-4, config_pth=synth.CONFIG, pretrained_pth="")
    for net in (m.generator, m.region_predictor, m.bg_predictor):
        net.eval()
        m.set_requires_grad(net, False)
    ref_img, real_vid, cond, _, _ = synth.train_inputs(b, t, hw)
    ref.vfd.tokenize = lambda texts: texts
    ref.vfd.bert_embed = lambda tokens, return_cls_repr=False: cond
    m.set_train_input(ref_img=ref_img, real_vid=real_vid, ref_text=["x"] * b)
    with FlopCounterMode(display=False) as fc:
        m.forward()
        fwd = fc.get_total_flops()
        m.optimizer_diff.zero_grad()
        m.loss.backward()
        m.optimizer_diff.step()
    total = fc.get_total_flops()
    print("reference training step, B=%d T=%d %dx%d: forward (pseudo-GT + UNet + logged decode) %.1f GFLOP, "
          "backward %.1f GFLOP, total %.1f GFLOP per step = %.1f GFLOP per video"
          % (b, t, hw, hw, fwd / 1e9, (total - fwd) / 1e9, total / 1e9, total / 1e9 / b))
    return total / b


def probes(x, n=64, seed=5):
    """Compact but sensitive summary of a big tensor: per-sample (mean, mean |x|, std) + n random projections per sample
    (numpy PCG64 directions over the flattened sample, unit variance) - what the full-size fixtures store instead of the
    tensors themselves."""
    x = x.detach().double().reshape(x.shape[0], -1)
    rng = np.random.Generator(np.random.PCG64(seed))
    d = torch.from_numpy(rng.standard_normal((n, x.shape[1])))
    stats = torch.stack((x.mean(1), x.abs().mean(1), x.std(1)), dim=1)
    return stats, (x @ d.t()) / np.sqrt(x.shape[1])


def c3_steps_case(name, b=16, t=40, s=32, steps=(999, 500, 0)):
    """BASELINE.json configs[2] at FULL size (MHAD, DDPM-1000, batch 16): teacher-forced single sampler steps.  For each
    timestep the reference's p_sample (:737-746) maps a seeded x_t to x_{t-1} with a seeded noise draw; the fixture keeps a
    strided sub-tensor, per-sample statistics and random projections of every result."""
    m = reference_model(s, t, 1000)
    rng = np.random.Generator(np.random.PCG64(31))
    fea = torch.from_numpy(rng.standard_normal((b, 256, s, s)).astype(np.float32))
    cond = torch.from_numpy(rng.standard_normal((b, 768)).astype(np.float32))
    out = {}
    for i, step in enumerate(steps):
        x_t = torch.from_numpy(rng.standard_normal((b, 3, t, s, s)).astype(np.float32))
        noise = torch.from_numpy(rng.standard_normal((b, 3, t, s, s)).astype(np.float32))
        randn_like = torch.randn_like
        torch.randn_like = lambda x, **k: noise.clone()
        try:
            y = m.diffusion.p_sample(x_t, torch.full((b,), step, dtype=torch.long), fea, cond=cond, cond_scale=1.0)
        finally:
            torch.randn_like = randn_like
        st, pr = probes(y)
        out["x_prev_%d" % i], out["stats_%d" % i], out["probes_%d" % i] = y[:, :, ::8, ::4, ::4].clone(), st, pr
        print("c3 step t=%d done" % step, flush=True)
    save(name, b=b, t=t, s=s, steps=np.array(steps), input_seed=31, **out)


def c3_full_schedule_case(name, b=2, t=40, s=32, hw=128, timesteps=1000, every=50, keep=(750, 500, 250)):
    """BASELINE.json configs[2] END TO END: the reference's p_sample_loop (:748-759) over ALL 1000 timesteps (MHAD shape class,
    40 frames, 32x32 latent, B = 2) on a recorded noise tape, then the per-frame LFAE decode of sample_one_video.  Besides the
    final tensors the fixture keeps the state x_t ENTERING the steps t in `keep` (strided sub-tensor) and statistics / random
    projections of x_t every `every` steps, so that a parity miss shows where the drift enters (~1 h on 8 cores)."""
    m = reference_model(s, t, timesteps, timesteps)
    assert not m.diffusion.is_ddim_sampling
    img, cond = synth.inputs(b, hw)
    m.set_sample_input(sample_img=img, sample_text=cond)
    inner = m.diffusion.p_sample
    trace = {}

    def p_sample(x, tt, *a, **k):
        step = int(tt[0])
        if step % every == 0 or step in keep:
            st, pr = probes(x)
            trace["xt_stats_%d" % step], trace["xt_probes_%d" % step] = st, pr
            if step in keep:
                trace["xt_%d" % step] = x[:, :, ::8, ::4, ::4].clone()
            print("c3 full schedule: entering t=%d  mean|x| %.4f" % (step, float(x.abs().mean())), flush=True)
        return inner(x, tt, *a, **k)

    m.diffusion.p_sample = p_sample
    with patched_noise(synth.NoiseTape(11)), torch.no_grad():
        m.sample_one_video(cond_scale=1.0)
    vf = np.array([0, 20, 39])
    pred = torch.cat((m.sample_vid_grid, m.sample_vid_conf * 2 - 1), dim=1)
    st, pr = probes(pred)
    so, po = probes(m.sample_out_vid)
    save(name, b=b, t=t, s=s, hw=hw, steps=timesteps, timesteps=timesteps, noise_seed=11, video_frames=vf, every=every, keep=np.array(keep),
         sample_vid_grid=m.sample_vid_grid[:, :, :, ::2, ::2], sample_vid_conf=m.sample_vid_conf[:, :, :, ::2, ::2],
         sample_out_vid=m.sample_out_vid[:, :, vf][..., ::2, ::2], sample_warped_vid=m.sample_warped_vid[:, :, vf][..., ::2, ::2],
         pred_stats=st, pred_probes=pr, out_stats=so, out_probes=po, **trace)


def train_full_case(name, b=4, t=40, hw=128):
    """BASELINE.json configs[3] per-GPU shape class at full T: one reference training step (B=4, T=40, 128x128)."""
    labels = (["label a", "None", "label c", "label d"] * 2)[:b]
    train_case(name, b, t, hw, labels, compact=True)


def c5_case(name, b=1, t=40, s=64, hw=256, steps=10, stride=2):
    """BASELINE.json configs[4] geometry at FULL frame count: NATOPS variant (learned null condition, nearest-upsample +
    reflect-padded Upsample), 64x64 latent, 256x256 frames, 40 frames, DDIM (10 of the 50 steps: minutes on this CPU)."""
    variant = dict(learn_null_cond=True, use_deconv=False, padding_mode="reflect")
    m = reference_model(s, t, steps, 1000, **variant)
    img, cond = synth.inputs(b, hw)
    m.set_sample_input(sample_img=img, sample_text=cond)
    with patched_noise(synth.NoiseTape(11)), torch.no_grad():
        m.sample_one_video(cond_scale=1.0)
    vf = np.array([0, 20, 39])
    st, pr = probes(m.sample_out_vid)
    q = stride          # sub-sampling of the stored tensors (the statistics / projections cover every element)
    save(name, b=b, t=t, s=s, hw=hw, steps=steps, timesteps=1000, noise_seed=11, video_frames=vf, stride=q,
         sample_vid_grid=m.sample_vid_grid[:, :, :, ::q, ::q], sample_vid_conf=m.sample_vid_conf[:, :, :, ::q, ::q],
         sample_out_vid=m.sample_out_vid[:, :, vf][..., ::q, ::q], sample_warped_vid=m.sample_warped_vid[:, :, vf][..., ::q, ::q],
         out_stats=st, out_probes=pr)


def lfae_train_case(name, kind):
    """One LFAE stage-1 training step of the reference: ReconstructionModel.forward (LFAE/modules/model.py:162-217) on the unmodified
    Generator / RegionPredictor / BGMotionPredictor in train() mode (BatchNorm batch statistics), sum of the loss terms, backward,
    Adam(lr, betas=(0.5, 0.999)).step() as LFAE/train.py:38-40,96-104 does.  Synthetic checkpoints (tests/synth.lfae_states), the
    VGG-19 of the perceptual loss = oracle/ref_shims/torchvision (torchvision's architecture, synthetic weights: no network), the two
    torch.normal draws of the equivariance Transform replaced by recorded tensors."""
    import importlib
    model_mod = importlib.import_module("LFAE.modules.model")
    rp_mod = importlib.import_module("LFAE.modules.region_predictor")
    bg_mod = importlib.import_module("LFAE.modules.bg_motion_predictor")
    mp, tp, hw, b = synth.lfae_train_setup(kind)
    gen = ref.generator.Generator(num_regions=mp["num_regions"], num_channels=mp["num_channels"], revert_axis_swap=mp["revert_axis_swap"],
                                  **mp["generator_params"])
    reg = rp_mod.RegionPredictor(num_regions=mp["num_regions"], num_channels=mp["num_channels"], estimate_affine=mp["estimate_affine"],
                                 **mp["region_predictor_params"])
    bgp = bg_mod.BGMotionPredictor(num_channels=mp["num_channels"], **mp["bg_predictor_params"])
    gsd, rsd, bsd = synth.lfae_states(mp)
    gen.load_state_dict(gsd)
    reg.load_state_dict(rsd)
    bgp.load_state_dict(bsd)
    for net in (gen, reg, bgp):
        net.train()
    model = model_mod.ReconstructionModel(reg, bgp, gen, tp)
    opt = torch.optim.Adam(list(gen.parameters()) + list(reg.parameters()) + list(bgp.parameters()), lr=tp["lr"], betas=(0.5, 0.999))
    src, drv, theta, tps = synth.lfae_train_inputs(b, hw, tp)
    draws = [theta, tps]
    normal = torch.normal
    torch.normal = lambda *a, **k: draws.pop(0).clone()
    try:
        opt.zero_grad()
        losses, generated = model({"source": src, "driving": drv})
        vals = [v.mean() for v in losses.values()]
        loss = sum(vals)
        loss.backward()
    finally:
        torch.normal = normal
    assert not draws
    rng = np.random.Generator(np.random.PCG64(78))
    names, gnorm, gprobe, small = [], [], [], {}
    nets = (("generator", gen), ("region_predictor", reg), ("bg_predictor", bgp))
    for nn_, net in nets:
        for k, p_ in net.named_parameters():
            names.append(nn_ + "/" + k)
            g = p_.grad.detach().double() if p_.grad is not None else torch.zeros_like(p_).double()
            probe = torch.from_numpy(rng.standard_normal(p_.numel())).view_as(g)
            gnorm.append(float(g.norm()))
            gprobe.append(float((g * probe).sum()))
            if p_.numel() <= 64 and p_.grad is not None:
                small["grad/" + nn_ + "/" + k] = p_.grad.detach().clone()
    opt.step()
    pnorm = [float(p_.detach().double().norm()) for _, net in nets for _, p_ in net.named_parameters()]
    bn = {"bn/" + nn_ + "/" + k: v.detach().clone() for nn_, net in nets for k, v in net.state_dict().items()
          if k.endswith("running_mean") and v.numel() <= 64}
    sub = (lambda v: v.detach()) if kind == "tiny" else (lambda v: v.detach()[:, :, ::4, ::4].clone())
    save(name, kind=np.array(kind), b=b, hw=hw, names=np.array(names), grad_norm=np.array(gnorm), grad_probe=np.array(gprobe),
         param_norm_after=np.array(pnorm), loss_names=np.array(list(losses.keys())), losses=np.array([float(v) for v in vals]),
         prediction=sub(generated["prediction"]), deformed=sub(generated["deformed"]), occlusion_map=generated["occlusion_map"].detach(),
         optical_flow=generated["optical_flow"].detach(), driving_shift=generated["driving_region_params"]["shift"].detach(),
         driving_affine=generated["driving_region_params"]["affine"].detach(), transformed_frame=sub(generated["transformed_frame"]),
         **small, **bn)


def op_cases():
    g = torch.Generator().manual_seed(21)
    emb = torch.randn(32, 8, generator=g)
    rpb = ref.vfd.RelativePositionBias(heads=8, max_distance=32)
    with torch.no_grad():
        rpb.relative_attention_bias.weight.copy_(emb)
        bias40 = rpb(40, device="cpu")
    import rotary_embedding_torch as rot
    q = torch.randn(2, 3, 8, 40, 32, generator=g)
    rq = rot.RotaryEmbedding(32).rotate_queries_or_keys(q)
    x = torch.randn(3, 7000, generator=g) * 2
    quant = torch.quantile(x.abs(), 0.9, dim=-1)
    gd = ref.vfd.GaussianDiffusion(torch.nn.Identity(), image_size=8, num_frames=4, timesteps=1000, sampling_timesteps=100)
    sched = {k: v for k, v in gd.state_dict().items()}
    times = torch.linspace(0., 1000, steps=102)[:-1].int()
    save("ops", emb=emb, bias40=bias40, rot_q=q, rot_out=rq, quant_in=x, quant_out=quant, ddim100_times=times, **sched)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c2", action="store_true")
    ap.add_argument("--train", action="store_true", help="only the DM training-step fixture")
    ap.add_argument("--variants", action="store_true", help="only the variant fixtures: static clipping, use_residual_flow (sampling and "
                    "training), stochastic null conditioning (0 < null_cond_prob < 1)")
    ap.add_argument("--focus", action="store_true", help="only the fixtures of the branches no LFDM script takes: focus_present_mask (unet_tiny_focus), Generator(skips=False)")
    ap.add_argument("--lfae-train", choices=["tiny", "mug128", "both"], help="only the LFAE stage-1 training-step fixtures (lfae_train.py)")
    ap.add_argument("--train-flops", action="store_true", help="count the FLOPs of one reference training step (B=1, T=40, 128x128); writes nothing")
    ap.add_argument("--full", choices=["c3", "c3full", "c4", "c4b8", "c5", "c5d50", "c5b4"], help="one full-size fixture of the other BASELINE.json configurations (minutes each)")
    args = ap.parse_args()
    torch.manual_seed(0)
    if args.full:
        {"c3": lambda: c3_steps_case("c3_ddpm_steps_b16"),
         "c3full": lambda: c3_full_schedule_case("sample_ddpm1000_c3_b2"),      # configs[2]'s whole 1000-step schedule, ~1 h here "c4": lambda: train_full_case("train_step_c4_b4_t40"),
         "c4b8": lambda: train_full_case("train_step_c4_b8_t40", b=8),
         "c5": lambda: c5_case("sample_ddim10_c5_256"),
         "c5b4": lambda: c5_case("sample_ddim50_c5_256_b4", b=4, steps=50, stride=4),      # configs[4] at its per-GPU batch (32 videos over 8 GPUs), ~35 min here
         "c5d50": lambda: c5_case("sample_ddim50_c5_256", steps=50)}[args.full]()      # the configuration's real step count (~8 min here)
        return
    if args.lfae_train:
        for kind in (("tiny", "mug128") if args.lfae_train == "both" else (args.lfae_train,)):
            lfae_train_case("lfae_train_" + kind, kind)
        return
    if args.focus:
        unet_focus_case("unet_tiny_focus")
        generator_noskips_case("generator_32_noskips")
        return
    if args.train:
        train_case("train_step_128", 2, 2, 128, ["label a", "None"])
        return
    if args.variants:
        unet_focus_case("unet_tiny_focus")
        sampler_case("sample_ddim5_tiny_static", 2, 4, 8, 32, 5, 1000, static_clip=True)
        sampler_case("sample_ddim5_tiny_resflow", 2, 4, 8, 32, 5, 1000, use_residual_flow=True)
        train_case("train_step_128_resflow_p05", 4, 2, 128, ["label a", "None", "label c", "label d"], null_cond_prob=0.5,
                   use_residual_flow=True)
        return
    if args.train_flops:
        train_flops()
        return
    op_cases()
    unet_case("unet_tiny_deconv", 2, 4, 8)
    unet_case("unet_tiny_upconv_lnc", 2, 4, 8, learn_null_cond=True, use_deconv=False, padding_mode="reflect")
    generator_case("generator_32", 2, 32)
    sampler_case("sample_ddim5_tiny", 2, 4, 8, 32, 5, 1000)
    sampler_case("sample_ddpm8_tiny", 1, 4, 8, 32, 8, 8)
    if args.c2:
        unet_case("unet_c2_deconv", 1, 40, 32)
        generator_case("generator_128", 1, 128)
        sampler_case("sample_ddim100_c2", 1, 40, 32, 128, 100, 1000, video_frames=[0, 13, 26, 39])


if __name__ == "__main__":
    main()
