"""LFAE stage-1 training step (cvpr23_lfdm_amd/lfae_train.py) against the UNMODIFIED reference's ReconstructionModel + Adam step
(fixtures minted by `oracle/make_golden.py --lfae-train`, LFAE/train.py:96-104, LFAE/modules/model.py:162-217): the three loss terms,
every parameter's gradient norm and random projection, every parameter's norm after the update, the generated frame and the BatchNorm
running statistics.  'tiny' runs on the x86 emulation of the kernels (CPU), 'mug128' = config/mug128.yaml at 128x128 on the GPU."""
import os

import numpy as np
import pytest
import torch

import synth
from cvpr23_lfdm_amd import Generator, lfae_train
from cvpr23_lfdm_amd.flow_diffusion import BGMotionPredictor, RegionPredictor
from util import assert_close

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _build(kind, dev):
    mp, tp, hw, b = synth.lfae_train_setup(kind)
    gen = Generator(num_regions=mp["num_regions"], num_channels=mp["num_channels"], revert_axis_swap=mp["revert_axis_swap"],
                    **mp["generator_params"])
    reg = RegionPredictor(num_regions=mp["num_regions"], num_channels=mp["num_channels"], estimate_affine=mp["estimate_affine"],
                          **mp["region_predictor_params"])
    bgp = BGMotionPredictor(num_channels=mp["num_channels"], **mp["bg_predictor_params"])
    gsd, rsd, bsd = synth.lfae_states(mp)
    gen.load_state_dict(gsd)
    reg.load_state_dict(rsd)
    bgp.load_state_dict(bsd)
    vgg = lfae_train.Vgg19()
    vgg.load_state_dict(synth.vgg_state())
    trainer = lfae_train.LFAETrainer(gen, reg, bgp, mp, tp, vgg=vgg).to(dev)
    return trainer, (mp, tp, hw, b)


def _check(kind, dev, tol, out_tol=None):
    """tol: losses and gradient quantities (DESIGN.md section 4 says why those stay at 5e-3 on the GPU); out_tol: every generated tensor and
    BatchNorm running statistic (default: tol)."""
    out_tol = tol if out_tol is None else out_tol
    gold = np.load(os.path.join(GOLDEN, "lfae_train_%s.npz" % kind))
    trainer, (mp, tp, hw, b) = _build(kind, dev)
    src, drv, theta, tps = synth.lfae_train_inputs(b, hw, tp)
    losses, gen = trainer.step({"source": src.to(dev), "driving": drv.to(dev)}, transform_noise=(theta, tps))
    for name, ref in zip(gold["loss_names"], gold["losses"]):
        assert abs(float(losses[str(name)]) - float(ref)) <= tol * max(1.0, abs(float(ref))), (name, float(losses[str(name)]), float(ref))
    sub = (lambda v: v) if kind == "tiny" else (lambda v: v[:, :, ::4, ::4])
    for key, got in (("prediction", sub(gen["prediction"])), ("deformed", sub(gen["deformed"])), ("occlusion_map", gen["occlusion_map"]),
                     ("optical_flow", gen["optical_flow"]), ("driving_shift", gen["driving_region_params"]["shift"]),
                     ("driving_affine", gen["driving_region_params"]["affine"]), ("transformed_frame", sub(gen["transformed_frame"]))):
        assert_close(got.detach().cpu(), torch.from_numpy(gold[key]), out_tol, key)
    nets = {"generator": trainer.generator, "region_predictor": trainer.region_predictor, "bg_predictor": trainer.bg_predictor}
    params = {n + "/" + k: p for n, net in nets.items() for k, p in net.named_parameters()}
    assert list(params) == [str(n) for n in gold["names"]]      # the reference's named_parameters() ORDER: its optimizer state (indexed by position) loads
    rng = np.random.Generator(np.random.PCG64(78))
    gscale = float(np.max(gold["grad_norm"]))
    worst = 0.0
    for name, gn, gp, pn in zip(gold["names"], gold["grad_norm"], gold["grad_probe"], gold["param_norm_after"]):
        p = params[str(name)]
        probe = torch.from_numpy(rng.standard_normal(p.numel())).view(p.shape)
        g = p.grad.detach().double().cpu()
        sc = max(float(gn), 1e-3 * gscale)
        e1 = abs(float(g.norm()) - float(gn)) / sc
        e3 = abs(float(p.detach().double().norm()) - float(pn)) / max(float(pn), 1e-6)
        assert e1 <= 5 * tol, (str(name), "gradient norm", e1)
        assert abs(float((g * probe).sum()) - float(gp)) <= 5 * tol * max(sc * float(probe.norm()), 1e-9), (str(name), "gradient projection")
        # A convolution bias in front of a BatchNorm has an exactly-zero gradient; what both implementations hold there is rounding
        # noise (~1e-8 of the scale), which Adam's g / (|g| + eps) turns into +-lr steps of arbitrary sign: no update to compare.
        if float(gn) > 1e-5 * gscale:
            assert e3 <= tol, (str(name), "parameter norm after the Adam step", e3)
            worst = max(worst, e3)
        worst = max(worst, e1)
    for key in gold.files:
        if key.startswith("grad/"):
            ref = torch.from_numpy(gold[key])
            sc = max(float(ref.abs().max()), 1e-3 * gscale)
            assert_close(params[key[5:]].grad.detach().cpu() / sc, ref / sc, 5 * tol, key)
        if key.startswith("bn/"):
            n, k = key[3:].split("/", 1)
            assert_close(nets[n].state_dict()[k].cpu(), torch.from_numpy(gold[key]), out_tol, key)
    return worst


def test_lfae_train_step_tiny(backend):
    if backend != "cpu":
        pytest.skip("the tiny configuration is the CPU (emulator) case; the GPU runs config/mug128.yaml")
    _check("tiny", "cpu", 2e-3)


@pytest.mark.gpu
def test_lfae_train_step_mug128():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _check("mug128", "cuda", 5e-3, out_tol=1e-3)      # generated tensors / running statistics at the north star's 1e-3 (achieved: 1e-2 of that bar)


def test_checkpoint_format_and_schedule():
    """train.py:136-160 checkpoint keys; MultiStepLR(milestones, 0.1) via end_epoch()."""
    trainer, _ = _build("tiny", "cpu")
    sd = trainer.state_dict()
    assert set(sd) == {"example", "epoch", "generator", "bg_predictor", "region_predictor", "optimizer"}
    assert "first.conv.weight" in sd["generator"] and "regions.weight" in sd["region_predictor"] and "fc.bias" in sd["bg_predictor"]
    lrs = []
    for _ in range(91):
        lrs.append(trainer.end_epoch())
    assert abs(lrs[58] - 2e-4) < 1e-12 and abs(lrs[59] - 2e-5) < 1e-12 and abs(lrs[89] - 2e-6) < 1e-12
    t2, _ = _build("tiny", "cpu")
    t2.load_state_dict(sd)
    for a, b in zip(t2.generator.parameters(), trainer.generator.parameters()):
        assert torch.equal(a, b)


def test_frame_pairs_dataset(tmp_path):
    """FramePairs (LFAE/mug_dataset.py items): walks nested video folders, two distinct frames of ONE video, [0, 1] floats, flips."""
    from PIL import Image
    for vid, n in (("s1/anger/take0", 5), ("s2/take1", 3)):
        d = tmp_path / vid
        d.mkdir(parents=True)
        for i in range(n):
            Image.fromarray(np.full((20, 24, 3), 10 * i + (100 if "s2" in vid else 0), np.uint8)).save(str(d / ("%03d.png" % i)))
    ds = lfae_train.FramePairs(str(tmp_path), frame_shape=16, jitter=None, seed=3)
    assert len(ds) == 2
    for idx in range(2):
        for _ in range(4):
            it = ds[idx]
            assert it["source"].shape == (3, 16, 16) and it["driving"].dtype == torch.float32
            assert 0.0 <= float(it["source"].min()) and float(it["driving"].max()) <= 1.0
            # (the two indices are drawn WITH replacement, mug_dataset.py:94: the same frame twice is a legal item)
            assert os.path.dirname(it["frame"][0]) == os.path.dirname(it["frame"][1]) and it["frame"][0] <= it["frame"][1]
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=2)))
    assert batch["source"].shape == (2, 3, 16, 16)
    # forked DataLoader workers must not replay one another's draws, and a new iterator (= a new epoch in tools/train_lfae.py) must not
    # replay the previous one: 12 items of the 5-frame video per worker and epoch
    big = lfae_train.FramePairs([ds.videos[0]] * 24, frame_shape=16, jitter=None, seed=3)
    seen = []
    for _ in range(2):
        frames = [[], []]
        for i, item in enumerate(torch.utils.data.DataLoader(big, batch_size=1, num_workers=2)):
            frames[i % 2].append((item["frame"][0][0], item["frame"][1][0], float(item["source"][0, 0, 0, 0]) > float(item["source"][0, 0, 0, -1])))
        assert frames[0] != frames[1], "both workers drew the same sequence"
        seen.append(frames)
    assert seen[0] != seen[1], "the second epoch replayed the first"


_DP_WORKER = r'''
import json, os, sys
repo = sys.argv[1]
sys.path.insert(0, repo); sys.path.insert(0, os.path.join(repo, "tests"))
import torch
import torch.distributed as dist
import synth
from cvpr23_lfdm_amd import _build, _native
_native._set_library_for_tests(_native.NativeLibrary(_build.build_emu(), "emu"))
import test_lfae_train as T
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
trainer, (mp, tp, hw, b) = T._build("tiny", "cpu")
if rank == 1:                        # different start: the start-up broadcast must pull it onto rank 0's weights
    with torch.no_grad():
        trainer.generator.get("final.bias").add_(0.5)
trainer.enable_data_parallel()
src, drv, theta, tps = synth.lfae_train_inputs(b, hw, tp, seed=21 + rank)      # every rank its own pairs (weak scaling)
losses, _ = trainer.step({"source": src, "driving": drv}, transform_noise=(theta, tps))
g = [p.grad.detach().double().flatten() for net in (trainer.generator, trainer.region_predictor, trainer.bg_predictor) for p in net.parameters()]
out = {"rank": rank, "loss": float(losses["total"]), "spread": trainer._dp.replica_checksum(),
       "grad_sum": float(sum(float(v.sum()) for v in g)), "final_bias": trainer.generator.get("final.bias").detach().double().tolist()}
open(os.path.join(os.environ["LFAE_DP_OUT"], "rank%d.json" % rank), "w").write(json.dumps(out))
dist.barrier(); dist.destroy_process_group()
'''


def test_lfae_trainer_two_ranks_gloo(tmp_path):
    """One process per GPU (here: two CPU ranks over gloo on the emulator): the start-up broadcast aligns the replicas, each rank trains
    on its own pairs with its own BatchNorm statistics (the reference's nn.DataParallel replicas, use_sync_bn: False), the bucketed
    all-reduce hands both the SAME mean gradient and after the fused Adam step the replicas are bit-identical."""
    import json
    import subprocess
    import sys
    script = tmp_path / "lfae_dp_worker.py"
    script.write_text(_DP_WORKER)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LFAE_DP_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", PYTHONPATH=os.path.join(repo, "tests"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           "29631", str(script), repo]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    a, b = (json.load(open(str(tmp_path / ("rank%d.json" % i)))) for i in (0, 1))
    assert a["spread"] == 0.0 and b["spread"] == 0.0
    assert a["grad_sum"] == b["grad_sum"] and a["final_bias"] == b["final_bias"]
    assert a["loss"] != b["loss"] and np.isfinite(a["loss"]) and np.isfinite(b["loss"])


@pytest.mark.gpu
def test_graphed_step_equals_eager_step():
    """LFAETrainer.step_graphed (forward + backward as one replayed hipGraph, optimizer outside) against the eager step on the same inputs and
    transform noise, eight steps in a row: loss terms and parameters must agree (the ROCm 7.2 graph memset bug of DESIGN.md would show up here
    as a drift after a few replays)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = "cuda"
    ta, (mp, tp, hw, b) = _build("tiny", dev)
    tb, _ = _build("tiny", dev)
    g = torch.Generator().manual_seed(5)
    for it in range(8):
        src, drv = torch.rand(b, 3, hw, hw, generator=g).to(dev), torch.rand(b, 3, hw, hw, generator=g).to(dev)
        noise = ta.draw_transform_noise(b)
        la, _ = ta.step({"source": src, "driving": drv}, transform_noise=noise)
        lb, _ = tb.step_graphed({"source": src, "driving": drv}, transform_noise=noise)
        for k in la:
            assert abs(float(la[k]) - float(lb[k])) <= 1e-5 * max(1.0, abs(float(la[k]))), (it, k, float(la[k]), float(lb[k]))
    assert "graph" in next(iter(tb._graphs.values()))
    pa = torch.cat([p.detach().reshape(-1) for p in ta.optimizer.param_groups[0]["params"]])
    pb = torch.cat([p.detach().reshape(-1) for p in tb.optimizer.param_groups[0]["params"]])
    assert float((pa - pb).abs().max()) <= 1e-5 * float(pa.abs().max())


@pytest.mark.gpu
def test_graphed_step_survives_foreign_work_between_replays():
    """The captured graph holds raw addresses of buffers it does not own (cached Winograd filter packs, the fixed-point scatter accumulator,
    the padded-parameter buffers).  Between two replays: an eval forward, a load_state_dict (torch rewrites every weight: the per-call path
    must refill the packs IN PLACE), an eager step of a LARGER batch (the accumulator grows: the graph must be captured again, not replayed
    on freed memory) - and the graphed trainer must still equal the eager one that saw the same sequence."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cvpr23_lfdm_amd import autograd as A
    dev = "cuda"
    ta, (mp, tp, hw, b) = _build("tiny", dev)
    tb, _ = _build("tiny", dev)
    g = torch.Generator().manual_seed(6)
    draw = lambda n: (torch.rand(n, 3, hw, hw, generator=g).to(dev), torch.rand(n, 3, hw, hw, generator=g).to(dev))

    def both(n, graphed=True):
        src, drv = draw(n)
        noise = ta.draw_transform_noise(n)
        la, _ = ta.step({"source": src, "driving": drv}, transform_noise=noise)
        lb, _ = (tb.step_graphed if graphed else tb.step)({"source": src, "driving": drv}, transform_noise=noise)
        for k in la:
            assert abs(float(la[k]) - float(lb[k])) <= 1e-5 * max(1.0, abs(float(la[k]))), (k, float(la[k]), float(lb[k]))

    for _ in range(4):
        both(b)
    key = next(iter(tb._graphs))
    assert "graph" in tb._graphs[key]
    packs_before = {k: v[2].data_ptr() for k, v in A._PACKS.items()}
    # (1) eval forward in between
    for t in (ta, tb):
        t.model.training = False                 # (ReconstructionModel is a plain object: `training` selects BatchNorm's running statistics)
        src, drv = draw(b)            # (no torch.no_grad(): the equivariance terms take autograd.grad of the transform inside the forward)
        t.model({"source": src, "driving": drv}, transform_noise=ta.draw_transform_noise(b))
        t.model.training = True
    both(b)
    # (2) load_state_dict: same values, but torch bumps every _version -> every cached pack is stale by tag
    for t in (ta, tb):
        for net in (t.generator, t.region_predictor, t.bg_predictor):
            net.load_state_dict({k: v.clone() for k, v in net.state_dict().items()})
    both(b)
    assert all(A._PACKS[k][2].data_ptr() == p for k, p in packs_before.items() if k in A._PACKS), "a cached pack was rebound, not refilled"
    # (3) an eager step of twice the batch (grows the scatter accumulator), then graphed steps of the captured shape again
    both(2 * b, graphed=False)
    both(b)
    both(b)
    pa = torch.cat([p.detach().reshape(-1) for p in ta.optimizer.param_groups[0]["params"]])
    pb = torch.cat([p.detach().reshape(-1) for p in tb.optimizer.param_groups[0]["params"]])
    assert float((pa - pb).abs().max()) <= 1e-5 * float(pa.abs().max())
