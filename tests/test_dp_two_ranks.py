"""Data-parallel DM training, two ranks against one: `FlowDiffusion` under `torchrun --nproc-per-node 2` (no call of
enable_data_parallel by the caller: the wrapper starts it itself, as it must for an unchanged reference training script)
shards the global batch in set_train_input, all-reduces the flat gradient from autograd hooks (GradAllReduce) and applies
the fused Adam step on every rank - and after two optimizer steps both ranks hold the parameters of a single process that
trained on the whole batch (mean of shard means = global mean; the step's random draws are made for the global batch and
sliced).  Reference semantics: nn.DataParallel scatter / gather, DM/train_video_flow_diffusion_mhad_multiGPU.py:207,249-254.

GPU flavour (`-m gpu`): both ranks on cuda:0 with the gloo backend - the only multi-rank evidence obtainable on a 1-GPU box
(RCCL itself needs two devices); the HIP kernels, the hooks and the collective run exactly as under RCCL otherwise.
CPU flavour: the same worker on the emulation build (opt-in, LFDM_DP_EMU=1: ~4 minutes)."""
import json
import os
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
repo, kind = sys.argv[1], sys.argv[2]
sys.path.insert(0, repo); sys.path.insert(0, os.path.join(repo, "tests"))
import torch
import synth
from cvpr23_lfdm_amd import FlowDiffusion, _build, _native
if kind == "emu":
    _native._set_library_for_tests(_native.NativeLibrary(_build.build_emu(), "emu"))
dev = "cuda" if kind == "hip" else "cpu"
if kind == "hip" and os.environ.get("LFDM_DP_ONE_RANK_PER_DEVICE") == "1":      # the RCCL flavour: rank r owns GPU r (RCCL refuses duplicate devices)
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
B, T, HW = int(os.environ.get("LFDM_DP_B", "2")), 2, 128
torch.manual_seed(11)
m = FlowDiffusion(img_size=HW // 4, num_frames=T, sampling_timesteps=5, null_cond_prob=0.5, is_train=True, lr=1e-4,
                  config_pth=synth.CONFIG, pretrained_pth="")
m.unet.load_state_dict(synth.unet_state())
m.generator.load_state_dict(synth.generator_state()); m.region_predictor.load_state_dict(synth.region_state())
m.bg_predictor.load_state_dict(synth.bg_state())
for net in (m.generator, m.region_predictor, m.bg_predictor):
    net.eval(); m.set_requires_grad(net, False)
m.to(dev)
table = {"anger": torch.randn(768, generator=torch.Generator().manual_seed(1)), "fear": torch.randn(768, generator=torch.Generator().manual_seed(2)),
         "None": torch.zeros(768)}
m.diffusion.text_encoder = lambda texts: torch.stack([table[t] for t in texts])
rank = int(os.environ.get("RANK", "0"))
if rank == 1:                      # a rank that starts from DIFFERENT weights must be pulled onto rank 0's by the start-up broadcast
    with torch.no_grad():
        m.unet.get("init_conv.bias").add_(0.25)
ref_img, real_vid, _, _, _ = synth.train_inputs(B, T, HW)
p0 = m.unet.get("mid_block1.block1.proj.weight").detach().clone()
torch.manual_seed(1234)
losses, masks = [], []
for step in range(2):
    m.set_train_input(ref_img=ref_img.to(dev), real_vid=torch.roll(real_vid, step, dims=2).to(dev), ref_text=(["anger", "None", "fear"] if step == 0 else ["fear", "anger", "anger"])[:B])
    m.optimize_parameters()
    losses.append(float(m.loss)); masks.append(m.unet.null_cond_mask.cpu().tolist())
names = ["init_conv.bias", "mid_block1.block1.proj.weight", "final_conv.1.weight", "downs.0.2.fn.fn.to_qkv.weight", "ups.3.0.mlp.1.bias"]
out = {"rank": rank, "world": int(os.environ.get("WORLD_SIZE", "1")), "losses": losses, "masks": masks, "shard_batch": int(m.real_vid.shape[0]),
       "params": {k: m.unet.get(k).detach().double().flatten()[:4096].cpu().tolist() for k in names},
       "moved": float((m.unet.get("mid_block1.block1.proj.weight").detach().cpu() - p0.cpu()).abs().mean()),
       "checksum_spread": m._dp.replica_checksum() if m._dp is not None else 0.0}
open(os.path.join(os.environ["LFDM_DP_OUT"], "rank%d.json" % rank), "w").write(json.dumps(out))      # (two ranks' long stdout lines interleave)
import torch.distributed as dist
if dist.is_initialized():
    dist.barrier(); dist.destroy_process_group()
'''


def _launch(tmp_path, kind, world, batch=2, backend="gloo"):
    script = tmp_path / "dp_worker.py"
    script.write_text(WORKER)
    outdir = tmp_path / ("out_w%d_b%d_%s" % (world, batch, backend))
    outdir.mkdir()
    env = dict(os.environ, LFDM_DIST_BACKEND=backend, MASTER_ADDR="127.0.0.1", LFDM_DP_OUT=str(outdir), LFDM_DP_B=str(batch))
    if backend == "nccl":      # RCCL: one rank per GPU; the host driver only supports dmabuf IPC
        env.update(LFDM_DP_ONE_RANK_PER_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None)
    if world == 1:
        cmd = [sys.executable, str(script), REPO, kind]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", "29581", str(script), REPO, kind]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=3000)
    assert r.returncode == 0, r.stdout[-4000:]
    outs = [json.loads((outdir / f).read_text()) for f in sorted(os.listdir(outdir))]
    assert len(outs) == world, r.stdout[-2000:]
    return sorted(outs, key=lambda o: o["rank"])


def _check(tmp_path, kind, backend="gloo"):
    single = _launch(tmp_path, kind, 1)[0]
    r0, r1 = _launch(tmp_path, kind, 2, backend=backend)
    assert single["shard_batch"] == 2 and r0["shard_batch"] == r1["shard_batch"] == 1 and r0["world"] == 2
    assert r0["checksum_spread"] == 0.0 and r1["checksum_spread"] == 0.0          # replicas bit-identical after two steps
    assert r0["params"] == r1["params"]
    assert r0["masks"][0] + r1["masks"][0] == single["masks"][0] and r0["masks"][1] + r1["masks"][1] == single["masks"][1]   # "None" + drawn null masks, sliced
    lr = 1e-4
    assert single["moved"] > 0.5 * lr                                                 # Adam really stepped (2 steps of ~lr each)
    for k, v in single["params"].items():
        a, b = torch.tensor(r0["params"][k]), torch.tensor(v)
        d = (a - b).abs()
        # Adam's first steps are ~lr * sign(g): identical except where a gradient is numerically zero
        assert float(d.mean()) < 0.02 * lr and float((d > 0.5 * lr).float().mean()) < 0.01, (k, float(d.mean()), float(d.max()))
    for s in range(2):                                                                # loss of step 2 was computed with step 1's update
        mean_of_shards = 0.5 * (r0["losses"][s] + r1["losses"][s])
        assert abs(mean_of_shards - single["losses"][s]) <= 2e-4 * abs(single["losses"][s]), (s, mean_of_shards, single["losses"][s])


def _check_uneven(tmp_path, kind, batch, shards):
    """A batch the ranks cannot split evenly (the reference scripts use BATCH_SIZE = 5 and no drop_last; nn.DataParallel scatters
    it unevenly): tensor_split shards, each rank's loss weighted by shard * world / batch - the update is the single process's."""
    single = _launch(tmp_path, kind, 1, batch)[0]
    r0, r1 = _launch(tmp_path, kind, 2, batch)
    assert (r0["shard_batch"], r1["shard_batch"]) == shards and r0["checksum_spread"] == 0.0 and r0["params"] == r1["params"]
    assert r0["masks"][0] + r1["masks"][0] == single["masks"][0]
    lr = 1e-4
    for k, v in single["params"].items():
        d = (torch.tensor(r0["params"][k]) - torch.tensor(v)).abs()
        assert float(d.mean()) < 0.02 * lr and float((d > 0.5 * lr).float().mean()) < 0.01, (k, float(d.mean()), float(d.max()))
    for s in range(2):
        weighted = (shards[0] * r0["losses"][s] + shards[1] * r1["losses"][s]) / batch
        assert abs(weighted - single["losses"][s]) <= 2e-4 * abs(single["losses"][s]), (s, weighted, single["losses"][s])


@pytest.mark.gpu
def test_two_ranks_uneven_batch(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _check_uneven(tmp_path, "hip", 3, (2, 1))


@pytest.mark.gpu
def test_two_ranks_one_video(tmp_path):
    """Batch 1 on two ranks: rank 1 has no video - it skips the model, keeps its random generator in step and contributes zero
    gradients; both ranks still apply the single process's update."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _check_uneven(tmp_path, "hip", 1, (1, 0))


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_match_single_process(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _check(tmp_path, "hip")


@pytest.mark.gpu
def test_two_ranks_rccl_one_rank_per_gpu(tmp_path):
    """The same two-rank step over RCCL (backend "nccl" on ROCm), one rank per device - the path `bench.py --gpus N` and an unchanged
    training script under torchrun take on a multi-GPU node (reference: nn.DataParallel over the node's GPUs,
    DM/train_video_flow_diffusion_mhad_multiGPU.py:207,249-254).  Needs two GPUs: on the one-GPU box of this pool it SKIPS, so the
    first box with two devices exercises RCCL without anybody having to remember to."""
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n < 2:
        pytest.skip("RCCL needs one device per rank: %d GPU(s) visible (the gloo flavour above covers the one-GPU box)" % n)
    _check(tmp_path, "hip", backend="nccl")


@pytest.mark.skipif(os.environ.get("LFDM_DP_EMU", "0") != "1", reason="the same check on the emulation build: opt-in (LFDM_DP_EMU=1, ~4 min)")
def test_two_ranks_emulator(tmp_path):
    _check(tmp_path, "emu")
