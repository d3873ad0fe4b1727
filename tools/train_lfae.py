#!/usr/bin/env python
"""LFAE stage-1 training driver (the reference's LFAE/run_mug.py + LFAE/train.py loop) on cvpr23_lfdm_amd.lfae_train.
  python tools/train_lfae.py [--config configs/lfae_128.yaml] [--data-dir DIR] [--batch 16] [--steps 100] [--log-dir out/lfae]
                             [--vgg vgg19.pth] [--checkpoint RegionMM.pth] [--bench]
Without --data-dir the (source, driving) pairs are synthetic (there is no dataset on the GPU boxes); --vgg takes a torchvision vgg19
state dict (the ImageNet weights of the perceptual loss; without it the network is randomly initialised and the run only measures /
smoke-tests).  One process per GPU: launch under `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 ...`, the
batch is per GPU, gradients are all-reduced over RCCL.  --bench: prints one JSON line (training frame pairs / s)."""
import argparse
import json
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvpr23_lfdm_amd import lfae_train, params as P  # noqa: E402


def synthetic_pairs(batch, hw, device, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(batch, 3, hw // 8, hw // 8, generator=g)
    src = torch.nn.functional.interpolate(base, size=(hw, hw), mode="bilinear", align_corners=False)
    drv = (0.8 * torch.roll(src, shifts=(hw // 16, -(hw // 16)), dims=(2, 3)) + 0.2 * torch.rand(batch, 3, hw, hw, generator=g)).clamp(0, 1)
    return {"source": src.to(device), "driving": drv.to(device)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "lfae_128.yaml"))
    ap.add_argument("--data-dir")
    ap.add_argument("--frame-shape", type=int, default=128)
    ap.add_argument("--batch", type=int, default=16, help="frame pairs per GPU per step")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-dir", default="")
    ap.add_argument("--vgg")
    ap.add_argument("--checkpoint")
    ap.add_argument("--bench", action="store_true")
    ap.add_argument("--graph", action="store_true", help="forward + backward of a step as one replayed hipGraph (single process)")
    args = ap.parse_args()
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("train_lfae.py needs a GPU (the convolutions have no CPU path)")
    torch.cuda.set_device(local % torch.cuda.device_count())
    dev = "cuda"
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl" if torch.cuda.device_count() >= world else "gloo")
    with open(args.config) as f:
        cfg = yaml.safe_load(f)
    torch.manual_seed(1234 + rank)
    gen, reg, bgp = lfae_train.build_from_config(cfg)
    vgg = lfae_train.Vgg19()
    if args.vgg:
        vgg.load_state_dict(P.vgg19_from_torchvision(torch.load(args.vgg, map_location="cpu")), strict=False)
    else:
        vgg.load_state_dict(P.synthetic_vgg19_state())
    trainer = lfae_train.LFAETrainer(gen, reg, bgp, cfg["model_params"], cfg["train_params"], vgg=vgg).to(dev)
    if args.checkpoint:
        trainer.load_state_dict(torch.load(args.checkpoint, map_location="cpu"))
    trainer.enable_data_parallel()
    loader = None
    if args.data_dir:
        aug = cfg.get("dataset_params", {}).get("augmentation_params", {})
        ds = lfae_train.FramePairs(args.data_dir, frame_shape=args.frame_shape, jitter=aug.get("jitter_param"),
                                   horizontal_flip=aug.get("flip_param", {}).get("horizontal_flip", True),
                                   time_flip=aug.get("flip_param", {}).get("time_flip", True), seed=rank)
        loader = iter(torch.utils.data.DataLoader(ds, batch_size=args.batch, shuffle=True, drop_last=True, num_workers=2))
    t0, timed = None, 0
    for it in range(args.warmup + args.steps):
        if it == args.warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if loader is not None:
            try:
                x = next(loader)
            except StopIteration:
                trainer.end_epoch()
                loader = iter(torch.utils.data.DataLoader(ds, batch_size=args.batch, shuffle=True, drop_last=True, num_workers=2))
                x = next(loader)
            x = {k: v.to(dev) for k, v in x.items() if k in ("source", "driving")}
        else:
            x = synthetic_pairs(args.batch, args.frame_shape, dev, seed=1000 * rank + it)
        losses, _ = (trainer.step_graphed(x) if args.graph and world == 1 else trainer.step(x))
        if it >= args.warmup:
            timed += 1
        if rank == 0 and not args.bench and it % cfg["train_params"].get("print_freq", 10) == 0:
            print("iter %d  loss %.4f  loss_perc %.4f  loss_shift %.4f  loss_affine %.4f" % (
                it, float(losses["total"]), float(losses.get("perceptual", 0)), float(losses.get("equivariance_shift", 0)),
                float(losses.get("equivariance_affine", 0))), flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        if args.log_dir:
            os.makedirs(args.log_dir, exist_ok=True)
            torch.save(trainer.state_dict(), os.path.join(args.log_dir, "RegionMM.pth"))
        print(json.dumps({"metric": "LFAE stage-1 training frame pairs / s", "value": round(timed * args.batch * world / dt, 2),
                          "ms_per_step": round(1e3 * dt / timed, 1), "batch_per_gpu": args.batch, "n_gpus": world, "frame": args.frame_shape,
                          "graphed": bool(args.graph and world == 1), "steps": timed, "loss_last": round(float(losses["total"]), 4), "data": "synthetic" if loader is None else args.data_dir,
                          "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))


if __name__ == "__main__":
    main()
