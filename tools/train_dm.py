#!/usr/bin/env python
"""The reference's DM training driver (DM/train_video_flow_diffusion_{mug,mhad,natops}.py:130-420 and the _multiGPU variant)
on this framework: same loop, checkpoint format ({'example', 'diffusion', 'optimizer_diff'}), MultiStepLR schedule and
restore semantics; one process per GPU under torchrun instead of nn.DataParallel threads.  Needs GPUs.

    python tools/train_dm.py --data DIR --lfae-ckpt RegionMM.pth --bert /data/bert-base-cased --out snapshots
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_dm.py --data DIR ...
    python tools/train_dm.py --synthetic --final-step 20          # random videos / random-init LFAE: exercises the loop
"""
import argparse
import math
import os
import sys
import timeit

import numpy as np
import torch
import torch.distributed as dist
from torch.optim.lr_scheduler import MultiStepLR
from torch.utils import data

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvpr23_lfdm_amd import FlowDiffusion, io_compat as C  # noqa: E402
from cvpr23_lfdm_amd.data import FrameFolderVideos, SyntheticVideos  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data", default="")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "lfae_128.yaml"))
    ap.add_argument("--lfae-ckpt", default="")
    ap.add_argument("--bert", default=os.environ.get("LFDM_BERT_PATH"))
    ap.add_argument("--restore-from", default="")
    ap.add_argument("--set-start", action="store_true")
    ap.add_argument("--out", default="snapshots")
    ap.add_argument("--batch-size", type=int, default=8, help="per GPU")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--null-cond-prob", type=float, default=0.1)
    ap.add_argument("--epoch-milestones", type=int, nargs="*", default=[800, 1000])
    ap.add_argument("--final-step", type=int, default=200000)
    ap.add_argument("--save-freq", type=int, default=2000)
    ap.add_argument("--print-freq", type=int, default=10)
    ap.add_argument("--save-img-freq", type=int, default=500)
    ap.add_argument("--num-workers", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1234)
    args = ap.parse_args()
    if not torch.cuda.is_available():
        sys.exit("tools/train_dm.py needs a GPU: the training step is liblfdm_hip.so only")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local % torch.cuda.device_count())
    if world > 1:
        dist.init_process_group("nccl")                      # RCCL
    # every rank must build the SAME initial UNet (parameters are initialised from torch's global generator and the
    # data-parallel step only averages gradients): common seed for construction, per-rank streams afterwards
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    os.makedirs(args.out, exist_ok=True)

    model = FlowDiffusion(lr=args.lr, is_train=True, img_size=args.size // 4, num_frames=args.frames,
                          null_cond_prob=args.null_cond_prob, sampling_timesteps=1000, config_pth=args.config,
                          pretrained_pth=args.lfae_ckpt, bert_path=None if args.synthetic else args.bert)     # :153-162
    model.cuda()
    if args.synthetic:
        emb = {}
        model.diffusion.text_encoder = lambda texts: torch.stack(
            [emb.setdefault(t, torch.randn(768, generator=torch.Generator().manual_seed(len(emb) + 1))) for t in texts])
        for net in (model.generator, model.region_predictor, model.bg_predictor):
            net.eval()
            model.set_requires_grad(net, False)
    start_step = 0
    if args.restore_from:                                                                                     # :169-184
        ck = torch.load(args.restore_from, map_location="cpu")
        if args.set_start:
            start_step = int(math.ceil(ck["example"] / (args.batch_size * world)))
        model.diffusion.load_state_dict(ck["diffusion"])
        if "optimizer_diff" in ck:
            model.optimizer_diff.load_state_dict(ck["optimizer_diff"])
        print("=> loaded checkpoint '%s' (step %d)" % (args.restore_from, start_step))
    if world > 1:
        # bucketed RCCL all-reduce of the flat gradient, overlapped with backward; rank 0's parameters and Adam
        # moments are broadcast once so the replicas start identical whatever happened above (restore on one rank ...)
        model.enable_data_parallel(shard_inputs=False)      # the DistributedSampler below already feeds rank-local batches
    torch.manual_seed(args.seed + 1 + rank)     # per-rank streams for t / noise / null-cond mask / data order
    np.random.seed(args.seed + 1 + rank)

    ds = SyntheticVideos(n=256, image_size=args.size, num_frames=args.frames) if args.synthetic else \
        FrameFolderVideos(args.data, image_size=args.size, num_frames=args.frames, sampling="random", jitter=True)
    sampler = data.distributed.DistributedSampler(ds, world, rank, shuffle=True, seed=args.seed) if world > 1 else None
    loader = data.DataLoader(ds, batch_size=args.batch_size, shuffle=sampler is None, sampler=sampler,
                             num_workers=args.num_workers, pin_memory=True, drop_last=True)
    per_epoch = max(1, len(loader))
    epoch = start_step // per_epoch
    sched = MultiStepLR(model.optimizer_diff, args.epoch_milestones, gamma=0.1, last_epoch=epoch - 1)         # :210-211
    step, t0 = start_step, timeit.default_timer()
    while step < args.final_step:
        if sampler is not None:
            sampler.set_epoch(epoch)
        for real_vids, ref_texts, real_names in loader:
            real_vids = real_vids.cuda(non_blocking=True)
            ref_imgs = real_vids[:, :, 0].clone().detach()                    # first frame = reference frame (:221)
            model.set_train_input(ref_img=ref_imgs, real_vid=real_vids, ref_text=list(ref_texts))
            model.optimize_parameters()
            step += 1
            if rank == 0 and step % args.print_freq == 0:
                dt = (timeit.default_timer() - t0) / args.print_freq
                t0 = timeit.default_timer()
                print("iter %d/%d  loss %.7f  loss_rec %.4f  loss_warp %.4f  lr %.2e  %.1f videos/s" % (
                    step, args.final_step, float(model.loss), float(model.rec_loss), float(model.rec_warp_loss),
                    model.optimizer_diff.param_groups[0]["lr"], args.batch_size * world / dt), flush=True)
            if rank == 0 and step % args.save_img_freq == 0:                   # the middle-frame panel of :246-275
                mid, s = args.frames // 2, args.size
                panel = np.zeros((2 * s, 4 * s, 3), np.uint8)
                for col, (top, bot) in enumerate([(ref_imgs, real_vids[:, :, mid]), (model.real_out_vid[:, :, mid], model.real_warped_vid[:, :, mid]),
                                                  (model.fake_out_vid[:, :, mid], model.fake_warped_vid[:, :, mid])]):
                    panel[:s, col * s:(col + 1) * s] = C.sample_img(top)
                    panel[s:, col * s:(col + 1) * s] = C.sample_img(bot)
                panel[:s, 3 * s:] = C.grid2fig(model.real_vid_grid[0, :, mid].permute(1, 2, 0).data.cpu().numpy(), grid_size=s // 4, img_size=s)
                panel[s:, 3 * s:] = C.grid2fig(model.fake_vid_grid[0, :, mid].permute(1, 2, 0).data.cpu().numpy(), grid_size=s // 4, img_size=s)
                C.imsave(os.path.join(args.out, "B%04d_S%06d_%s.png" % (args.batch_size, step, real_names[0])), panel)
            if rank == 0 and (step % args.save_freq == 0 or step >= args.final_step):                         # :330-340
                torch.save({"example": step * args.batch_size * world, "diffusion": model.diffusion.state_dict(),
                            "optimizer_diff": model.optimizer_diff.state_dict()},
                           os.path.join(args.out, "flowdiff_%04d_S%06d.pth" % (args.batch_size, step)))
            if step >= args.final_step:
                break
        epoch += 1
        sched.step()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
