#!/usr/bin/env python
"""The reference's demo flow (demo/demo_mug.py:60-146, demo_mhad.py, demo_natops.py) on this framework: load LFAE + DM
checkpoints, read one reference image, sample a 40-frame video per text prompt and write the five-panel GIF
[source | generated | warped source | flow grid | occlusion].  Needs a GPU (liblfdm_hip.so; there is no CPU path).

    python tools/demo.py --config configs/lfae_128.yaml --lfae-ckpt RegionMM.pth --dm-ckpt flowdiff.pth \
        --bert /data/bert-base-cased --image face.jpg --text happiness anger --out demo_out
    python tools/demo.py --synthetic --out demo_out        # random-init weights, random image, fixed embedding:
                                                            # exercises the whole pipeline where no checkpoint exists
"""
import argparse
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvpr23_lfdm_amd import FlowDiffusion, io_compat as C  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "lfae_128.yaml"))
    ap.add_argument("--lfae-ckpt", default="", help="RegionMM_*.pth (keys generator / region_predictor / bg_predictor)")
    ap.add_argument("--dm-ckpt", default="", help="flowdiff_*.pth (key 'diffusion')")
    ap.add_argument("--bert", default=os.environ.get("LFDM_BERT_PATH"), help="local bert-base-cased directory")
    ap.add_argument("--image", default="")
    ap.add_argument("--text", nargs="+", default=["happiness"])
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--steps", type=int, default=100, help="DDIM steps (sampling_timesteps)")
    ap.add_argument("--cond-scale", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--out", default="demo_out")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        sys.exit("tools/demo.py needs a GPU: the sampling path is liblfdm_hip.so only")
    os.makedirs(args.out, exist_ok=True)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)

    model = FlowDiffusion(is_train=False, img_size=args.size // 4, num_frames=args.frames, sampling_timesteps=args.steps,
                          null_cond_prob=0.1, config_pth=args.config, pretrained_pth=args.lfae_ckpt,
                          bert_path=None if args.synthetic else args.bert)          # demo_mug.py:80-88
    if args.synthetic:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import synth
        model.unet.load_state_dict(synth.unet_state())
        model.generator.load_state_dict(synth.generator_state())
        emb = {t: torch.randn(1, 768, generator=torch.Generator().manual_seed(zlib.crc32(t.encode()))) for t in args.text}
        model.diffusion.text_encoder = lambda texts: torch.cat([emb[t] for t in texts])
    elif args.dm_ckpt:
        model.diffusion.load_state_dict(torch.load(args.dm_ckpt, map_location="cpu")["diffusion"])   # demo_mug.py:93-97
    else:
        sys.exit("give --dm-ckpt (and --lfae-ckpt), or --synthetic")
    model.cuda().eval()

    if args.image:
        img = C.resize(C.imread(args.image)[:, :, :3], args.size, interpolation=C.INTER_AREA)        # demo_mug.py:113-114
    else:
        img = np.random.default_rng(args.seed).integers(0, 256, size=(args.size, args.size, 3), dtype=np.uint8)
    ref = torch.from_numpy(np.asarray(img, np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0).cuda()
    name = os.path.splitext(os.path.basename(args.image))[0] if args.image else "random"
    for i, text in enumerate(args.text):
        model.set_sample_input(sample_img=ref, sample_text=[text])
        model.sample_one_video(cond_scale=args.cond_scale)
        path = os.path.join(args.out, "%04d_%s_%s_%.2f.gif" % (i, text.replace(" ", "_"), name, args.cond_scale))
        C.mimsave(path, C.video_strip(model, ref))
        print(path)


if __name__ == "__main__":
    main()
