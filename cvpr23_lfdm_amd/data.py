"""Video-folder dataset for the DM training / test drivers: the item format and the frame
sampling of the reference's dataset classes (DM/datasets_mug.py:54-114, datasets_mhad.py, datasets_natops.py) without their
dataset-specific directory conventions, cv2, imageio or torchvision.

    root/<label>/<video>/*.jpg|*.png          one directory of frames per video, the parent directory name is the text label

Item = (video (3, T, H, W) float32 = (frame - mean) / 255, label str, name str) - what the training loop unpacks as
`real_vids, ref_texts, real_names` (train_video_flow_diffusion_mug.py:218).  Host-side Python; not on the hot path.
"""
import os
import random

import numpy as np
import torch
from torch.utils import data

from .io_compat import INTER_AREA, imread, resize


def sample_indices(total, num_frames, sampling="uniform", rng=np.random):
    """datasets_mug.py:64-87: `uniform`, `random` (uniform grid jittered inside its cells), `very_random` (sorted draws with
    replacement, first frame 0); a video shorter than num_frames repeats its last frame."""
    if total >= num_frames:
        idx = np.linspace(start=0, stop=total - 1, num=num_frames, dtype=int)
        if sampling == "random":
            step = idx[1:] - idx[:-1]
            out = idx.copy()
            for i in range(1, num_frames - 1):
                out[i] = out[i] + rng.randint(low=1 - step[i - 1], high=step[i])
            idx = np.sort(out)
    else:
        idx = np.pad(list(range(total)), (0, num_frames - total), "edge")
    if sampling == "very_random":
        idx = np.sort(rng.choice(total, num_frames, replace=True))
        idx[0] = 0
    return idx


def color_jitter(frames, bright=64. / 255, contrast=0.25, sat=0.25, hue=0.04, rnd=random):
    """datasets_mug.py:93-105: ONE random brightness / contrast / saturation / hue change applied to all frames of a video
    (torchvision.transforms.functional's PIL implementations, written with PIL directly)."""
    from PIL import Image, ImageEnhance
    bf = rnd.uniform(max(0, 1 - bright), 1 + bright)
    cf = rnd.uniform(max(0, 1 - contrast), 1 + contrast)
    sf = rnd.uniform(max(0, 1 - sat), 1 + sat)
    hf = rnd.uniform(-hue, hue)
    out = []
    for arr in frames:
        im = Image.fromarray(arr)
        im = ImageEnhance.Brightness(im).enhance(bf)
        im = ImageEnhance.Contrast(im).enhance(cf)
        im = ImageEnhance.Color(im).enhance(sf)
        h, s, v = im.convert("HSV").split()
        h = Image.fromarray((np.asarray(h, np.uint8).astype(np.int16) + int(hf * 255)).astype(np.uint8))   # wraps like uint8
        out.append(np.asarray(Image.merge("HSV", (h, s, v)).convert("RGB")))
    return out


class FrameFolderVideos(data.Dataset):
    def __init__(self, root, image_size=128, num_frames=40, sampling="uniform", mean=(0, 0, 0), jitter=False):
        self.image_size, self.num_frames, self.sampling, self.jitter = image_size, num_frames, sampling, jitter
        self.mean = np.asarray(mean, np.float32)
        self.videos = []
        for label in sorted(os.listdir(root)):
            ldir = os.path.join(root, label)
            if not os.path.isdir(ldir):
                continue
            for vid in sorted(os.listdir(ldir)):
                vdir = os.path.join(ldir, vid)
                frames = sorted(f for f in os.listdir(vdir) if f.endswith(("jpg", "png"))) if os.path.isdir(vdir) else []
                if frames:
                    self.videos.append((label, vid, [os.path.join(vdir, f) for f in frames]))
        if not self.videos:
            raise FileNotFoundError("no <label>/<video>/*.jpg|png under %r" % (root,))

    def __len__(self):
        return len(self.videos)

    def __getitem__(self, index):
        label, vid, paths = self.videos[index]
        idx = sample_indices(len(paths), self.num_frames, self.sampling)
        frames = [_rgb(imread(paths[i])) for i in idx]
        if self.jitter:
            frames = color_jitter(frames)
        frames = [resize(np.asarray(f, np.float32), self.image_size, interpolation=INTER_AREA) - self.mean for f in frames]
        video = np.stack([np.transpose(f, (2, 0, 1)) for f in frames], axis=1)
        return np.array(video / 255.0, dtype=np.float32), label, "%s_%s" % (label, vid)


def _rgb(a):
    """(H,W) grayscale / (H,W,1) / (H,W,4) RGBA frames -> (H,W,3), as the datasets' cv2.imread(..., IMREAD_COLOR) yields."""
    a = np.asarray(a)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.shape[2] == 1:
        a = np.repeat(a, 3, axis=2)
    return a[:, :, :3]


class SyntheticVideos(data.Dataset):
    """Seeded random videos of the same item format (this image has no dataset): smooth random motion of a random image."""

    def __init__(self, n=64, image_size=128, num_frames=40, labels=("happiness", "anger", "surprise"), seed=0):
        self.n, self.image_size, self.num_frames, self.labels, self.seed = n, image_size, num_frames, labels, seed

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 100003 + index)
        s, t = self.image_size, self.num_frames
        base = torch.nn.functional.interpolate(torch.rand(1, 3, 8, 8, generator=g), size=(s, s), mode="bicubic", align_corners=False)
        shift = torch.cumsum(torch.randn(t, 2, generator=g) * 0.01, dim=0)
        ys, xs = torch.meshgrid(torch.linspace(-1, 1, s), torch.linspace(-1, 1, s), indexing="ij")
        grid = torch.stack((xs, ys), -1)[None] + shift.view(t, 1, 1, 2)
        vid = torch.nn.functional.grid_sample(base.expand(t, -1, -1, -1), grid, padding_mode="border", align_corners=True)
        return vid.clamp(0, 1).permute(1, 0, 2, 3).numpy().astype(np.float32), self.labels[index % len(self.labels)], "synthetic_%04d" % index
