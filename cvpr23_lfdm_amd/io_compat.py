"""Caller-side helpers of the reference's demo / test scripts without the packages this image
lacks (cv2, imageio, flow_vis): the functions of the reference's `misc.py` that demo_*.py / test_video_flow_diffusion_*.py
call around `FlowDiffusion.sample_one_video`, plus image / GIF I/O.  Same names, argument meaning and return types
(numpy uint8 HxWxC images), built on numpy / torch / PIL / matplotlib, which ARE here.  Nothing in this file is on the hot
path; it exists so a user of the reference finds the surrounding script vocabulary when switching over (tools/demo.py).

  misc.py:44 grid2fig    misc.py:66 flow2fig    misc.py:76 conf2fig    misc.py:83 Logger
  misc.py:96 resize      misc.py:113 resample   misc.py:137 get_grid
  imageio.v2.imread / imageio.mimsave / imageio.imsave  -> imread / mimsave / imsave
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA = 0, 1, 2, 3          # cv2's constants, accepted by resize()


class Logger(object):
    """misc.py:83-93: tee a stream into a log file (`sys.stdout = Logger(path, sys.stdout)`)."""

    def __init__(self, filename="default.log", stream=sys.stdout):
        self.terminal = stream
        # one process per GPU runs the whole unchanged script: only rank 0 owns the script's log file, the others write theirs next to it
        rank = int(os.environ.get("RANK", "0"))
        self.log = open(filename if rank == 0 else "%s.rank%d" % (filename, rank), "w")

    def write(self, message):
        self.terminal.write(message)
        self.log.write(message)

    def flush(self):
        self.terminal.flush()
        self.log.flush()


def _resize_hw(im, h, w, interpolation):
    """(H, W[, C]) uint8/float array -> (h, w[, C]), cv2.resize semantics for the modes the scripts use."""
    arr = np.asarray(im)
    squeeze = arr.ndim == 2
    t = torch.from_numpy(np.ascontiguousarray(arr if not squeeze else arr[..., None])).permute(2, 0, 1)[None].float()
    if interpolation == INTER_AREA and h <= t.shape[2] and w <= t.shape[3]:
        out = F.adaptive_avg_pool2d(t, (h, w))               # area averaging (exact for integer shrink factors)
    elif interpolation == INTER_NEAREST:
        out = F.interpolate(t, size=(h, w), mode="nearest")
    elif interpolation == INTER_CUBIC:
        out = F.interpolate(t, size=(h, w), mode="bicubic", align_corners=False)
    else:
        out = F.interpolate(t, size=(h, w), mode="bilinear", align_corners=False)
    out = out[0].permute(1, 2, 0).numpy()
    if arr.dtype == np.uint8:
        out = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out[..., 0] if squeeze else out


def resize(im, desired_size, interpolation=INTER_AREA):
    """misc.py:96-110: scale the longer side to `desired_size` keeping the aspect ratio, zero-pad to a square."""
    old = im.shape[:2]
    ratio = float(desired_size) / max(old)
    new = tuple(int(x * ratio) for x in old)
    im = _resize_hw(im, new[0], new[1], interpolation)
    dh, dw = desired_size - new[0], desired_size - new[1]
    pad = [(dh // 2, dh - dh // 2), (dw // 2, dw - dw // 2)] + [(0, 0)] * (im.ndim - 2)
    return np.pad(im, pad, mode="constant")


def conf2fig(conf, img_size=128):
    """misc.py:76-80: (1, h, w) occlusion map in [0,1] -> (img_size, img_size) uint8."""
    conf = F.interpolate(conf.unsqueeze(dim=0).float(), size=img_size).data.cpu().numpy()
    return np.array(np.transpose(conf, [0, 2, 3, 1])[0, :, :, 0] * 255, dtype=np.uint8)


def grid2fig(warped_grid, grid_size=32, img_size=256):
    """misc.py:44-63: the deformed sampling grid drawn over the identity grid -> (img_size, img_size, 3) uint8."""
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    from matplotlib.collections import LineCollection

    def plot_grid(x, y, ax, **kw):
        s1 = np.stack((x, y), axis=2)
        ax.add_collection(LineCollection(s1, **kw))
        ax.add_collection(LineCollection(s1.transpose(1, 0, 2), **kw))
        ax.autoscale()

    r = torch.linspace(-1, 1, grid_size)
    ident = torch.stack(torch.meshgrid([r, r], indexing="ij"), -1).flip(2).numpy()
    fig, ax = plt.subplots()
    plot_grid(ident[..., 0], ident[..., 1], ax, color="lightgrey")
    plot_grid(np.asarray(warped_grid)[..., 0], np.asarray(warped_grid)[..., 1], ax, color="C0")
    plt.axis("off")
    plt.tight_layout(pad=0)
    fig.set_size_inches(img_size / 100, img_size / 100)
    fig.set_dpi(100)
    fig.canvas.draw()
    out = np.asarray(fig.canvas.buffer_rgba())[:, :, :3].copy()
    plt.close(fig)
    return out


def _color_wheel():
    """The Middlebury optical-flow colour wheel (Baker et al., IJCV 2011) that flow_vis.flow_to_color uses: 55 hues."""
    ry, yg, gc, cb, bm, mr = 15, 6, 4, 11, 13, 6
    wheel = np.zeros((ry + yg + gc + cb + bm + mr, 3))
    col = 0
    for n, (fixed, ramp, up) in zip((ry, yg, gc, cb, bm, mr),
                                    ((0, 1, True), (1, 0, False), (1, 2, True), (2, 1, False), (2, 0, True), (0, 2, False))):
        wheel[col:col + n, fixed] = 255
        steps = np.floor(255 * np.arange(n) / n)
        wheel[col:col + n, ramp] = steps if up else 255 - steps
        col += n
    return wheel


def flow_to_color(flow_uv, clip_flow=None):
    """flow_vis.flow_to_color: (H, W, 2) flow -> (H, W, 3) uint8; hue = direction, saturation = magnitude / max magnitude."""
    flow_uv = np.asarray(flow_uv, dtype=np.float64)
    if clip_flow is not None:
        flow_uv = np.clip(flow_uv, 0, clip_flow)
    u, v = flow_uv[..., 0], flow_uv[..., 1]
    rad_max = np.sqrt(u * u + v * v).max()
    u, v = u / (rad_max + 1e-5), v / (rad_max + 1e-5)
    wheel = _color_wheel()
    ncols = wheel.shape[0]
    rad = np.sqrt(u * u + v * v)
    fk = (np.arctan2(-v, -u) / np.pi + 1) / 2 * (ncols - 1)
    k0 = np.floor(fk).astype(np.int32)
    k1 = np.where(k0 + 1 == ncols, 0, k0 + 1)
    f = fk - k0
    img = np.zeros(u.shape + (3,), np.uint8)
    for i in range(3):
        col = (1 - f) * wheel[k0, i] / 255.0 + f * wheel[k1, i] / 255.0
        col = np.where(rad <= 1, 1 - rad * (1 - col), col * 0.75)
        img[..., i] = np.floor(255 * col)
    return img


def flow2fig(warped_grid, id_grid, grid_size=32, img_size=128):
    """misc.py:66-73: colour-coded (warped grid - identity grid), resized to img_size."""
    img = flow_to_color(np.asarray(warped_grid) - np.asarray(id_grid))
    return _resize_hw(img, img_size, img_size, INTER_AREA if img_size <= img.shape[0] else INTER_LINEAR)


def get_grid(batchsize, size, minval=-1.0, maxval=1.0, device=None):
    """misc.py:137-173 for 2-D sizes: (B, 2, H, W) grid of (x, y) in [minval, maxval]; `device` defaults to the GPU if any."""
    rows, cols = size
    x = torch.linspace(minval, maxval, cols).view(1, 1, 1, cols).expand(batchsize, 1, rows, cols)
    y = torch.linspace(minval, maxval, rows).view(1, 1, rows, 1).expand(batchsize, 1, rows, cols)
    dev = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
    return torch.cat([x, y], dim=1).to(dev)


def resample(image, flow):
    """misc.py:113-134: bilinear warp of (N,C,H,W) by a pixel-unit flow (N,2,H,W), border padding, align_corners."""
    assert flow.shape[1] == 2
    b, c, h, w = image.size()
    grid = get_grid(b, (h, w), device=image.device)
    flow = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], dim=1)
    return F.grid_sample(image, (grid + flow).permute(0, 2, 3, 1), mode="bilinear", padding_mode="border", align_corners=True)


# ---------------------------------------------------------------------------------------------
# image / GIF files (imageio.v2.imread, imageio.imsave, imageio.mimsave in the scripts)
# ---------------------------------------------------------------------------------------------
def imread(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB") if im.mode not in ("RGB", "L") else im).copy()


def imsave(path, arr):
    from PIL import Image
    Image.fromarray(np.asarray(arr)).save(path)


def mimsave(path, frames, duration=0.1, loop=0):
    """Animated GIF from a list of (H, W[, 3]) uint8 frames (imageio.mimsave's default 10 fps)."""
    from PIL import Image
    ims = [Image.fromarray(np.asarray(f)) for f in frames]
    if not ims:
        raise ValueError("mimsave: no frames")
    ims[0].save(path, save_all=True, append_images=ims[1:], duration=int(round(duration * 1000)), loop=loop)


def sample_img(rec_img_batch, index=0, mean=(0.0, 0.0, 0.0)):
    """demo_mug.py:26-32: (B,3,H,W) model-range image -> (H,W,3) uint8 (adds the dataset mean back, scales by 255)."""
    rec = rec_img_batch[index].permute(1, 2, 0).data.cpu().numpy().copy()
    rec += np.array(mean) / 255.0
    rec[rec < 0], rec[rec > 1] = 0, 1
    return np.array(rec * 255, dtype=np.uint8)


def video_strip(model, ref_imgs, mean=(0.0, 0.0, 0.0), grid_size=32):
    """The per-frame panel demo_*.py assembles (demo_mug.py:124-143): [source | generated | warped | flow grid | occlusion]
    for every frame of model.sample_* (batch element 0) -> list of (S, 5S, 3) uint8 arrays ready for mimsave."""
    s = ref_imgs.shape[-1]
    src = sample_img(ref_imgs, 0, mean)
    frames = []
    for t in range(model.sample_out_vid.shape[2]):
        panel = np.zeros((s, 5 * s, 3), np.uint8)
        panel[:, 0:s] = src
        panel[:, s:2 * s] = sample_img(model.sample_out_vid[:, :, t], 0, mean)
        panel[:, 2 * s:3 * s] = sample_img(model.sample_warped_vid[:, :, t], 0, mean)
        panel[:, 3 * s:4 * s] = grid2fig(model.sample_vid_grid[0, :, t].permute(1, 2, 0).data.cpu().numpy(),
                                         grid_size=grid_size, img_size=s)
        panel[:, 4 * s:5 * s] = conf2fig(model.sample_vid_conf[0, :, t], img_size=s)[..., None]
        frames.append(panel)
    return frames
