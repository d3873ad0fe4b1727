"""Caller-side helpers (cvpr23_lfdm_amd/io_compat.py = the reference's misc.py vocabulary + image/GIF I/O without cv2 /
imageio / flow_vis).  Semantics are checked against their definitions: misc.py:44-173, flow_vis' Middlebury wheel."""
import numpy as np
import torch

from cvpr23_lfdm_amd import io_compat as C


def test_resize_keeps_aspect_and_pads():
    g = np.random.default_rng(0)
    im = g.integers(0, 256, size=(256, 128, 3), dtype=np.uint8)
    out = C.resize(im, 128, interpolation=C.INTER_AREA)                 # misc.py:96-110
    assert out.shape == (128, 128, 3) and out.dtype == np.uint8
    assert (out[:, :32] == 0).all() and (out[:, 96:] == 0).all()        # 64 columns of image centred in 128
    area = im.reshape(128, 2, 64, 2, 3).astype(np.float64).mean(axis=(1, 3))     # INTER_AREA at an integer factor
    assert np.abs(out[:, 32:96].astype(np.float64) - area).max() <= 0.5 + 1e-9
    assert C.resize(im[..., 0], 64).shape == (64, 64)


def test_conf_and_grid_figures():
    conf = torch.linspace(0, 1, 32 * 32).view(1, 32, 32)
    c = C.conf2fig(conf, img_size=128)                                  # misc.py:76-80
    assert c.shape == (128, 128) and c.dtype == np.uint8 and c[0, 0] == 0 and c[-1, -1] == 255
    r = torch.linspace(-1, 1, 32)
    ident = torch.stack(torch.meshgrid([r, r], indexing="ij"), -1).flip(2).numpy()
    fig = C.grid2fig(ident * 0.9, grid_size=32, img_size=128)           # misc.py:44-63
    assert fig.shape == (128, 128, 3) and fig.dtype == np.uint8
    assert (fig != 255).any() and (fig == 255).any()                    # lines on a white canvas


def test_flow_colour_wheel():
    wheel = C._color_wheel()
    assert wheel.shape == (55, 3) and wheel[0].tolist() == [255, 0, 0] and wheel[15].tolist() == [255, 255, 0]
    flow = np.zeros((4, 4, 2))
    flow[0, 0] = (1, 0)       # +u -> red
    flow[1, 1] = (-1, 0)      # -u -> cyan side
    img = C.flow_to_color(flow)
    assert img[2, 2].tolist() == [255, 255, 255]                        # zero flow is white
    assert img[0, 0, 0] == 255 and img[0, 0, 1] < 10 and img[0, 0, 2] < 10
    assert img[1, 1, 0] < 10 and img[1, 1, 1] > 200 and img[1, 1, 2] > 200
    fig = C.flow2fig(np.random.default_rng(1).normal(size=(32, 32, 2)), np.zeros((32, 32, 2)), img_size=16)
    assert fig.shape == (16, 16, 3)


def test_gif_and_png_round_trip(tmp_path):
    from PIL import Image
    frames = [np.full((16, 80, 3), i * 20, np.uint8) for i in range(5)]
    p = str(tmp_path / "v.gif")
    C.mimsave(p, frames)
    with Image.open(p) as im:
        assert im.n_frames == 5 and im.size == (80, 16)
    q = str(tmp_path / "f.png")
    C.imsave(q, frames[3])
    assert (C.imread(q) == frames[3]).all()


def test_video_strip_layout():
    class M:
        pass
    m, s, t = M(), 32, 3
    m.sample_out_vid = torch.rand(1, 3, t, s, s)
    m.sample_warped_vid = torch.rand(1, 3, t, s, s)
    m.sample_vid_grid = torch.rand(1, 2, t, 8, 8) * 2 - 1
    m.sample_vid_conf = torch.rand(1, 1, t, 8, 8)
    ref = torch.rand(1, 3, s, s)
    frames = C.video_strip(m, ref, grid_size=8)                         # demo_mug.py:124-143
    assert len(frames) == t and frames[0].shape == (s, 5 * s, 3) and frames[0].dtype == np.uint8
    assert (frames[1][:, :s] == C.sample_img(ref)).all()
    assert (frames[2][:, s:2 * s] == C.sample_img(m.sample_out_vid[:, :, 2])).all()


def test_resample_identity_and_logger(tmp_path, capsys):
    x = torch.rand(2, 3, 8, 8)
    assert torch.allclose(C.resample(x, torch.zeros(2, 2, 8, 8)), x, atol=1e-6)       # misc.py:113-134
    g = C.get_grid(2, (4, 6), device="cpu")
    assert g.shape == (2, 2, 4, 6) and float(g[0, 0, 0, 0]) == -1 and float(g[0, 1, -1, 0]) == 1
    import sys
    log = C.Logger(str(tmp_path / "log.txt"), sys.stdout)
    log.write("hello\n")
    log.flush()
    assert "hello" in open(str(tmp_path / "log.txt")).read() and "hello" in capsys.readouterr().out
