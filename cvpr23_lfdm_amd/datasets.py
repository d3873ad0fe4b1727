"""The reference's dataset classes for the DM training / test drivers, with ITS directory conventions and train / test
subject splits (DM/datasets_mug.py:14-200, DM/datasets_mhad.py:16-232, DM/datasets_natops.py:16-232), so that
`from datasets_mug import MUG` / `from DM.datasets_mhad import MHAD` keep working (the repo's DM/datasets_*.py re-export
these).  One shared item pipeline (`_FrameVideos.__getitem__`) instead of six copies; what differs per dataset is only the
catalogue: which directories are videos, how the text label and the video name are derived, the crop box.

    MUG      <data_dir>/<subject>/[session0/]<expression>/<take>/*.jpg|png      label = expression
    MHAD     <data_dir>/a<action>_s<subject>_t<trial>_color/*                   label = action_list[action - 1]
    NATOPS   <data_dir>/g<action:02d>s<subject:02d>.../*                        label = action_list[action - 1], crop box

Item = (video (3, T, H, W) float32 = (frame - mean) / 255, label str, video name str).
cv2 / imageio / torchvision are not needed (io_compat + PIL).  The `*_gen` / `*_select` classes of the reference (fixed
sample lists for its evaluation scripts) are not rebuilt.  Host-side Python; nothing here is on the hot path.
"""
import os

import numpy as np
from torch.utils import data

from .data import _rgb, color_jitter, sample_indices
from .io_compat import INTER_AREA, imread, resize

MUG_EXPRESSIONS = ['anger', 'disgust', 'fear', 'happiness', 'neutral', 'sadness', 'surprise']
MUG_TRAIN_SUBJECTS = ['008', '017', '021', '028', '030', '031', '034', '036', '037', '038', '039', '042', '043', '044', '045',
                      '055', '060', '061', '062', '063', '071', '075', '076', '077', '083', '084']
MUG_TEST_SUBJECTS = ['001', '002', '006', '007', '010', '013', '014', '020', '027', '032', '033', '040', '046', '048', '049',
                     '052', '064', '065', '066', '070', '072', '073', '074', '078', '079', '082']
MUG_SESSION_SUBJECTS = ["002", "003", "049"]          # their takes sit one level deeper, under session0/

MHAD_ACTIONS = ["right arm swipe to the left", "right arm swipe to the right", "right hand wave", "two hand front clap",
                "right arm throw", "cross arms in the chest", "basketball shooting", "draw x", "draw circle clockwise",
                "draw circle counter clockwise", "draw triangle", "right hand bowling", "front boxing",
                "baseball swing from right", "tennis forehand swing", "two arms curl", "tennis serve", "two hand push",
                "knock on door", "hand catch", "pick up and throw", "jogging", "walking", "sit to stand", "stand to sit",
                "forward lunge (left foot forward)", "squat"]
MHAD_TRAIN_SUBJECTS, MHAD_TEST_SUBJECTS = [1, 5, 2, 3], [6, 8, 4, 7]

NATOPS_ACTIONS = ["I Have Command", "All Clear", "Not Clear", "Spread Wings", "Fold Wings", "Lock Wings", "Up Hook",
                  "Down Hook", "Remove Tiedowns", "Remove Chocks", "Insert Chocks", "Move Ahead", "Turn Left", "Turn Right",
                  "Next Marshaller", "Slow Down", "Stop", "Nosegear Steering", "Hot Brakes", "Brakes On", "Brakes Off",
                  "Install Tiedowns", "Fire", "Cut Engine"]
NATOPS_TRAIN_SUBJECTS, NATOPS_TEST_SUBJECTS = [3, 4, 8, 9, 12, 13, 15, 17, 19, 20], [2, 5, 6, 7, 10, 11, 14, 16, 18]
NATOPS_CROP = (10, 239, 30, 290)                       # y_min, y_max, x_min, x_max


class _FrameVideos(data.Dataset):
    """Shared item pipeline: list frames -> sample T indices -> read -> [crop] -> [one colour jitter per video] ->
    aspect-preserving resize + pad to a square -> subtract mean -> (3, T, H, W) / 255."""
    only_images = False        # MUG filters *.jpg|png; MHAD / NATOPS take every directory entry
    crop = None

    def __init__(self, num_frames, image_size, mean, color_jitter_, sampling):
        super().__init__()
        self.num_frames, self.image_size, self.mean = num_frames, image_size, mean
        self.is_jitter, self.sampling = color_jitter_, sampling
        self.video_path_list = []

    def __len__(self):
        return len(self.video_path_list)

    def describe(self, video_path):
        """-> (label, video name)"""
        raise NotImplementedError

    def __getitem__(self, index):
        video_path = self.video_path_list[index]
        label, name = self.describe(video_path)
        frames = sorted(os.listdir(video_path))
        if self.only_images:
            frames = [f for f in frames if f.endswith("jpg") or f.endswith("png")]
        idx = sample_indices(len(frames), self.num_frames, self.sampling)
        imgs = [_rgb(imread(os.path.join(video_path, frames[i]))) for i in idx]
        if self.crop is not None:
            y0, y1, x0, x1 = self.crop
            imgs = [im[y0:y1, x0:x1, :] for im in imgs]
        if self.is_jitter:
            imgs = color_jitter(imgs)
        imgs = [resize(np.asarray(im, np.float32), self.image_size, interpolation=INTER_AREA) - self.mean for im in imgs]
        video = np.stack([np.transpose(im, (2, 0, 1)) for im in imgs], axis=1)
        return np.array(video / 255.0, dtype=np.float32), label, name


class _MUGBase(_FrameVideos):
    only_images = True

    def _scan(self, data_dir, subjects):
        for subject in subjects:
            # (the reference walks "session0" twice for the session subjects - datasets_mug.py:42 - so their takes appear twice)
            prefixes = [()] if subject not in MUG_SESSION_SUBJECTS else [("session0",), ("session0",)]
            for prefix in prefixes:
                for exp in MUG_EXPRESSIONS:
                    d = os.path.join(data_dir, subject, *prefix, exp)
                    if os.path.exists(d):
                        self.video_path_list += [os.path.join(d, take) for take in sorted(os.listdir(d))]

    def describe(self, video_path):
        parts = video_path.split("/")
        name = "_".join(parts[-3:] if "session" not in video_path else parts[-4:])
        label = name.split("_")[-2]
        assert label in MUG_EXPRESSIONS, video_path
        return label, name


class MUG(_MUGBase):
    """DM/datasets_mug.py:14-114 (training subjects)."""

    def __init__(self, data_dir, num_frames=40, image_size=128, mean=(128, 128, 128), color_jitter=True, sampling="random"):
        super().__init__(num_frames, image_size, mean, color_jitter, sampling)
        self.exp_list = MUG_EXPRESSIONS
        self._scan(data_dir, MUG_TRAIN_SUBJECTS)


class MUG_test(_MUGBase):
    """DM/datasets_mug.py:117-200 (held-out subjects, uniform frame sampling)."""

    def __init__(self, data_dir, num_frames=16, image_size=256, mean=(128, 128, 128), color_jitter=False):
        super().__init__(num_frames, image_size, mean, color_jitter, "uniform")
        self.exp_list = MUG_EXPRESSIONS
        self._scan(data_dir, MUG_TEST_SUBJECTS)


class _MHADBase(_FrameVideos):
    def _scan(self, data_dir, subjects):
        names = sorted(os.listdir(data_dir))
        if subjects is not None:
            names = [n for n in names if int(n.split("_")[1][1:]) in subjects]
        self.video_path_list = [os.path.join(data_dir, n) for n in names]

    def describe(self, video_path):
        name = os.path.basename(video_path)
        return MHAD_ACTIONS[int(name.split("_")[0][1:]) - 1], name


class MHAD(_MHADBase):
    """DM/datasets_mhad.py:16-132."""

    def __init__(self, data_dir, num_frames=40, image_size=128, transform=None, mean=(0, 0, 0), color_jitter=True,
                 split_train_test=True, sampling="random"):
        super().__init__(num_frames, image_size, mean, color_jitter, sampling)
        self.action_list = MHAD_ACTIONS
        self._scan(data_dir, MHAD_TRAIN_SUBJECTS if split_train_test else None)


class MHAD_test(_MHADBase):
    """DM/datasets_mhad.py:135-232."""

    def __init__(self, data_dir, num_frames=40, image_size=256, mean=(0, 0, 0), color_jitter=False, split_train_test=True):
        super().__init__(num_frames, image_size, mean, color_jitter, "uniform")
        self.action_list = MHAD_ACTIONS
        self._scan(data_dir, MHAD_TEST_SUBJECTS if split_train_test else None)


class _NATOPSBase(_FrameVideos):
    def _scan(self, data_dir, subjects, use_crop):
        self.use_crop = use_crop
        if use_crop:
            self.crop = NATOPS_CROP
            self.y_min, self.y_max, self.x_min, self.x_max = NATOPS_CROP
            print("use crop box:", *NATOPS_CROP)
        names = [n for n in sorted(os.listdir(data_dir)) if int(n[4:6]) in subjects]
        self.video_path_list = [os.path.join(data_dir, n) for n in names]

    def describe(self, video_path):
        name = os.path.basename(video_path)
        return NATOPS_ACTIONS[int(name[1:3]) - 1], name


class NATOPS(_NATOPSBase):
    """DM/datasets_natops.py:16-132."""

    def __init__(self, data_dir, num_frames=40, image_size=128, mean=(0, 0, 0), color_jitter=True, use_crop=True,
                 sampling="very_random"):
        super().__init__(num_frames, image_size, mean, color_jitter, sampling)
        self.action_list = NATOPS_ACTIONS
        self._scan(data_dir, NATOPS_TRAIN_SUBJECTS, use_crop)


class NATOPS_test(_NATOPSBase):
    """DM/datasets_natops.py:136-232."""

    def __init__(self, data_dir, num_frames=40, image_size=256, mean=(0, 0, 0), color_jitter=False, use_crop=True):
        super().__init__(num_frames, image_size, mean, color_jitter, "uniform")
        self.action_list = NATOPS_ACTIONS
        self._scan(data_dir, NATOPS_TEST_SUBJECTS, use_crop)
