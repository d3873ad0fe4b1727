"""cvpr23_lfdm_amd/datasets.py (MUG / MHAD / NATOPS with the reference's directory conventions, splits and item pipeline)
against the reference's own dataset classes on synthetic trees - run in a fresh interpreter where the reference's
DM/datasets_*.py can be imported with the repository's imageio / cv2 stand-ins (those packages are not installed)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from cvpr23_lfdm_amd import datasets as DS
from cvpr23_lfdm_amd import io_compat as C

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("LFDM_REFERENCE_ROOT", "/root/reference")


def _frames(d, n, hw=(60, 80), seed=0):
    os.makedirs(d)
    g = np.random.default_rng(seed)
    for i in range(n):
        C.imsave(os.path.join(d, "img_%04d.png" % i), g.integers(0, 256, size=hw + (3,), dtype=np.uint8))


def _trees(tmp):
    mug, mhad, nat = (os.path.join(tmp, n) for n in ("MUG", "MHAD", "NATOPS"))
    _frames(os.path.join(mug, "008", "anger", "take000"), 12, seed=1)           # train subject
    _frames(os.path.join(mug, "017", "surprise", "take001"), 3, seed=2)         # shorter than num_frames
    _frames(os.path.join(mug, "001", "fear", "take000"), 7, seed=3)             # test subject
    _frames(os.path.join(mug, "049", "session0", "neutral", "take002"), 6, seed=4)   # test subject with a session level (listed twice)
    _frames(os.path.join(mhad, "a3_s1_t2_color"), 9, seed=5)                    # train
    _frames(os.path.join(mhad, "a27_s6_t1_color"), 8, seed=6)                   # test
    _frames(os.path.join(nat, "g05s03r01"), 10, hw=(240, 320), seed=7)          # train (crop box 10:239, 30:290)
    _frames(os.path.join(nat, "g24s02r02"), 8, hw=(240, 320), seed=8)           # test
    return mug, mhad, nat


def test_catalogues_and_items(tmp_path):
    mug, mhad, nat = _trees(str(tmp_path))
    ds = DS.MUG(mug, num_frames=6, image_size=32, color_jitter=False, sampling="uniform")
    assert [os.path.relpath(p, mug) for p in ds.video_path_list] == ["008/anger/take000", "017/surprise/take001"]
    vid, label, name = ds[0]
    assert vid.shape == (3, 6, 32, 32) and vid.dtype == np.float32 and label == "anger" and name == "008_anger_take000"
    assert ds[1][1] == "surprise" and np.array_equal(ds[1][0][:, 2], ds[1][0][:, 5])       # 3 frames -> the last one repeats
    t = DS.MUG_test(mug, num_frames=4, image_size=32)
    assert [n for _, _, n in (t[i] for i in range(len(t)))] == ["001_fear_take000", "049_session0_neutral_take002"] * 1 + ["049_session0_neutral_take002"]
    m = DS.MHAD(mhad, num_frames=5, image_size=32, color_jitter=False, sampling="uniform")
    assert len(m) == 1 and m[0][1] == "right hand wave" and m[0][2] == "a3_s1_t2_color"
    assert DS.MHAD_test(mhad, num_frames=5, image_size=32)[0][1] == "squat"
    assert len(DS.MHAD(mhad, num_frames=5, image_size=32, split_train_test=False)) == 2
    n = DS.NATOPS(nat, num_frames=5, image_size=32, color_jitter=False, sampling="uniform")
    assert len(n) == 1 and n[0][1] == "Fold Wings" and n[0][0].shape == (3, 5, 32, 32)
    assert DS.NATOPS_test(nat, num_frames=5, image_size=32)[0][1] == "Cut Engine"
    v, _, _ = DS.NATOPS(nat, num_frames=5, image_size=64, color_jitter=True, sampling="very_random")[0]   # jitter + random sampling run
    assert v.shape == (3, 5, 64, 64) and np.isfinite(v).all()


WORKER = r'''
import importlib.util, json, os, sys
import numpy as np
repo, ref, tmp = sys.argv[1:4]
sys.path[:] = [os.path.join(repo, "oracle", "ref_shims")] + [p for p in sys.path if os.path.abspath(p or ".") != repo] + [repo]
def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path); mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod; spec.loader.exec_module(mod); return mod
for name in ("imageio", "cv2"):                      # working stand-ins instead of the import-only stubs
    load(name, os.path.join(repo, "cvpr23_lfdm_amd", "standins", name + ".py"))
load("misc", os.path.join(repo, "misc.py"))
import torchvision.transforms as T
sys.modules.setdefault("torchvision.transforms.functional", T)        # only imported; colour jitter is off in this comparison
T.functional = T
from cvpr23_lfdm_amd import datasets as DS
out = {}
cases = [("mug", "MUG", dict(color_jitter=False, sampling="random")), ("mug", "MUG", dict(color_jitter=False, sampling="very_random")),
         ("mug", "MUG_test", {}), ("mhad", "MHAD", dict(color_jitter=False, sampling="random")), ("mhad", "MHAD_test", {}),
         ("natops", "NATOPS", dict(color_jitter=False)), ("natops", "NATOPS_test", {})]
worst = 0.0
for ds_name, cls, kw in cases:
    rmod = load("ref_datasets_" + ds_name, os.path.join(ref, "DM", "datasets_%s.py" % ds_name))
    root = os.path.join(tmp, {"mug": "MUG", "mhad": "MHAD", "natops": "NATOPS"}[ds_name])
    a, b = getattr(DS, cls)(root, num_frames=6, image_size=48, **kw), getattr(rmod, cls)(root, num_frames=6, image_size=48, **kw)
    assert len(a) == len(b) > 0 and a.video_path_list == b.video_path_list, (cls, a.video_path_list, b.video_path_list)
    for i in range(len(a)):
        np.random.seed(7 + i); va, la, na = a[i]
        np.random.seed(7 + i); vb, lb, nb = b[i]
        assert la == lb and na == nb and va.shape == vb.shape and va.dtype == vb.dtype, (cls, i, la, lb, na, nb)
        worst = max(worst, float(np.abs(va - vb).max()))
print(json.dumps({"worst": worst, "cases": len(cases)}))
'''


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "DM", "datasets_mug.py")), reason="reference tree not present")
def test_items_match_the_reference_classes(tmp_path):
    _trees(str(tmp_path))
    r = subprocess.run([sys.executable, "-c", WORKER, REPO, REF, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, env=dict(os.environ, PYTHONPATH="", MPLBACKEND="Agg"), cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["cases"] == 7 and out["worst"] == 0.0, out          # same files, same indices, same resize: bit-identical items
