"""Text conditioning (cvpr23_lfdm_amd/text.py) against the fixture recorded from the UNMODIFIED reference
DM/modules/text.py (oracle/make_golden_text.py: tokenize + bert_embed on the tiny BERT in tests/golden/tiny_bert)."""
import os

import numpy as np
import pytest
import torch

import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "text_embed.npz")
BERT = os.path.join(HERE, "golden", "tiny_bert")


@pytest.fixture(autouse=True)
def _no_reference_import_shims(monkeypatch):
    """tests/test_oracle_vs_reference.py (this container only) imports the reference through oracle/ref_shims, whose stub
    `torchvision` makes `transformers` believe torchvision is installed.  Hide the shims while these tests run."""
    shims = os.path.join(os.path.dirname(HERE), "oracle", "ref_shims")
    monkeypatch.setattr(sys, "path", [p for p in sys.path if os.path.abspath(p) != shims])
    for name, mod in list(sys.modules.items()):
        if os.path.abspath(getattr(mod, "__file__", None) or "").startswith(shims):
            monkeypatch.delitem(sys.modules, name)


def test_tokenize_and_pooling_match_reference():
    from cvpr23_lfdm_amd.text import BertTextEncoder
    g = np.load(GOLD)
    texts = [str(t) for t in g["texts"]]
    enc = BertTextEncoder(BERT)
    ids = enc.tokenize(texts)
    assert ids.tolist() == g["token_ids"].tolist()
    assert torch.allclose(enc(texts), torch.from_numpy(g["mean"]), atol=1e-6), "masked mean over the tokens after [CLS]"
    assert torch.allclose(BertTextEncoder(BERT, use_cls=True)(texts), torch.from_numpy(g["cls"]), atol=1e-6)
    assert torch.allclose(enc(texts[1]), torch.from_numpy(g["mean"][1:2]), atol=1e-5), "a single string is a batch of one"


def test_missing_weights_fail_loudly(tmp_path):
    from cvpr23_lfdm_amd.text import BertTextEncoder
    with pytest.raises(FileNotFoundError, match="no\\s+network"):
        BertTextEncoder(str(tmp_path / "nope"))


def test_flow_diffusion_picks_up_bert_path(monkeypatch):
    """FlowDiffusion(bert_path=...) / LFDM_BERT_PATH install the encoder the reference gets from torch.hub."""
    import synth
    from cvpr23_lfdm_amd import FlowDiffusion
    monkeypatch.setenv("LFDM_BERT_PATH", BERT)
    m = FlowDiffusion(img_size=8, num_frames=2, sampling_timesteps=2, is_train=False, config_pth=synth.CONFIG, pretrained_pth="")
    out = m.diffusion.text_encoder(["happiness", "None"])
    assert out.shape == (2, 32)                    # the tiny fixture model; bert-base-cased gives (B, 768)
