#!/usr/bin/env python
"""TEST INFRASTRUCTURE - text-conditioning fixture: runs the UNMODIFIED reference DM/modules/text.py (tokenize + bert_embed,
both poolings) on a tiny random-init BERT whose Hugging Face files are committed next to the fixture
(tests/golden/tiny_bert, ~60 KB; bert-base-cased itself cannot travel and there is no network).

    python oracle/make_golden_text.py        -> tests/golden/text_embed.npz + tests/golden/tiny_bert/

Separate from make_golden.py because the import shims that script installs for the rest of the reference (a stub
torchvision among them) break `transformers`; text.py needs none of them."""
import importlib.util
import os

import numpy as np
import torch
from transformers import BertConfig, BertModel, BertTokenizer

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")
TEXTS = ["happiness", "a person waves the right hand", "None", "Anger disgust, and Fear !"]
WORDS = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "a", "person", "waves", "the", "right", "hand", "happiness", "None",
         "Anger", "disgust", "and", "Fear", ",", "!", "##s", "wave", "happ", "##iness"]


def main():
    spec = importlib.util.spec_from_file_location("ref_text", "/root/reference/DM/modules/text.py")
    rtext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rtext)
    d = os.path.join(OUT, "tiny_bert")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "vocab.txt"), "w") as f:
        f.write("\n".join(WORDS) + "\n")
    torch.manual_seed(4242)
    cfg = BertConfig(vocab_size=len(WORDS), hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                     intermediate_size=64, max_position_embeddings=32)
    model = BertModel(cfg).eval()
    model.save_pretrained(d)
    tok = BertTokenizer(os.path.join(d, "vocab.txt"), do_lower_case=False)
    tok.save_pretrained(d)
    if not hasattr(tok, "batch_encode_plus"):            # transformers >= 5 dropped the alias text.py:44 calls; __call__ is
        tok.batch_encode_plus = lambda texts, **kw: tok(texts, **kw)   # its documented replacement (same arguments)
    rtext.MODEL, rtext.TOKENIZER = model, tok            # the singletons torch.hub would have filled (text.py:11-31)
    cuda = torch.cuda.is_available
    torch.cuda.is_available = lambda: False              # bert_embed moves ids to .cuda() when it sees one
    try:
        ids = rtext.tokenize(TEXTS)
        mean, cls = rtext.bert_embed(ids), rtext.bert_embed(ids, return_cls_repr=True)
    finally:
        torch.cuda.is_available = cuda
    path = os.path.join(OUT, "text_embed.npz")
    np.savez_compressed(path, token_ids=ids.numpy(), mean=mean.numpy(), cls=cls.numpy(), texts=np.array(TEXTS))
    print("wrote", path, ids.shape, mean.shape)


if __name__ == "__main__":
    main()
