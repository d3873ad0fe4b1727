"""Text conditioning for `cond=list[str]` (reference DM/modules/text.py:1-89, used by GaussianDiffusion.sample :767 and
.forward :864): BERT-base-cased token embeddings -> masked mean over the tokens after [CLS] (or the [CLS] vector).

The reference fetches tokenizer and weights with torch.hub at first use; this image (and the GPU boxes) have no network, so
the model is loaded from a LOCAL Hugging Face directory (config.json, vocab.txt, model.safetensors / pytorch_model.bin)
supplied offline, through the `transformers` package that is installed here.  BERT runs once per sample()/training batch
on (B, <= 512) tokens - caller-side conditioning, not part of the hot path; its (B, 768) output feeds the native path
exactly like a precomputed tensor would.

    enc = BertTextEncoder("/data/bert-base-cased")         # directory supplied offline
    model.diffusion.text_encoder = enc                     # or FlowDiffusion(..., bert_path="/data/bert-base-cased")
"""
import os

import torch

BERT_MODEL_DIM = 768


def _load(path):
    if not path or not os.path.isdir(path):
        raise FileNotFoundError(
            "BERT directory %r not found: the reference downloads 'bert-base-cased' with torch.hub, this image has no "
            "network - supply the Hugging Face files (config.json, vocab.txt, weights) offline and pass their directory" % (path,))
    try:
        from transformers import BertModel, BertTokenizer
    except ImportError as e:                                    # pragma: no cover
        raise RuntimeError("text conditioning needs the `transformers` package") from e
    tok = BertTokenizer.from_pretrained(path, local_files_only=True, do_lower_case=False)
    model = BertModel.from_pretrained(path, local_files_only=True).eval()
    return tok, model


class BertTextEncoder:
    """list[str] -> (B, hidden) tensor with the reference's pooling.  `use_cls` = text_use_bert_cls."""

    def __init__(self, path, use_cls=False, device=None):
        self.tokenizer, self.model = _load(path)
        self.use_cls = use_cls
        if device is not None:
            self.model = self.model.to(device)

    def tokenize(self, texts, add_special_tokens=True):
        """text.py:37-52: padded batch of token ids (pad id 0)."""
        if not isinstance(texts, (list, tuple)):
            texts = [texts]
        enc = self.tokenizer(list(texts), add_special_tokens=add_special_tokens, padding=True, return_tensors="pt")
        return enc.input_ids

    @torch.no_grad()
    def embed(self, token_ids, return_cls_repr=False, eps=1e-8, pad_id=0):
        """text.py:55-89."""
        dev = next(self.model.parameters()).device
        token_ids = token_ids.to(dev)
        mask = token_ids != pad_id
        hidden = self.model(input_ids=token_ids, attention_mask=mask, output_hidden_states=True).hidden_states[-1]
        if return_cls_repr:
            return hidden[:, 0]
        m = mask[:, 1:].unsqueeze(-1)                          # all tokens after [CLS], padding excluded
        numer = (hidden[:, 1:] * m).sum(dim=1)
        denom = m.sum(dim=1)
        return numer / (denom + eps)

    def __call__(self, texts):
        return self.embed(self.tokenize(texts), return_cls_repr=self.use_cls)
