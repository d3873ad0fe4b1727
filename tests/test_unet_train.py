"""UNet training forward + backward (cvpr23_lfdm_amd/unet_train.py: native kernels under torch.autograd) against
torch autograd through the CPU oracle (oracle/lfdm_oracle.py::unet_forward, the reference dataflow) on the same
synthetic weights: output and the gradient of EVERY parameter."""
import os

import pytest
import torch

import lfdm_oracle as O
import synth
from cvpr23_lfdm_amd import Unet3D
from cvpr23_lfdm_amd.unet_train import unet_train_forward
from util import assert_close


def _run(dev, b, t, s, learn_null=False, null_mask=None, use_deconv=True, padding_mode="zeros", focus=None):
    usd = synth.unet_state(learn_null_cond=learn_null, use_deconv=use_deconv)
    unet = Unet3D(dim=64, channels=259, out_grid_dim=2, out_conf_dim=1, use_bert_text_cond=True, learn_null_cond=learn_null,
                  use_deconv=use_deconv, padding_mode=padding_mode)
    unet.load_state_dict(usd)
    unet.to(dev).train()
    x, time, cond = synth.unet_inputs(b, t, s)
    dy = synth.NoiseTape(11)((b, 3, t, s, s))
    # oracle + torch autograd (reference dataflow)
    sd = {"denoise_fn." + k: v.clone().requires_grad_(v.is_floating_point() and "rotary" not in k) for k, v in usd.items()}
    ref = O.unet_forward(sd, x, time, cond, null_mask=null_mask, focus_mask=None if focus is None else torch.tensor(focus))
    ref.backward(dy)
    # native
    unet.zero_grad()
    prob = 0.0
    out = unet_train_forward(unet, x[:, :3].to(dev), x[:, 3:, 0].contiguous().to(dev), time.to(dev), cond.to(dev),
                             null_cond_prob=prob, none_cond_mask=null_mask, focus=focus)
    out.backward(dy.to(dev))
    assert_close(out, ref, 1e-3, "unet train forward")
    worst = ("", 0.0)
    names = dict(unet.named_parameters())
    for k, p in names.items():
        rg = sd["denoise_fn." + k].grad
        assert p.grad is not None, k
        assert rg is not None, k
        scale = float(rg.abs().max()) + 1e-12
        err = float((p.grad.cpu() - rg).abs().max()) / scale
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] < 2e-3, "largest relative gradient error %.3e at %s" % (worst[1], worst[0])


@pytest.mark.parametrize("case", ["plain", "null_cond", "upconv_reflect", "focus_mixed", "focus_all"])
def test_unet_train_grads(backend, case):
    dev = backend
    if dev == "cpu":
        if os.environ.get("LFDM_EMU_E2E", "0") != "1":
            pytest.skip("UNet forward+backward under the emulator is opt-in (LFDM_EMU_E2E=1); it runs on the GPU")
        if case != "plain":
            pytest.skip("one emulator case is enough")
        _run(dev, 1, 2, 8)
    elif case == "plain":
        _run(dev, 2, 4, 8)
    elif case == "null_cond":
        _run(dev, 2, 3, 8, learn_null=True, null_mask=torch.tensor([True, False]))
    elif case == "focus_mixed":      # focus_present_mask (Attention.forward :342-352): forward and every gradient
        _run(dev, 3, 3, 8, focus=[True, False, True])
    elif case == "focus_all":        # (:313-317: to_qkv's q / k rows get no gradient from the blocks' temporal attentions)
        _run(dev, 2, 3, 8, focus=[True, True])
    else:   # the NATOPS configuration: learned null cond, nearest-upsample + reflect-pad Upsample
        _run(dev, 1, 2, 8, learn_null=True, use_deconv=False, padding_mode="reflect")
