"""Parameter trees that reproduce the reference's state-dict keys without
reproducing its module classes: a `ParamTree` registers tensors under dotted keys
('downs.0.2.fn.fn.to_qkv.weight'), so `state_dict()` / `load_state_dict()` / `parameters()` are
checkpoint-compatible with nihaomiao/CVPR23_LFDM while the forward pass is our own executor.

Also: deterministic synthetic checkpoints (numpy PCG64, platform independent) used by the golden
fixtures, the tests and bench.py - there is no network for real checkpoints.
"""
import math

import numpy as np
import torch
from torch import nn


_WEIGHTS_EPOCH = [0]


def weights_epoch():
    """Counter of parameter writes torch's `_version` cannot see (raw-pointer kernels, i.e. FlatAdam.step)."""
    return _WEIGHTS_EPOCH[0]


def bump_weights_epoch():
    _WEIGHTS_EPOCH[0] += 1


_BUFFERS_EPOCH = [0]


def buffers_epoch():
    """Counts (re)allocations of persistent device buffers whose RAW POINTERS a captured hipGraph may hold without owning them (cached
    Winograd filter packs of autograd._pack_wino, the fixed-point scatter accumulator of lfae_ops): a graph captured at another epoch must
    be captured again (LFAETrainer.step_graphed)."""
    return _BUFFERS_EPOCH[0]


def bump_buffers_epoch():
    _BUFFERS_EPOCH[0] += 1


class ParamTree(nn.Module):
    def _walk(self, parts, create):
        node = self
        for p in parts:
            child = node._modules.get(p)
            if child is None:
                if not create:
                    raise KeyError(".".join(parts))
                child = ParamTree()
                node.add_module(p, child)
            node = child
        return node

    def put(self, key, tensor, buffer=False):
        parts = key.split(".")
        node = self._walk(parts[:-1], True)
        if buffer:
            node.register_buffer(parts[-1], tensor)
        else:
            node.register_parameter(parts[-1], nn.Parameter(tensor))

    def get(self, key):
        parts = key.split(".")
        node = self._walk(parts[:-1], False)
        return getattr(node, parts[-1])

    def has(self, key):
        try:
            self.get(key)
            return True
        except (KeyError, AttributeError):
            return False


# init kinds: ("uniform", bound) | ("ones",) | ("zeros",) | ("normal",) | ("const", tensor) | ("var",)
def _conv_entries(prefix, cout, cin, *kernel, bias=True):
    fan_in = cin * int(np.prod(kernel)) if kernel else cin
    bound = 1.0 / math.sqrt(fan_in)
    out = [(prefix + "weight", (cout, cin) + tuple(kernel), ("uniform", bound), False)]
    if bias:
        out.append((prefix + "bias", (cout,), ("uniform", bound), False))
    return out


def _bn_entries(prefix, c):
    return [
        (prefix + "weight", (c,), ("ones",), False),
        (prefix + "bias", (c,), ("zeros",), False),
        (prefix + "running_mean", (c,), ("zeros",), True),
        (prefix + "running_var", (c,), ("var",), True),
        (prefix + "num_batches_tracked", (), ("const", torch.tensor(0, dtype=torch.long)), True),
    ]


def rotary_freqs(dim=32, theta=10000.0):
    return 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))


def unet_spec(dim=64, dim_mults=(1, 2, 4, 8), channels=259, out_grid_dim=2, out_conf_dim=1, heads=8,
              dim_head=32, cond_dim=768, learn_null_cond=False, use_deconv=True, init_kernel_size=7):
    """(key, shape, init, is_buffer) for Unet3D - mirrors the module tree of
    DM/modules/video_flow_diffusion.py:368-509."""
    hidden = heads * dim_head
    time_dim = dim * 4
    emb_dim = time_dim + cond_dim
    spec = []
    if learn_null_cond:      # a root-level nn.Parameter: named_parameters() yields it before every sub-module's (reference :438-440)
        spec.append(("null_cond_emb", (1, cond_dim), ("normal",), False))
    spec += [("time_rel_pos_bias.relative_attention_bias.weight", (32, heads), ("normal",), False)]
    spec += _conv_entries("init_conv.", dim, channels, 1, init_kernel_size, init_kernel_size)

    def temporal(prefix, c):
        # registration order = the reference's named_parameters() order (PreNorm registers `fn` before `norm`, :182-186), so
        # that a torch.optim state_dict saved by the reference (indexed by parameter position) loads into FlatAdam
        return [
            (prefix + "fn.fn.fn.to_qkv.weight", (3 * hidden, c), ("uniform", 1 / math.sqrt(c)), False),
            (prefix + "fn.fn.fn.to_out.weight", (c, hidden), ("uniform", 1 / math.sqrt(hidden)), False),
            (prefix + "fn.fn.fn.rotary_emb.freqs", (dim_head // 2,), ("const", rotary_freqs(dim_head)), True),
            (prefix + "fn.norm.gamma", (1, c, 1, 1, 1), ("ones",), False),
        ]

    def spatial_linear(prefix, c):
        return (_conv_entries(prefix + "fn.fn.to_qkv.", 3 * hidden, c, 1, 1, bias=False)
                + _conv_entries(prefix + "fn.fn.to_out.", c, hidden, 1, 1)
                + [(prefix + "fn.norm.gamma", (1, c, 1, 1, 1), ("ones",), False)])

    def resblock(prefix, cin, cout, cond=True):
        out = []
        if cond:
            out += _conv_entries(prefix + "mlp.1.", 2 * cout, emb_dim)
        for blk, ci in (("block1.", cin), ("block2.", cout)):
            out += _conv_entries(prefix + blk + "proj.", cout, ci, 1, 3, 3)
            out += [(prefix + blk + "norm.weight", (cout,), ("ones",), False),
                    (prefix + blk + "norm.bias", (cout,), ("zeros",), False)]
        if cin != cout:
            out += _conv_entries(prefix + "res_conv.", cout, cin, 1, 1, 1)
        return out

    spec += temporal("init_temporal_attn.", dim)
    spec += _conv_entries("time_mlp.1.", time_dim, dim)
    spec += _conv_entries("time_mlp.3.", time_dim, time_dim)

    dims = [dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    n_res = len(in_out)
    for lvl, (ci, co) in enumerate(in_out):
        p = "downs.%d." % lvl
        spec += resblock(p + "0.", ci, co) + resblock(p + "1.", co, co)
        spec += spatial_linear(p + "2.", co) + temporal(p + "3.", co)
        if lvl < n_res - 1:
            spec += _conv_entries(p + "4.", co, co, 1, 4, 4)
    # (the reference creates `downs` and `ups` before the mid_* modules, :455-456: same registration order here)
    for lvl, (ci, co) in enumerate(reversed(in_out)):
        p = "ups.%d." % lvl
        spec += resblock(p + "0.", co * 2, ci) + resblock(p + "1.", ci, ci)
        spec += spatial_linear(p + "2.", ci) + temporal(p + "3.", ci)
        if lvl < n_res - 1:
            if use_deconv:   # ConvTranspose3d weight layout (Cin, Cout, 1, 4, 4); fan_in uses dim 1
                b = 1.0 / math.sqrt(ci * 16)
                spec += [(p + "4.weight", (ci, ci, 1, 4, 4), ("uniform", b), False),
                         (p + "4.bias", (ci,), ("uniform", b), False)]
            else:
                spec += _conv_entries(p + "4.1.", ci, ci, 1, 3, 3)
    mid = dims[-1]
    spec += resblock("mid_block1.", mid, mid)
    spec += [("mid_spatial_attn.fn.fn.fn.to_qkv.weight", (3 * hidden, mid), ("uniform", 1 / math.sqrt(mid)), False),
             ("mid_spatial_attn.fn.fn.fn.to_out.weight", (mid, hidden), ("uniform", 1 / math.sqrt(hidden)), False),
             ("mid_spatial_attn.fn.norm.gamma", (1, mid, 1, 1, 1), ("ones",), False)]
    spec += temporal("mid_temporal_attn.", mid)
    spec += resblock("mid_block2.", mid, mid)
    for head, od in (("final_conv.", out_grid_dim), ("occlusion_map.", out_conf_dim)):
        spec += resblock(head + "0.", dim * 2, dim, cond=False)
        spec += _conv_entries(head + "1.", od, dim, 1, 1, 1)
    return spec


def generator_spec(num_channels=3, block_expansion=64, max_features=512, num_down_blocks=2,
                   num_bottleneck_blocks=6, num_regions=10, with_flow_predictor=True,
                   fp_block_expansion=64, fp_max_features=1024, fp_num_blocks=5, use_deformed_source=True):
    """State-dict layout of LFAE Generator (LFAE/modules/generator.py:23-56, util.py).  The
    pixelwise_flow_predictor entries are held for checkpoint compatibility (training pseudo-GT path)."""
    spec = []
    if with_flow_predictor:
        hp = "pixelwise_flow_predictor.hourglass."
        in_f = (num_regions + 1) * (num_channels * int(bool(use_deformed_source)) + 1)      # pixelwise_flow_predictor.py:28
        for i in range(fp_num_blocks):
            ci = in_f if i == 0 else min(fp_max_features, fp_block_expansion * (2 ** i))
            co = min(fp_max_features, fp_block_expansion * (2 ** (i + 1)))
            spec += _conv_entries(hp + "encoder.down_blocks.%d.conv." % i, co, ci, 3, 3)
            spec += _bn_entries(hp + "encoder.down_blocks.%d.norm." % i, co)
        for j, i in enumerate(reversed(range(fp_num_blocks))):
            ci = (1 if i == fp_num_blocks - 1 else 2) * min(fp_max_features, fp_block_expansion * (2 ** (i + 1)))
            co = min(fp_max_features, fp_block_expansion * (2 ** i))
            spec += _conv_entries(hp + "decoder.up_blocks.%d.conv." % j, co, ci, 3, 3)
            spec += _bn_entries(hp + "decoder.up_blocks.%d.norm." % j, co)
        out_f = fp_block_expansion + in_f
        spec += _conv_entries("pixelwise_flow_predictor.mask.", num_regions + 1, out_f, 7, 7)
        spec += _conv_entries("pixelwise_flow_predictor.occlusion.", 1, out_f, 7, 7)
        spec.append(("pixelwise_flow_predictor.down.weight", (num_channels, 1, 13, 13),
                     ("const", antialias_kernel(num_channels, 0.25)), True))
    spec += _conv_entries("first.conv.", block_expansion, num_channels, 7, 7)
    spec += _bn_entries("first.norm.", block_expansion)
    for i in range(num_down_blocks):
        ci = min(max_features, block_expansion * (2 ** i))
        co = min(max_features, block_expansion * (2 ** (i + 1)))
        spec += _conv_entries("down_blocks.%d.conv." % i, co, ci, 3, 3)
        spec += _bn_entries("down_blocks.%d.norm." % i, co)
    for i in range(num_down_blocks):
        ci = min(max_features, block_expansion * (2 ** (num_down_blocks - i)))
        co = min(max_features, block_expansion * (2 ** (num_down_blocks - i - 1)))
        spec += _conv_entries("up_blocks.%d.conv." % i, co, ci, 3, 3)
        spec += _bn_entries("up_blocks.%d.norm." % i, co)
    cb = min(max_features, block_expansion * (2 ** num_down_blocks))
    for i in range(num_bottleneck_blocks):
        p = "bottleneck.r%d." % i
        spec += _conv_entries(p + "conv1.", cb, cb, 3, 3) + _conv_entries(p + "conv2.", cb, cb, 3, 3)
        spec += _bn_entries(p + "norm1.", cb) + _bn_entries(p + "norm2.", cb)
    spec += _conv_entries("final.", num_channels, block_expansion, 7, 7)
    return spec


def antialias_kernel(channels, scale):
    """Gaussian kernel buffer of AntiAliasInterpolation2d (LFAE/modules/util.py:222-252)."""
    sigma = (1 / scale - 1) / 2
    ks = 2 * round(sigma * 4) + 1
    ax = torch.arange(ks, dtype=torch.float32)
    g = torch.exp(-(ax - (ks - 1) / 2) ** 2 / (2 * sigma ** 2))
    k2 = g[:, None] * g[None, :]
    k2 = k2 / k2.sum()
    return k2.view(1, 1, ks, ks).repeat(channels, 1, 1, 1)


def build_tree(tree, spec):
    """Registers every spec entry on `tree` with PyTorch-default style initialisation
    (uniform +-1/sqrt(fan_in) for conv/linear, N(0,1) embeddings, unit norms)."""
    for key, shape, init, is_buffer in spec:
        kind = init[0]
        if kind == "uniform":
            t = torch.empty(shape).uniform_(-init[1], init[1])
        elif kind == "ones" or kind == "var":
            t = torch.ones(shape)
        elif kind == "zeros":
            t = torch.zeros(shape)
        elif kind == "normal":
            t = torch.randn(shape)
        elif kind == "const":
            t = init[1].clone()
        else:
            raise ValueError(kind)
        tree.put(key, t, buffer=is_buffer)
    return tree


def synthetic_state_dict(spec, seed):
    """Deterministic, platform-independent synthetic checkpoint for a spec: weights uniform
    +-1/sqrt(fan_in); norm scales 1 + 0.1 N(0,1), norm shifts 0.1 N(0,1); BN running stats
    non-trivial.  Keys are visited in sorted order so the result does not depend on spec order."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for key, shape, init, _ in sorted(spec, key=lambda e: e[0]):
        kind = init[0]
        n = int(np.prod(shape)) if len(shape) else 1
        if kind == "uniform":
            v = rng.uniform(-init[1], init[1], n)
        elif kind == "ones":
            v = 1.0 + 0.1 * rng.standard_normal(n)
        elif kind == "zeros":
            v = 0.1 * rng.standard_normal(n)
        elif kind == "var":
            v = rng.uniform(0.5, 1.5, n)
        elif kind == "normal":
            v = rng.standard_normal(n)
        elif kind == "const":
            out[key] = init[1].clone()
            continue
        else:
            raise ValueError(kind)
        out[key] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape)).clone()
    return out


def _hourglass_entries(prefix, block_expansion, in_features, num_blocks, max_features, decoder=True):
    """Encoder / Decoder of LFAE/modules/util.py:153-214 (conv3x3 + BatchNorm per block)."""
    spec = []
    for i in range(num_blocks):
        ci = in_features if i == 0 else min(max_features, block_expansion * (2 ** i))
        co = min(max_features, block_expansion * (2 ** (i + 1)))
        spec += _conv_entries(prefix + "encoder.down_blocks.%d.conv." % i, co, ci, 3, 3)
        spec += _bn_entries(prefix + "encoder.down_blocks.%d.norm." % i, co)
    if decoder:
        for j, i in enumerate(reversed(range(num_blocks))):
            ci = (1 if i == num_blocks - 1 else 2) * min(max_features, block_expansion * (2 ** (i + 1)))
            co = min(max_features, block_expansion * (2 ** i))
            spec += _conv_entries(prefix + "decoder.up_blocks.%d.conv." % j, co, ci, 3, 3)
            spec += _bn_entries(prefix + "decoder.up_blocks.%d.norm." % j, co)
    return spec


def region_predictor_spec(num_regions=10, num_channels=3, block_expansion=32, max_features=1024,
                          num_blocks=5, scale_factor=0.25, estimate_affine=False, pca_based=False, **_):
    """RegionPredictor state-dict layout (LFAE/modules/region_predictor.py:28-50); the FOMM-like regression head `jacobian` exists when
    estimate_affine and not pca_based (:43-49).  estimate_affine / pca_based default as the reference constructor does (:34-35): a tree
    built from a yaml without those keys has the reference's keys."""
    spec = _hourglass_entries("predictor.", block_expansion, num_channels, num_blocks, max_features)
    spec += _conv_entries("regions.", num_regions, block_expansion + num_channels, 7, 7)
    if estimate_affine and not pca_based:
        spec += _conv_entries("jacobian.", 4, block_expansion + num_channels, 7, 7)
    if scale_factor != 1:
        spec.append(("down.weight", (num_channels, 1, 13, 13), ("const", antialias_kernel(num_channels, scale_factor)), True))
    return spec


BG_FC_OUT = {"shift": 2, "affine": 6, "perspective": 8}        # bg_motion_predictor.py:27-40; bg_type 'zero' has no parameters at all
BG_FC_BIAS = {"shift": (0, 0), "affine": (1, 0, 0, 0, 1, 0), "perspective": (1, 0, 0, 0, 1, 0, 0, 0)}


def bg_predictor_spec(num_channels=3, block_expansion=32, max_features=1024, num_blocks=5, bg_type="zero", **_):
    """BGMotionPredictor state-dict layout (LFAE/modules/bg_motion_predictor.py:15-40) for every bg_type the reference accepts (default 'zero', :20)."""
    if bg_type == "zero":
        return []
    if bg_type not in BG_FC_OUT:
        raise ValueError("bg_type %r (the reference accepts 'zero', 'shift', 'affine', 'perspective')" % (bg_type,))
    spec = _hourglass_entries("", block_expansion, num_channels * 2, num_blocks, max_features, decoder=False)
    co = min(max_features, block_expansion * (2 ** num_blocks))
    n = BG_FC_OUT[bg_type]
    spec += [("fc.weight", (n, co), ("zeros",), False), ("fc.bias", (n,), ("zeros",), False)]
    return spec


def bg_matrix(pred, bs, bg_type):
    """(bs, 3, 3) background transform from the fc output (bg_motion_predictor.py:44-57); pred is None for bg_type 'zero'."""
    import torch
    if bg_type == "zero":
        return None
    dev, dt = pred.device, pred.dtype
    one, zero = torch.ones(bs, 1, device=dev, dtype=dt), torch.zeros(bs, 1, device=dev, dtype=dt)
    if bg_type == "shift":
        rows = torch.cat((one, zero, pred[:, 0:1], zero, one, pred[:, 1:2], zero, zero, one), dim=1)
    elif bg_type == "affine":
        rows = torch.cat((pred, zero, zero, one), dim=1)
    else:
        rows = torch.cat((pred, one), dim=1)
    return rows.view(bs, 3, 3)


VGG19_CONVS = ((1, 0, 3, 64), (2, 2, 64, 64), (2, 5, 64, 128), (3, 7, 128, 128), (3, 10, 128, 256), (4, 12, 256, 256), (4, 14, 256, 256),
               (4, 16, 256, 256), (4, 19, 256, 512), (5, 21, 512, 512), (5, 23, 512, 512), (5, 25, 512, 512), (5, 28, 512, 512))
VGG19_POOLS_BEFORE = (5, 10, 19, 28)       # torchvision vgg19.features: a 2x2 max-pool sits in front of these convolutions


def vgg19_spec():
    """State-dict layout of the perceptual-loss network of LFAE stage-1 training (LFAE/modules/model.py:19-59): the first 30 layers of
    torchvision's vgg19.features split into slice1..slice5, every layer keeping its torchvision index, plus the ImageNet mean / std
    held as frozen parameters.  (slice, index, c_in, c_out) per convolution in VGG19_CONVS."""
    spec = []
    for sl, idx, ci, co in VGG19_CONVS:
        spec += _conv_entries("slice%d.%d." % (sl, idx), co, ci, 3, 3)
    spec += [("mean", (1, 3, 1, 1), ("const", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)), False),
             ("std", (1, 3, 1, 1), ("const", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)), False)]
    return spec


def vgg19_from_torchvision(state_dict):
    """torchvision `vgg19().state_dict()` (or its `.features` one) -> the keys of vgg19_spec: features.N.* -> sliceK.N.*"""
    slice_of = {idx: sl for sl, idx, _, _ in VGG19_CONVS}
    out = {}
    for k, v in state_dict.items():
        parts = k.split(".")
        if parts[0] == "features":
            parts = parts[1:]
        if len(parts) == 2 and parts[0].isdigit() and int(parts[0]) in slice_of:
            out["slice%d.%s.%s" % (slice_of[int(parts[0])], parts[0], parts[1])] = v
    return out


def synthetic_vgg19_state(seed=1919, gain=2.4):
    """Deterministic stand-in for the ImageNet VGG-19 weights (no network): synthetic_state_dict scaled by `gain` so that the feature
    magnitudes stay O(1) through the 13 convolutions (uniform +-1/sqrt(fan_in) alone shrinks them ~6x per layer and the deep slices
    would not contribute to the perceptual loss)."""
    sd = synthetic_state_dict(vgg19_spec(), seed)
    for k in sd:
        if k.endswith(".weight"):
            sd[k] = sd[k] * gain
    return sd
