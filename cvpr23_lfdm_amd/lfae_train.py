"""LFAE stage-1 training: the reconstruction step of the latent-flow auto-encoder (reference LFAE/train.py:35-173 and
LFAE/modules/model.py:141-217 `ReconstructionModel`) - region predictor, background-motion predictor and generator trained
jointly on (source, driving) frame pairs with the pyramid perceptual loss (VGG-19 features, model.py:19-82) and the two
equivariance losses (random affine + thin-plate-spline transform, model.py:85-136).

How it runs here.  The step is >95 % convolution FLOPs (generator at full resolution, VGG-19 on four pyramid levels, three
hourglasses): every Conv2d - forward, data gradient, weight gradient and bias gradient - is the library's native kernel family
through `autograd.ConvCL` (channels-last rows, Winograd where the channel counts allow, gradients written into the optimizer's flat
slots), the same `torch.autograd.Function`s the DM training step uses.  What connects the convolutions stays on tensors in
channels-last memory format, so entering / leaving a convolution is a view: BatchNorm with batch statistics (per rank, as the
reference's nn.DataParallel replicas compute them with `use_sync_bn: False`), ReLU, 2x2 pooling, nearest up-sampling, softmax
heat-maps, `grid_sample` warps and the 2x2 SVD (on the device: `lfdm_svd2x2_sym_f32` with LAPACK's sign conventions and an analytic
backward - region_predictor.py:16-26 round-trips through the host).  Parameters live in the
same `ParamTree`s as on the sampling path (reference state-dict keys: a reference checkpoint loads, a checkpoint written here loads
into the reference).  The optimizer is the flat fused Adam (betas (0.5, 0.999), LFAE/train.py:38-40) with the DM path's bucketed
gradient all-reduce for one-process-per-GPU data parallelism.

Covered: the RGB configurations of the LFDM yaml files and their switches - pca_based / estimate_affine (Jacobian head) region
parameters, bg_type zero / shift / affine / perspective, use_deformed_source on or off (lfae_predictors.py); the AVD network is not.
"""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import autograd as A
from . import lfae_ops as L
from . import ops
from . import params as P
from .params import ParamTree


# ------------------------------------------------------------------------------------------------ layers
def _cl(x):
    """NCHW tensor in channels-last memory (a no-op when it already is)."""
    return x.contiguous(memory_format=torch.channels_last)


_NBT_PENDING = []      # num_batches_tracked buffers touched by the current forward (one multi-tensor add at its end, not one launch per layer)


def flush_batches_tracked():
    """num_batches_tracked += (uses this forward) for every BatchNorm that ran in training mode (nn.BatchNorm2d does it per call)."""
    if not _NBT_PENDING:
        return
    seen = {}
    for t in _NBT_PENDING:
        seen.setdefault(id(t), [t, 0])[1] += 1
    del _NBT_PENDING[:]
    torch._foreach_add_([v[0] for v in seen.values()], [v[1] for v in seen.values()])


class _PadParam(Function):
    """A parameter zero-padded to the channel multiples the kernels take (RGB inputs, the 10 / 11 / 1 / 3-channel heads), into a
    persistent buffer whose pad region stays zero: one copy per use instead of F.pad's fill + copy, the gradient is the slice back."""

    @staticmethod
    def forward(ctx, p, shape):
        key = (id(p), shape)
        hit = _PAD_BUFFERS.get(key)
        if hit is None or hit[0]() is not p or hit[1].device != p.device:
            import weakref
            if len(_PAD_BUFFERS) > 256:            # entries of deleted parameters (ids are re-used)
                for k in [k for k, v in _PAD_BUFFERS.items() if v[0]() is None]:
                    del _PAD_BUFFERS[k]
            hit = [weakref.ref(p), torch.zeros(shape, dtype=p.dtype, device=p.device), None]
            _PAD_BUFFERS[key] = hit
            P.bump_buffers_epoch()
        ctx.sl = tuple(slice(0, n) for n in p.shape)
        # (valid while neither torch nor a raw-pointer optimizer step has written p - the tag of autograd._pack_wino; a second use in the
        # same step - the frozen VGG filters, eight times - must not write the buffer again: earlier uses have saved it for their backward)
        tag = (p._version, P.weights_epoch() if p.requires_grad else -1, p.data_ptr())
        if hit[2] != tag:
            hit[1][ctx.sl].copy_(p)
            hit[2] = tag
        out = hit[1].detach()
        out._lfdm_pack_owner = hit[1]        # autograd._pack_wino: the persistent tensor a cached Winograd pack of this alias belongs to
        return out

    @staticmethod
    def backward(ctx, d):
        return d[ctx.sl], None


_PAD_BUFFERS = {}


def refresh_pad_buffers():
    """Bring every padded-parameter buffer up to date with its parameter (what _PadParam.forward does at its next use) - for callers that
    REPLAY a captured forward: a replay runs no Python, so a parameter torch has rewritten since the capture (load_state_dict) would be
    read through a stale buffer."""
    for key, hit in list(_PAD_BUFFERS.items()):
        p = hit[0]()
        if p is None:
            del _PAD_BUFFERS[key]
            continue
        tag = (p._version, P.weights_epoch() if p.requires_grad else -1, p.data_ptr())
        if hit[2] != tag and hit[1].device == p.device:
            with torch.no_grad():
                hit[1][tuple(slice(0, n) for n in p.shape)].copy_(p)
            hit[2] = tag


def conv2d(x, weight, bias, padding, relu=False, residual=None):
    """nn.Conv2d (stride 1, square kernel) through the native kernels.  x: (N, C, H, W); returns a channels-last (N, O, H', W').
    Channel counts that are not multiples of 4 (RGB inputs, the region / mask / occlusion / RGB heads) are zero-padded (_PadParam).
    residual: an (N, O, H', W') tensor added in the convolution's epilogue."""
    n, c, h, w = x.shape
    cout, k = weight.shape[0], weight.shape[-1]
    rows = x.permute(0, 2, 3, 1).reshape(n * h * w, c)
    cin = weight.shape[1]
    cp, op = -cin % 4, -cout % 4
    if c == cin and cp:                 # (c == cin + cp: the producer already wrote zero pad channels - lfae_ops.blur_down(rows4=True))
        rows = F.pad(rows, (0, cp))
    assert rows.shape[1] == cin + cp
    if cp or op:
        weight = _PadParam.apply(weight, (cout + op, cin + cp) + tuple(weight.shape[2:]))
        bias = _PadParam.apply(bias, (cout + op,)) if bias is not None and op else bias
    res_rows = None
    if residual is not None:
        assert not op
        res_rows = residual.permute(0, 2, 3, 1).reshape(-1, cout)
    y = A.conv_cl(rows, weight, bias, residual=res_rows, n_img=n, hi=h, wi=w, pad=(padding, padding), relu=relu)   # relu: in the epilogue
    ho, wo = h + 2 * padding - k + 1, w + 2 * padding - k + 1
    y = y.view(n, ho, wo, cout + op).permute(0, 3, 1, 2)
    return y[:, :cout] if op else y


class _Net:
    """Forward helpers over one ParamTree (reference state-dict keys); `training` selects batch statistics (and updates the running
    ones, momentum 0.1 like nn.BatchNorm2d) or the stored statistics."""

    def __init__(self, tree, training=True, segments=1):
        self.t, self.training, self.segments = tree, training, segments

    def conv(self, x, prefix, padding, relu=False, residual=None):
        return conv2d(x, self.t.get(prefix + "weight"), self.t.get(prefix + "bias") if self.t.has(prefix + "bias") else None, padding, relu,
                      residual)

    def bn_relu(self, x, prefix, fork=False):
        """BatchNorm2d -> ReLU (every BatchNorm of the LFAE networks is followed by one: util.py:84-90, 108-112, 128-133, 146-150).  Training:
        one native forward (batch statistics, running statistics updated in place), one native backward (lfae_ops.BatchNormReLU)."""
        g = self.t.get
        if self.training:
            _NBT_PENDING.extend([g(prefix + "num_batches_tracked")] * self.segments)
            return L.BatchNormReLU.apply(x, g(prefix + "weight"), g(prefix + "bias"), g(prefix + "running_mean"), g(prefix + "running_var"),
                                         0.1, 1e-5, True, self.segments, fork)
        assert not fork
        return F.relu(F.batch_norm(x, g(prefix + "running_mean"), g(prefix + "running_var"), g(prefix + "weight"), g(prefix + "bias"),
                                   False, 0.1, 1e-5))

    def conv_bn_relu(self, x, prefix, padding=1):          # SameBlock2d / the body of Down- and UpBlock2d (util.py:95-150)
        return self.bn_relu(self.conv(x, prefix + "conv.", padding), prefix + "norm.")

    def down_block(self, x, prefix):                      # DownBlock2d: conv -> BN -> ReLU -> AvgPool 2x2
        return L.pool2(self.conv_bn_relu(x, prefix), "avg")

    def up_block(self, x, prefix):                        # UpBlock2d: nearest x2 -> conv -> BN -> ReLU
        return self.conv_bn_relu(L.pool2(x, "up"), prefix)

    def res_block(self, x, prefix):                       # ResBlock2d (util.py:70-92): pre-activation, identity skip
        if not self.training:
            out = self.conv(self.bn_relu(x, prefix + "norm1."), prefix + "conv1.", 1)
            return self.conv(self.bn_relu(out, prefix + "norm2."), prefix + "conv2.", 1) + x
        # training: the skip is conv2's residual operand and its gradient is summed inside norm1's backward kernel - no add launches
        out, x = self.bn_relu(x, prefix + "norm1.", fork=True)
        out = self.conv(out, prefix + "conv1.", 1)
        return self.conv(self.bn_relu(out, prefix + "norm2."), prefix + "conv2.", 1, residual=x)

    def hourglass(self, x, prefix, num_blocks, decoder=True):
        """Encoder / Decoder of util.py:153-214.  decoder=False returns the encoder's feature list."""
        outs = [x]
        for i in range(num_blocks):
            outs.append(self.down_block(outs[-1], "%sencoder.down_blocks.%d." % (prefix, i)))
        if not decoder:
            return outs
        out = outs.pop()
        for j in range(num_blocks):
            out = self.up_block(out, "%sdecoder.up_blocks.%d." % (prefix, j))
            out = torch.cat([out, outs.pop()], dim=1)
        return out


_CONST = {}


def make_coordinate_grid(h, w, like):
    """util.py:51-67: (h, w, 2) grid of (x, y) in [-1, 1].  A constant per (h, w, device): built once (9 launches per call otherwise, 8 calls
    per step); callers only read it."""
    key = ("grid", h, w, str(like.device), like.dtype)
    g = _CONST.get(key)
    if g is None:
        x = 2 * (torch.arange(w, dtype=like.dtype, device=like.device) / (w - 1)) - 1
        y = 2 * (torch.arange(h, dtype=like.dtype, device=like.device) / (h - 1)) - 1
        g = _CONST[key] = torch.stack((x.view(1, -1).expand(h, w), y.view(-1, 1).expand(h, w)), dim=2)
    return g


def antialias_down(x, weight, scale, rows4=False, affine=None):
    """AntiAliasInterpolation2d (util.py:217-264): Gaussian blur (depthwise) then every 1/scale-th pixel - one native launch that computes
    only the kept pixels (lfae_ops.BlurDown; backward native too).  rows4: the result as the zero-padded 4-channel rows the next
    convolution reads; affine = (scale, bias) per channel folded into the same launch.  scale == 1 with rows4 / affine is the identity
    "blur" (a 1x1 kernel): the re-layout / normalisation alone."""
    if scale == 1 and not rows4 and affine is None:
        return x
    if scale == 1:
        key = ("ones", x.shape[1], str(x.device))
        weight = _CONST.get(key)
        if weight is None:
            weight = _CONST[key] = x.new_ones(x.shape[1], 1, 1)
    sc, bi = affine if affine is not None else (None, None)
    return L.BlurDown.apply(x, weight, int(round(1 / scale)), rows4, sc, bi)


# The reference writes its per-pixel 2x2 / 3x3 algebra as torch.matmul over (B, K, h, w, 2, 2) operands: hundreds of thousands of
# 2x2 products per call, which a GEMM library runs as a batched GEMM at ~1 ms each (39 % of a step when this file did the same,
# profiles/r04_x_lfae_*).  The same sums written out as broadcast multiply-adds are a handful of element-wise launches.
# 2x2 inverses and products are written out too: torch.inverse / torch.matmul on (B, K, 2, 2) tensors run rocSOLVER / Tensile kernels and
# torch.inverse synchronises with the host to check for singular inputs.
def _sign22(like):
    key = ("sign22", like.device)
    if key not in _CONST:
        _CONST[key] = (torch.tensor([[1.0, -1.0], [-1.0, 1.0]], device=like.device), torch.tensor([1.0, -1.0], device=like.device))
    return _CONST[key]


# The 2x2 algebra as a few broadcast launches each (torch.inverse / matmul would bring rocSOLVER, a host sync and Tensile kernels into the
# step; one launch per scalar product made ~25 launches per call).  Same arithmetic: every entry is the two-term sum of products.
def _inv2(m):
    sign, pm = _sign22(m)
    rev = m.flip(-1, -2)                                   # [[d, c], [b, a]]
    det = ((m * rev)[..., 0, :] * pm).sum(-1)              # a d - b c
    return rev.transpose(-1, -2) * sign / det.unsqueeze(-1).unsqueeze(-1)


def _mm2(x, y):
    return (x.unsqueeze(-1) * y.unsqueeze(-3)).sum(-2)


def _mat2_vec(m, v):
    """(..., 2, 2) @ (..., 2) with broadcasting."""
    return (m * v.unsqueeze(-2)).sum(-1)


def region2gaussian(center, covar, h, w):
    """util.py:22-48 with a matrix covariance: exp(-0.5 d^T covar^-1 d) on the coordinate grid."""
    grid = make_coordinate_grid(h, w, center).view(1, 1, h, w, 2)
    d = grid - center.view(*center.shape[:2], 1, 1, 2)
    inv_t = _inv2(covar).transpose(-1, -2).unsqueeze(-3).unsqueeze(-3)
    under = (_mat2_vec(inv_t, d) * d).sum(-1)
    return torch.exp(-0.5 * under)


# ------------------------------------------------------------------------------------------------ the three networks
def region_predictor_forward(tree, x, cfg, training=True, segments=1):
    """RegionPredictor.forward, pca_based (region_predictor.py:52-117) -> shift, covar, affine, heatmap, u, d.
    segments = S: x is S equal batches stacked along dim 0 that the reference sends through S separate calls (source, driving,
    transformed driving frames: model.py:157-160, :190-191); every BatchNorm keeps per-call statistics (lfae_ops.BatchNormReLU), the
    rest of the network is per-sample, so the result is the concatenation of the S calls' results."""
    net = _Net(tree, training, segments)
    x = antialias_down(x, tree.get("down.weight") if cfg["scale_factor"] != 1 else None, cfg["scale_factor"], rows4=x.shape[1] == 3)
    fmap = net.hourglass(_cl(x), "predictor.", cfg["num_blocks"])
    pred = net.conv(fmap, "regions.", cfg.get("pad", 3))
    shp = pred.shape
    region = F.softmax(pred.reshape(shp[0], shp[1], -1) / cfg["temperature"], dim=2).view(*shp)
    grid = make_coordinate_grid(shp[2], shp[3], region).view(1, 1, shp[2], shp[3], 2)
    r = region.unsqueeze(-1)
    mean = (r * grid).sum(dim=(2, 3))
    if not cfg.get("pca_based", False):           # region_predictor.py:98-108: the regression head, or centres + heat-maps only
        out = {"shift": mean, "heatmap": region}
        if cfg.get("estimate_affine", False):
            jmap = net.conv(fmap, "jacobian.", cfg.get("pad", 3)).reshape(shp[0], 1, 4, shp[2] * shp[3])
            jac = (region.reshape(shp[0], shp[1], 1, -1) * jmap).sum(dim=-1).view(shp[0], shp[1], 2, 2)
            out["affine"] = jac
            out["covar"] = _mm2(jac, jac.transpose(-1, -2))
        return out
    mean_sub = grid - mean.unsqueeze(-2).unsqueeze(-2)
    covar = ((mean_sub.unsqueeze(-1) * mean_sub.unsqueeze(-2)) * r.unsqueeze(-1)).sum(dim=(2, 3))       # outer product per pixel
    # region_predictor.py:21-25 moves the covariances to the host for torch.svd; here: LAPACK's 2x2 path in closed form on the device,
    # analytic backward (lfae_ops.Svd2x2Sym) - no host round trip in the step
    u, s = L.Svd2x2Sym.apply(covar.reshape(-1, 2, 2))
    sq = s ** 0.5
    d = torch.diag_embed(sq)
    return {"shift": mean, "covar": covar, "heatmap": region, "affine": (u * sq.unsqueeze(-2)).view(*covar.shape), "u": u, "d": d}      # U @ diag


def bg_predictor_forward(tree, source, driving, cfg, training=True):
    """BGMotionPredictor.forward (bg_motion_predictor.py:42-57) -> (B, 3, 3): identity ('zero'), translation, affine or perspective."""
    bg_type = cfg.get("bg_type", "zero")
    bs = source.shape[0]
    if bg_type == "zero":
        return torch.eye(3, dtype=source.dtype, device=source.device).unsqueeze(0).repeat(bs, 1, 1)
    net = _Net(tree, training)
    feats = net.hourglass(_cl(torch.cat([source, driving], dim=1)), "", cfg["num_blocks"], decoder=False)
    pred = F.linear(feats[-1].mean(dim=(2, 3)), tree.get("fc.weight"), tree.get("fc.bias"))
    return P.bg_matrix(pred, bs, bg_type)


def pixelwise_flow_forward(tree, source_image, driving, source, bg_params, cfg, num_regions, revert_axis_swap=True, training=True):
    """PixelwiseFlowPredictor.forward (pixelwise_flow_predictor.py:48-137; covariance heat-maps, deformed sources, occlusion)."""
    net = _Net(tree, training)
    p = "pixelwise_flow_predictor."
    img = antialias_down(source_image, tree.get(p + "down.weight"), cfg["scale_factor"]) if cfg["scale_factor"] != 1 else source_image
    bs, _, h, w = img.shape
    k = num_regions
    if cfg.get("use_covar_heatmap", False):        # pixelwise_flow_predictor.py:53-56 (every LFDM yaml: true); else an isotropic variance
        heat = region2gaussian(driving["shift"], driving["covar"], h, w) - region2gaussian(source["shift"], source["covar"], h, w)
    else:
        var = float(cfg.get("region_var", 0.01))
        gd = make_coordinate_grid(h, w, driving["shift"]).view(1, 1, h, w, 2)
        iso = lambda c: torch.exp(-0.5 * ((gd - c.view(*c.shape[:2], 1, 1, 2)) ** 2).sum(-1) / var)
        heat = iso(driving["shift"]) - iso(source["shift"])
    heat = torch.cat([heat.new_zeros(bs, 1, h, w), heat], dim=1).unsqueeze(2)
    ident = make_coordinate_grid(h, w, heat).view(1, 1, h, w, 2)
    cg = ident - driving["shift"].view(bs, k, 1, 1, 2)
    if "affine" in driving:                        # pixelwise_flow_predictor.py:71-78
        affine = _mm2(source["affine"], _inv2(driving["affine"]))
        if revert_axis_swap:
            affine = affine * torch.sign(affine[:, :, 0:1, 0:1])
        cg = _mat2_vec(affine.unsqueeze(-3).unsqueeze(-3), cg)
    d2s = cg + source["shift"].view(bs, k, 1, 1, 2)
    bg = ident.repeat(bs, 1, 1, 1, 1)
    if bg_params is not None:      # homogeneous 3x3 transform of the identity grid
        m = bg_params.reshape(bs, 1, 1, 1, 3, 3)
        hom = (m[..., :2] * ident.unsqueeze(-2)).sum(-1) + m[..., 2]
        bg = hom[..., :2] / hom[..., 2:]
    sparse = torch.cat([bg, d2s], dim=1)
    # (pixelwise_flow_predictor.py:95-102 repeats the source K+1 times; the native kernel reads one source per K+1 grids)
    deformed = L.GridSample.apply(img, sparse.reshape(bs * (k + 1), h, w, 2), k + 1, False).view(bs, k + 1, -1, h, w)
    # pixelwise_flow_predictor.py:116-119 (the deformed sources are computed either way, like the reference does)
    inp = (torch.cat([heat, deformed], dim=2) if cfg.get("use_deformed_source", True) else heat).reshape(bs, -1, h, w)
    pred = net.hourglass(_cl(inp), p + "hourglass.", cfg["num_blocks"])
    mask = F.softmax(net.conv(pred, p + "mask.", 3), dim=1).unsqueeze(2)
    out = {"optical_flow": (sparse.permute(0, 1, 4, 2, 3) * mask).sum(dim=1).permute(0, 2, 3, 1)}
    if tree.has(p + "occlusion.weight"):
        out["occlusion_map"] = torch.sigmoid(net.conv(pred, p + "occlusion.", 3))
    return out


def _apply_optical(prev, skip, maps):
    """Generator.deform_input + apply_optical (generator.py:59-88): the flow / occlusion maps resized bilinearly to the feature map,
    grid_sample, blend with the decoder state - ONE native launch forward (the sampling path's warp kernel), native backward
    (lfae_ops.ApplyOpticalCL).  maps: (N, 3, fh, fw) planes [flow_x, flow_y, occlusion]."""
    return L.ApplyOpticalCL.apply(_cl(skip), None if prev is None else _cl(prev), maps)


def generator_forward(tree, source_image, driving, source, bg_params, cfg, num_regions, revert_axis_swap=True, training=True):
    """Generator.forward (generator.py:90-128) -> prediction, deformed, optical_flow, occlusion_map, bottle_neck_feat."""
    net = _Net(tree, training)
    nd, use_skips = cfg["num_down_blocks"], cfg.get("skips", False)
    out = net.conv_bn_relu(antialias_down(source_image, None, 1, rows4=source_image.shape[1] == 3), "first.", 3)
    skips = [out]
    for i in range(nd):
        out = net.down_block(out, "down_blocks.%d." % i)
        skips.append(out)
    res = {"bottle_neck_feat": out}
    motion = pixelwise_flow_forward(tree, source_image, driving, source, bg_params, cfg["pixelwise_flow_predictor_params"], num_regions,
                                    revert_axis_swap, training)
    flow, occ = motion["optical_flow"], motion.get("occlusion_map")
    with torch.no_grad():       # generator.py:104: logged only, feeds no loss
        fh, fw = flow.shape[1], flow.shape[2]
        fpl = flow.permute(0, 3, 1, 2).contiguous()
        res["deformed"] = ops.warp_planar(source_image.contiguous(), 1, fpl[:, 0], fpl[:, 1], None, fh, fw, 2 * fh * fw, 0).view(source_image.shape)
    res.update(motion)
    maps = L._maps_planar(flow, occ)
    out = _apply_optical(None, out, maps)
    for i in range(cfg["num_bottleneck_blocks"]):
        out = net.res_block(_cl(out), "bottleneck.r%d." % i)
    for i in range(nd):
        if use_skips:
            out = _apply_optical(out, skips[-(i + 1)], maps)
        out = net.up_block(_cl(out), "up_blocks.%d." % i)
    if use_skips:
        out = _apply_optical(out, skips[0], maps)
    out = torch.sigmoid(net.conv(_cl(out), "final.", 3))
    if use_skips:
        out = L.ApplyOpticalImage.apply(source_image, out, maps)
    res["prediction"] = out
    return res


# ------------------------------------------------------------------------------------------------ losses
class Vgg19(ParamTree):
    """The perceptual-loss network (model.py:19-59): frozen VGG-19 feature slices.  There is no network for the ImageNet weights:
    `load_state_dict(params.vgg19_from_torchvision(torchvision_vgg19.state_dict()), strict=False)` takes a downloaded checkpoint."""

    def __init__(self):
        super().__init__()
        P.build_tree(self, P.vgg19_spec())
        for p in self.parameters():
            p.requires_grad = False

    def features(self, x, prepared=False):
        """prepared: x is already normalised, zero-padded 4-channel rows (ReconstructionModel.pyramid(vgg_input=True))."""
        net = _Net(self, False)
        if not prepared:
            x = _cl((x - self.get("mean")) / self.get("std"))
        outs, cur = [], 1
        for sl, idx, _, _ in P.VGG19_CONVS:
            if sl != cur:
                outs.append(x)
                cur = sl
            if idx in P.VGG19_POOLS_BEFORE:
                x = L.pool2(x, "max")
            x = net.conv(x, "slice%d.%d." % (sl, idx), 1, relu=True)
        outs.append(x)
        return outs


class Transform:
    """Random affine + thin-plate-spline warp for the equivariance losses (model.py:85-136).  `noise` = (theta noise (B, 2, 3), tps
    control parameters (B, 1, points^2)) replays recorded draws; by default they are drawn like the reference draws them."""

    def __init__(self, bs, sigma_affine, sigma_tps=None, points_tps=None, noise=None, device="cpu"):
        if noise is None:
            theta_noise = torch.normal(mean=0, std=sigma_affine * torch.ones([bs, 2, 3]))
            tps_noise = torch.normal(mean=0, std=sigma_tps * torch.ones([bs, 1, points_tps ** 2])) if sigma_tps is not None else None
        else:
            theta_noise, tps_noise = noise
        # (noise already on the device - the graphed step hands its static buffers over - is used where it lies)
        self.theta = (theta_noise + torch.eye(2, 3, device=theta_noise.device).view(1, 2, 3)).to(device)
        self.bs = bs
        self.tps = tps_noise is not None
        if self.tps:
            self.control_points = make_coordinate_grid(points_tps, points_tps, self.theta).view(1, points_tps ** 2, 2)
            self.control_params = tps_noise.to(device)

    def warp_coordinates(self, coordinates):
        theta = self.theta.unsqueeze(1)
        out = _mat2_vec(theta[:, :, :, :2], coordinates) + theta[:, :, :, 2]
        if self.tps:
            dist = torch.abs(coordinates.view(coordinates.shape[0], -1, 1, 2) - self.control_points.view(1, 1, -1, 2)).sum(-1)
            res = (dist ** 2) * torch.log(dist + 1e-6) * self.control_params
            out = out + res.sum(dim=2).view(self.bs, coordinates.shape[1], 1)
        return out

    def transform_frame(self, frame):
        h, w = frame.shape[2:]
        grid = make_coordinate_grid(h, w, frame).view(1, h * w, 2)
        grid = self.warp_coordinates(grid).view(self.bs, h, w, 2)
        return L.GridSample.apply(frame, grid, 1, True)          # F.grid_sample(padding_mode="reflection"); no gradient flows here

    def jacobian(self, coordinates):
        new = self.warp_coordinates(coordinates)
        gx = torch.autograd.grad(new[..., 0].sum(), coordinates, create_graph=True)
        gy = torch.autograd.grad(new[..., 1].sum(), coordinates, create_graph=True)
        return torch.cat([gx[0].unsqueeze(-2), gy[0].unsqueeze(-2)], dim=-2)


class ReconstructionModel:
    """model.py:141-217: one forward of the three networks on a (source, driving) batch and the loss terms.  `generator`,
    `region_predictor`, `bg_predictor` are the package's ParamTrees (cvpr23_lfdm_amd.Generator / RegionPredictor / BGMotionPredictor or
    anything with the reference state-dict keys); model_params / train_params follow config/*.yaml."""

    def __init__(self, region_predictor, bg_predictor, generator, model_params, train_params, vgg=None):
        self.region_predictor, self.bg_predictor, self.generator = region_predictor, bg_predictor, generator
        self.mp, self.tp = model_params, train_params
        self.scales = list(train_params["scales"])
        self.loss_weights = train_params["loss_weights"]
        nc = model_params["num_channels"]
        self.pyramid_kernels = {s: (P.antialias_kernel(nc, s) if s != 1 else None) for s in self.scales}
        self.vgg = vgg if vgg is not None else (Vgg19() if sum(self.loss_weights["perceptual"]) != 0 else None)
        self.training = True
        self._vgg_affine = None

    def to(self, device):
        self._vgg_affine = None
        for k, v in self.pyramid_kernels.items():
            self.pyramid_kernels[k] = None if v is None else v.to(device)
        if self.vgg is not None:
            self.vgg.to(device)
        return self

    def _regions(self, *frames):
        """The region predictor on each batch of frames -> one dict per batch.  All the batches go through the network together (one
        launch sequence instead of len(frames), BatchNorm statistics per batch like the separate calls of model.py:157-160, :190-191)."""
        cfg = dict(self.mp["region_predictor_params"], estimate_affine=self.mp.get("estimate_affine", False))
        if len(frames) == 1 or not self.training or any(f.shape != frames[0].shape for f in frames):
            return [region_predictor_forward(self.region_predictor, f, cfg, self.training) for f in frames]
        b = frames[0].shape[0]
        out = region_predictor_forward(self.region_predictor, torch.cat(frames), cfg, True, segments=len(frames))
        return [{k: v[i * b:(i + 1) * b] for k, v in out.items()} for i in range(len(frames))]

    def pyramid(self, x, vgg_input=False):
        """ImagePyramide (model.py:62-82).  vgg_input: every level leaves its launch as the perceptual network's input - normalised
        ((x - mean) / std, model.py:52) zero-padded 4-channel rows."""
        if not vgg_input:
            return {s: (x if k is None else antialias_down(x, k, s)) for s, k in self.pyramid_kernels.items()}
        if self._vgg_affine is None:
            mean, std = self.vgg.get("mean").reshape(-1), self.vgg.get("std").reshape(-1)
            self._vgg_affine = ((1.0 / std).contiguous(), (-mean / std).contiguous())
        return {s: antialias_down(x, k, s, rows4=True, affine=self._vgg_affine) for s, k in self.pyramid_kernels.items()}

    def forward(self, x, transform_noise=None):
        src, drv = x["source"], x["driving"]
        mp, lw = self.mp, self.loss_weights
        equivariance = lw["equivariance_shift"] + lw["equivariance_affine"] != 0
        if equivariance:
            # (the transform's draws are the forward's only random numbers: building it before the generator changes nothing)
            tr = Transform(drv.shape[0], noise=transform_noise, device=drv.device, **self.tp["transform_params"])
            frame = tr.transform_frame(drv)
            source_rp, driving_rp, trp = self._regions(src, drv, frame)
        else:
            source_rp, driving_rp = self._regions(src, drv)
        bg = bg_predictor_forward(self.bg_predictor, src, drv, mp["bg_predictor_params"], self.training)
        gen = generator_forward(self.generator, src, driving_rp, source_rp, bg, mp["generator_params"], mp["num_regions"],
                                mp.get("revert_axis_swap", True), self.training)
        gen.update({"source_region_params": source_rp, "driving_region_params": driving_rp})
        losses = {}
        if sum(lw["perceptual"]) != 0:
            fast = drv.shape[1] == 3
            with torch.no_grad():
                pyr_real = self.pyramid(drv, vgg_input=fast)
            pyr_gen = self.pyramid(gen["prediction"], vgg_input=fast)
            terms = []
            for s in self.scales:
                x_vgg = self.vgg.features(pyr_gen[s], prepared=fast)
                with torch.no_grad():
                    y_vgg = self.vgg.features(pyr_real[s], prepared=fast)
                for i, wgt in enumerate(lw["perceptual"]):
                    if x_vgg[i].numel() % 4 == 0:
                        terms.append(L.L1Mean.apply(x_vgg[i], y_vgg[i], wgt))           # one launch forward, one backward per term
                    else:
                        terms.append((wgt * torch.abs(x_vgg[i] - y_vgg[i]).mean()).reshape(1))
            losses["perceptual"] = torch.cat(terms).sum()
        if equivariance:
            gen["transformed_frame"], gen["transformed_region_params"] = frame, trp
            if lw["equivariance_shift"] != 0:
                losses["equivariance_shift"] = lw["equivariance_shift"] * torch.abs(driving_rp["shift"] - tr.warp_coordinates(trp["shift"])).mean()
            if lw["equivariance_affine"] != 0:
                value = _mm2(_inv2(driving_rp["affine"]), _mm2(tr.jacobian(trp["shift"]), trp["affine"]))
                if mp.get("revert_axis_swap", True):
                    value = value * torch.sign(value[:, :, 0:1, 0:1])
                eye = torch.eye(2, dtype=value.dtype, device=value.device).view(1, 1, 2, 2)
                losses["equivariance_affine"] = lw["equivariance_affine"] * torch.abs(eye - value).mean()
        flush_batches_tracked()
        return losses, gen

    __call__ = forward


class LFAETrainer:
    """The loop body of LFAE/train.py:96-104 - zero_grad, forward, sum of the loss terms, backward, Adam(betas=(0.5, 0.999)) - on
    the flat fused optimizer, with the epoch-milestone learning-rate schedule (train.py:59, MultiStepLR gamma 0.1) and checkpoints in
    the reference's format (train.py:136-160).  One process per GPU: `enable_data_parallel()` all-reduces the gradients (mean over
    ranks = what nn.DataParallel's replica mean does, train.py:99-100) overlapped with backward; BatchNorm statistics stay per rank."""

    def __init__(self, generator, region_predictor, bg_predictor, model_params, train_params, vgg=None):
        from .optim import FlatAdam as Adam
        self.generator, self.region_predictor, self.bg_predictor = generator, region_predictor, bg_predictor
        self.model = ReconstructionModel(region_predictor, bg_predictor, generator, model_params, train_params, vgg=vgg)
        self.train_params = train_params
        params = [p for net in (generator, region_predictor, bg_predictor) for p in net.parameters() if p.requires_grad]
        self.optimizer = Adam(params, lr=train_params["lr"], betas=(0.5, 0.999))
        self.milestones = list(train_params.get("epoch_milestones", []))
        self.base_lr = train_params["lr"]
        self.epoch, self.examples = 0, 0
        self._dp = None
        self._graphs = {}

    def to(self, device):
        for net in (self.generator, self.region_predictor, self.bg_predictor):
            net.to(device)
        self.model.to(device)
        return self

    def enable_data_parallel(self, bucket_bytes=64 << 20):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from .optim import GradAllReduce
            self._dp = GradAllReduce(self.optimizer, bucket_bytes=bucket_bytes)
            self._dp.sync_replicas()
        return self

    def draw_transform_noise(self, bs):
        """(theta noise (B, 2, 3), tps noise (B, 1, points^2) or None) drawn like Transform.__init__ draws them (model.py:90-100)."""
        tp = self.train_params["transform_params"]
        theta = torch.normal(mean=0, std=tp["sigma_affine"] * torch.ones([bs, 2, 3]))
        tps = (torch.normal(mean=0, std=tp["sigma_tps"] * torch.ones([bs, 1, tp["points_tps"] ** 2]))
               if tp.get("sigma_tps") is not None else None)
        return theta, tps

    def step_graphed(self, x, transform_noise=None):
        """The same step with forward + backward replayed as ONE hipGraph (opt-in; single process).  A step is ~2 200 launches: below ~32 pairs
        per GPU the host cannot issue them as fast as the GPU retires them (45 ms per step at 8 pairs for 30 ms of kernels) - and the
        reference's batch_size 100 over 8 GPUs is 12 pairs per GPU.  The first call for a batch shape runs two eager steps (filter packs,
        job tables, workspaces, constant grids come into being), captures the third into a graph on static input / noise buffers, and every later
        call copies its inputs and the step's transform noise (drawn on the host exactly as the eager step draws it) into those buffers and replays;
        the optimizer step (its bias correction changes per step) stays outside.  Returns the loss terms; `generated` tensors are the graph's
        static outputs (overwritten by the next replay)."""
        if self._dp is not None:
            raise RuntimeError("step_graphed is single-process (the gradient all-reduce is launched from autograd hooks)")
        bs = x["source"].shape[0]
        dev = x["source"].device
        key = (tuple(x["source"].shape), str(dev))
        g = self._graphs.get(key)
        noise = transform_noise if transform_noise is not None else self.draw_transform_noise(bs)
        if g is None:
            g = self._graphs[key] = {"warm": 0}
        if "graph" in g and g["buffers_epoch"] != P.buffers_epoch():
            # a filter pack or the scatter accumulator was (re)allocated since the capture (an eager / eval forward of another shape, a
            # new model, a larger batch): the graph holds raw addresses of the old buffers - capture again on the live ones
            for k in ("graph", "losses", "generated", "loss", "grads"):
                g.pop(k, None)
            g["warm"] = 1
        if "graph" not in g:
            if g["warm"] < 2:                          # eager warm-up steps (they are real training steps) - on a SIDE stream, as torch's
                g["warm"] += 1                         # whole-network capture recipe asks: the AccumulateGrad nodes must not be born on the default stream
                side = g.setdefault("side", torch.cuda.Stream())
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    res = self.step(x, transform_noise=noise)
                torch.cuda.current_stream().wait_stream(side)
                return res
            g["src"], g["drv"] = x["source"].clone(), x["driving"].clone()
            g["theta"] = noise[0].to(dev)
            g["tps"] = None if noise[1] is None else noise[1].to(dev)
            self.optimizer.zero_grad()
            A.repack_stale()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                losses, generated = self.model({"source": g["src"], "driving": g["drv"]}, transform_noise=(g["theta"], g["tps"]))
                loss = sum(v.mean() for v in losses.values())
                loss.backward()
            g["graph"], g["losses"], g["generated"], g["loss"] = graph, losses, generated, loss
            g["buffers_epoch"] = P.buffers_epoch()
            # the graph reads these through raw addresses: keep them alive for as long as it can be replayed
            g["held"] = ([v[2] for v in A._PACKS.values()], [v[1] for v in _PAD_BUFFERS.values()], dict(L._state(dev)))
            g["grads"] = [p.grad for grp in self.optimizer.param_groups for p in grp["params"]]      # the capture's gradient tensors
            graph.replay()                              # (capturing records the launches, it does not run them)
        else:
            g["src"].copy_(x["source"])
            g["drv"].copy_(x["driving"])
            g["theta"].copy_(noise[0])
            if g["tps"] is not None:
                g["tps"].copy_(noise[1])
            refresh_pad_buffers()
            A.repack_stale()
            # autograd does not run on a replay: hand every parameter the gradient tensor the captured backward writes
            for p, gr in zip((p for grp in self.optimizer.param_groups for p in grp["params"]), g["grads"]):
                p.grad = gr
            self.optimizer._staged = False
            g["graph"].replay()
        self.optimizer.step()
        self.examples += bs
        out = {k: v.detach() for k, v in g["losses"].items()}
        out["total"] = g["loss"].detach()
        return out, g["generated"]

    def step(self, x, transform_noise=None):
        """x: {'source': (B, 3, H, W), 'driving': (B, 3, H, W)} in [0, 1].  -> (loss terms (detached), generated)."""
        self.optimizer.zero_grad()
        if self._dp is not None:
            self._dp.prepare()
        A.repack_stale()          # the Winograd filters of the three trained networks, forward + data-gradient forms: one launch
        losses, generated = self.model(x, transform_noise=transform_noise)
        loss = sum(v.mean() for v in losses.values())
        loss.backward()
        if self._dp is not None:
            self._dp.finish()
        self.optimizer.step()
        self.examples += x["source"].shape[0] * (self._dp.world if self._dp is not None else 1)     # global count, like train.py's 'example'
        out = {k: v.detach() for k, v in losses.items()}
        out["total"] = loss.detach()
        return out, generated

    def end_epoch(self):
        """scheduler.step() of train.py:167: lr = base * 0.1 ** (milestones passed)."""
        self.epoch += 1
        lr = self.base_lr * (0.1 ** sum(1 for m in self.milestones if self.epoch >= m))
        for g in self.optimizer.param_groups:
            g["lr"] = lr
        return lr

    def state_dict(self):
        return {"example": self.examples, "epoch": self.epoch, "generator": self.generator.state_dict(),
                "bg_predictor": self.bg_predictor.state_dict(), "region_predictor": self.region_predictor.state_dict(),
                "optimizer": self.optimizer.state_dict()}

    def load_state_dict(self, ckpt, set_start=True):
        self.generator.load_state_dict(ckpt["generator"])
        self.region_predictor.load_state_dict(ckpt["region_predictor"])
        self.bg_predictor.load_state_dict(ckpt["bg_predictor"])
        if "optimizer" in ckpt:
            self.optimizer.load_state_dict(ckpt["optimizer"])
        if set_start:
            self.examples, self.epoch = int(ckpt.get("example", 0)), int(ckpt.get("epoch", 0))
            self.epoch -= 1
            self.end_epoch()


# ------------------------------------------------------------------------------------------------ data
class FramePairs(torch.utils.data.Dataset):
    """Training items of the reference's FramesDataset (LFAE/mug_dataset.py:60-140, is_train=True): two random frames of one video as
    {'source', 'driving'} (3, H, W) float32 in [0, 1], with the yaml's augmentation_params (horizontal flip, time flip = the two frames
    exchanged, one colour jitter for the pair).  `videos`: a list of frame-file lists, or a directory that is walked for folders of
    *.jpg|png frames (any depth: MUG's <subject>/<expression>/<take>, MHAD's / NATOPS' flat folders)."""

    def __init__(self, videos, frame_shape=128, horizontal_flip=True, time_flip=True, jitter=None, seed=None):
        import os
        if isinstance(videos, str):
            found = []
            for root, _, files in sorted(os.walk(videos)):
                frames = sorted(f for f in files if f.lower().endswith(("jpg", "jpeg", "png")))
                if len(frames) >= 2:
                    found.append([os.path.join(root, f) for f in frames])
            if not found:
                raise FileNotFoundError("no folders with >= 2 *.jpg|png frames under %r" % (videos,))
            videos = found
        self.videos, self.frame_shape = videos, frame_shape
        self.horizontal_flip, self.time_flip, self.jitter = horizontal_flip, time_flip, jitter
        self.seed = seed
        self._rng, self._rng_key = None, None

    def _generator(self):
        """The item's random source.  The reference draws from the global np.random, which torch's DataLoader re-seeds per worker
        and per epoch; a generator object copied into every forked worker would replay the same draws in all of them.  Here: one
        generator per (process, DataLoader worker seed) - the worker seed changes with the worker and with every new iterator -
        mixed with the dataset's own seed."""
        import os
        import numpy as np
        info = torch.utils.data.get_worker_info()
        key = (os.getpid(), None if info is None else info.seed)
        if self._rng is None or self._rng_key != key:
            mix = [] if self.seed is None else [int(self.seed)]
            if info is not None:
                mix.append(int(info.seed) % (1 << 63))
            self._rng = np.random.default_rng(mix if mix else None)
            self._rng_key = key
        return self._rng

    def __len__(self):
        return len(self.videos)

    def __getitem__(self, index):
        import numpy as np
        from .data import _rgb, color_jitter
        from .io_compat import INTER_AREA, imread, resize
        paths = self.videos[index]
        rng = self._generator()
        i, j = np.sort(rng.choice(len(paths), size=2, replace=True))         # replace=True: mug_dataset.py:94
        frames = [_rgb(imread(paths[i])), _rgb(imread(paths[j]))]
        if self.jitter:
            frames = color_jitter(frames, bright=self.jitter.get("brightness", 0.1), contrast=self.jitter.get("contrast", 0.1),
                                  sat=self.jitter.get("saturation", 0.1), hue=self.jitter.get("hue", 0.1))
        frames = [resize(np.asarray(f, np.float32), self.frame_shape, interpolation=INTER_AREA) / 255.0 for f in frames]
        if self.horizontal_flip and rng.random() < 0.5:
            frames = [f[:, ::-1] for f in frames]
        if self.time_flip and rng.random() < 0.5:
            frames = frames[::-1]
        src, drv = (torch.from_numpy(np.ascontiguousarray(np.transpose(f, (2, 0, 1)), dtype=np.float32)) for f in frames)
        return {"source": src, "driving": drv, "frame": [paths[i], paths[j]]}


def build_from_config(config):
    """(generator, region_predictor, bg_predictor) ParamTrees for a config/*.yaml dict (`model_params`), as LFAE/run_mug.py:96-109."""
    from .flow_diffusion import BGMotionPredictor, RegionPredictor
    from .generator import Generator
    mp = config["model_params"]
    gen = Generator(num_regions=mp["num_regions"], num_channels=mp["num_channels"], revert_axis_swap=mp.get("revert_axis_swap", True),
                    **mp["generator_params"])
    reg = RegionPredictor(num_regions=mp["num_regions"], num_channels=mp["num_channels"], estimate_affine=mp["estimate_affine"],      # (LFAE/run_mug.py:100 reads the key unconditionally)
                          **mp["region_predictor_params"])
    bgp = BGMotionPredictor(num_channels=mp["num_channels"], **mp["bg_predictor_params"])
    return gen, reg, bgp
