"""The frozen LFAE motion predictors of the DM training step, batched over all frames of a video batch:

  RegionPredictor        LFAE/modules/region_predictor.py:77-117   (a37)
  BGMotionPredictor      LFAE/modules/bg_motion_predictor.py:42-57  (a38)
  PixelwiseFlowPredictor LFAE/modules/pixelwise_flow_predictor.py:104-137 (a39)

The reference evaluates them once per frame inside a Python loop (video_flow_diffusion_model.py:124-137, with a
device->host copy for `torch.svd(covar.cpu())` in every iteration); here the N = B*T frames go through each network
in ONE pass.  All convolutions (the three hourglass networks, 98 % of the FLOPs) run on the native implicit-GEMM
kernels with eval-BatchNorm folded into the weights, channels-last, skip concatenations passed as two-source
convolutions.  What is left in torch are the few-KB closed-form pieces (soft-argmax moments, 2x2 inverses, Gaussian
heat-maps, the 11-way softmax blend) and the 2x2 SVD of the region covariances.  The reference takes that SVD on the host
(`torch.svd(covar.cpu())`, LAPACK xGESDD) and uses U * sqrt(S), so the sign convention of U matters; `svd2x2_sym_lapack`
replays xGESDD's 2x2 path (xGEBRD Householder -> xBDSQR/xLASV2 -> sort) in closed form on the device: same signs as LAPACK
on 800 k random + edge-case covariances (tests/test_svd2x2.py), and no device->host sync in the training step.
"""
import torch
import torch.nn.functional as F

from . import ops

BN_EPS = 1e-5


def _fold(tree, cprefix, nprefix, cin_pad=0):
    """conv + eval-BatchNorm folded and packed.  cin_pad: zero input channels appended to the filter so that it matches a
    zero-padded input buffer (a 3- / 6- / 44-channel image in a 16- / 16- / 64-wide CL buffer): the first convolution of an
    hourglass then qualifies for the Winograd / 32-channel fast schedules instead of the generic gather path (44 -> 128 at
    32x32 over 320 frames: 4.1 ms -> 0.4 ms in the training step)."""
    g = lambda k: tree.get(k).detach().float()
    a = g(nprefix + "weight") / torch.sqrt(g(nprefix + "running_var") + BN_EPS)
    b = g(nprefix + "bias") - g(nprefix + "running_mean") * a
    w = g(cprefix + "weight") * a.view(-1, 1, 1, 1)
    if cin_pad > w.shape[1]:
        w = F.pad(w, (0, 0, 0, 0, 0, cin_pad - w.shape[1]))
    w = w.contiguous()
    ww = ops.pack_wino_weight(w) if (w.shape[1] % 16 == 0 and w.shape[-1] == 3) else None     # Winograd form of the same filter
    return ops.pack_conv_weight(w), (g(cprefix + "bias") * a + b).contiguous(), w.shape[0], ww


class HourglassExec:
    """Hourglass / Encoder of LFAE/modules/util.py:153-214 on CL rows.  DownBlock2d = conv3x3+BN+ReLU+AvgPool2,
    UpBlock2d = nearest x2 + conv3x3+BN+ReLU; `cat([out, skip])` is never materialised."""

    def __init__(self, tree, prefix, num_blocks, decoder=True, in_pad_to=1):
        """in_pad_to: the caller passes its input rows zero-padded to a multiple of this many channels (_image_rows(pad_to=,
        full=True)); the first filter is padded to match."""
        self.tree, self.prefix, self.num_blocks, self.decoder, self.in_pad_to = tree, prefix, num_blocks, decoder, in_pad_to
        self._pk, self._sig = None, None

    def _packed(self):
        sig = (sum(p._version for p in self.tree.parameters()) + sum(b._version for b in self.tree.buffers()),
               next(self.tree.parameters()).device)
        if self._pk is None or self._sig != sig:
            pk = {"down": [], "up": []}
            for i in range(self.num_blocks):
                q = "%sencoder.down_blocks.%d." % (self.prefix, i)
                cin = self.tree.get(q + "conv.weight").shape[1]
                pad = (cin + self.in_pad_to - 1) // self.in_pad_to * self.in_pad_to if i == 0 else 0
                pk["down"].append(_fold(self.tree, q + "conv.", q + "norm.", cin_pad=pad))
            if self.decoder:
                for j in range(self.num_blocks):
                    q = "%sdecoder.up_blocks.%d." % (self.prefix, j)
                    pk["up"].append(_fold(self.tree, q + "conv.", q + "norm."))
            self._pk, self._sig = pk, sig
        return self._pk

    def encode(self, x_cl, n, h, w):
        """-> list of (rows, C) CL feature maps at h, h/2, ... (outs[0] = the input)."""
        pk = self._packed()
        outs = [(x_cl, h, w)]
        for wp, b, co, wino in pk["down"]:
            src, hh, ww = outs[-1]
            try:          # DownBlock2d: the 2x2 average pool taken in the convolution's epilogue (Winograd schedule)
                pooled = ops.conv2d_cl(src, wp, co, 3, 3, n, hh, ww, bias=b, act=ops.ACT_RELU, weight_wino=wino, pool2=True) \
                    if (wino is not None and hh % 2 == 0 and ww % 2 == 0) else None
            except ops.WinogradUnavailable:
                pooled = None
            if pooled is None:
                y = ops.conv2d_cl(src, wp, co, 3, 3, n, hh, ww, bias=b, act=ops.ACT_RELU, weight_wino=wino)
                pooled = ops.avgpool2_cl(y, n, hh, ww)
            outs.append((pooled, hh // 2, ww // 2))
        return outs

    def forward(self, x_cl, n, h, w):
        """-> (out, skip): the hourglass output is cat([out, skip]) with skip = the input rows."""
        pk = self._packed()
        outs = self.encode(x_cl, n, h, w)
        out, hh, ww = outs.pop()
        src1 = None
        for wp, b, co, wino in pk["up"]:
            ok = wino is not None and out.shape[1] % 16 == 0 and (src1 is None or src1.shape[1] % 16 == 0)
            out = ops.conv2d_cl(out, wp, co, 3, 3, n, hh, ww, src1=src1, bias=b, upsample=True, act=ops.ACT_RELU,
                                weight_wino=wino if ok else None)
            hh, ww = hh * 2, ww * 2
            src1 = outs.pop()[0]
        return out, src1


def _image_rows(x, pad_to=4, full=False):
    """(N, C, H, W) planar image -> CL rows (N*H*W, C) inside a buffer of row stride `pad_to`-aligned.
    full=True returns the whole zero-padded buffer (rows, ld) - for consumers whose filters are padded to match."""
    n, c, h, w = x.shape
    ld = (c + pad_to - 1) // pad_to * pad_to
    buf = torch.zeros(n * h * w, ld, dtype=torch.float32, device=x.device)
    view = buf[:, :c]
    ops.planar_to_cl(x.reshape(n, c, h * w).contiguous().float(), n, c, h * w, out=view)
    return buf if full else view


def _head_conv(tree, prefixes, out, skip, n, h, w, pad, planar=True):
    """The 7x7 head convolutions `prefixes` (same input cat([out, skip])) as ONE launch -> planar (N, sum of couts, H', W'),
    no activation.  The filters are stacked along the output axis and zero-padded to a multiple of 4 outputs: that is what
    lets the library take its 160x32 K-split schedule (float4 epilogue) instead of a 64-column tile for 1 / 10 / 11 outputs -
    at B*T = 320 frames the mask + occlusion heads were 2 x 1.96 ms, the region head 1.8 ms of the training step."""
    prefixes = (prefixes,) if isinstance(prefixes, str) else tuple(prefixes)
    cin = out.shape[1] + skip.shape[1]
    cache = tree.__dict__.setdefault("_lfdm_head_packs", {})
    sig = (sum(tree.get(q + "weight")._version + tree.get(q + "bias")._version for q in prefixes), out.device)
    hit = cache.get((prefixes, cin))
    if hit is None or hit[0] != sig:
        wt = torch.cat([tree.get(q + "weight").detach().float() for q in prefixes], dim=0)
        bias = torch.cat([tree.get(q + "bias").detach().float() for q in prefixes], dim=0)
        cout, k = wt.shape[0], wt.shape[-1]
        # input channels: the skip rows are a zero-padded image buffer; outputs: up to a multiple of 4
        wt = F.pad(wt, (0, 0, 0, 0, 0, cin - wt.shape[1], 0, -cout % 4))
        hit = (sig, ops.pack_conv_weight(wt.contiguous()), F.pad(bias, (0, -cout % 4)).contiguous(), cout, k)
        cache[(prefixes, cin)] = hit
    _, wp, bias, cout, k = hit
    coutp = bias.numel()
    y = ops.conv2d_cl(out, wp, coutp, k, k, n, h, w, src1=skip, bias=bias, pad=(pad, pad))
    if not planar:
        return y                                              # channels-last rows (N*H'*W', coutp)
    ho, wo = h + 2 * pad - k + 1, w + 2 * pad - k + 1
    return ops.cl_to_planar(y, n, coutp, ho * wo).view(n, coutp, ho, wo)[:, :cout]


def antialias_down(x, weight, scale):
    """AntiAliasInterpolation2d (util.py:217-264): depthwise Gaussian, then every (1/scale)-th pixel."""
    if scale == 1:
        return x
    ks = weight.shape[-1]
    ka = ks // 2
    kb = ka - 1 if ks % 2 == 0 else ka
    return ops.depthwise_down_planar(x.float(), weight.detach().float(), int(1 / scale), ka, kb)


def _sign1(x):
    """Fortran SIGN(1, x): +1 for x >= 0."""
    return torch.where(x < 0, -torch.ones_like(x), torch.ones_like(x))


def _lasv2(f, g, h):
    """LAPACK xLASV2: SVD of [[f, g], [0, h]] -> (ssmin, ssmax, snl, csl), vectorised; every branch of the routine is
    evaluated and selected with torch.where (few-KB tensors)."""
    one, zero = torch.ones_like(f), torch.zeros_like(f)
    fa, ha = f.abs(), h.abs()
    swap = ha > fa
    ft, ht = torch.where(swap, h, f), torch.where(swap, f, h)
    fa, ha = torch.where(swap, ha, fa), torch.where(swap, fa, ha)
    gt, ga = g, g.abs()
    diag = ga == 0
    eps = torch.finfo(f.dtype).eps / 2                      # xLAMCH('E')
    gbig = (~diag) & (ga > fa)
    ft_ = torch.where(ft == 0, one, ft)                     # guards for the lanes whose branch is not taken
    ga_ = torch.where(diag, one, ga)
    gt_ = torch.where(diag, one, gt)
    large = gbig & ((fa / ga_) < eps)
    d = fa - ha
    l = torch.where(d == fa, one, d / torch.where(fa == 0, one, fa))
    m = gt / ft_
    t = 2 - l
    mm, tt = m * m, t * t
    sq = torch.sqrt(tt + mm)
    r = torch.where(l == 0, m.abs(), torch.sqrt(l * l + mm))
    a = 0.5 * (sq + r)
    ssmin, ssmax = ha / a, fa * a
    d_ = torch.where(d == 0, one, d)
    t_tiny = torch.where(l == 0, 2 * _sign1(ft) * _sign1(gt), gt / (d_.abs() * _sign1(ft)) + m / t)
    rl = torch.where(r + l == 0, one, r + l)
    t_norm = (m / (sq + t) + m / rl) * (1 + a)
    t2 = torch.where(mm == 0, t_tiny, t_norm)
    l2 = torch.sqrt(t2 * t2 + 4)
    crt, srt = 2 / l2, t2 / l2
    clt = (crt + srt * m) / a
    slt = (ht / ft_) * srt / a
    clt, slt = torch.where(large, one, clt), torch.where(large, ht / gt_, slt)
    srt, crt = torch.where(large, one, srt), torch.where(large, ft / gt_, crt)
    ssmax = torch.where(large, ga, ssmax)
    ssmin = torch.where(large, torch.where(ha > 1, fa / (ga_ / torch.where(ha == 0, one, ha)), (fa / ga_) * ha), ssmin)
    clt, crt = torch.where(diag, one, clt), torch.where(diag, one, crt)
    slt, srt = torch.where(diag, zero, slt), torch.where(diag, zero, srt)
    ssmax, ssmin = torch.where(diag, fa, ssmax), torch.where(diag, ha, ssmin)
    csl, snl = torch.where(swap, srt, clt), torch.where(swap, crt, slt)
    return ssmin.abs(), ssmax.abs(), snl, csl


def svd2x2_sym_lapack(a, b, c):
    """(U, S) of the symmetric [[a, b], [b, c]] exactly as torch.svd / LAPACK xGESDD return them for a 2x2 input (path 5:
    xGEBRD = one Householder reflection H1 of the first column, xBDSQR on the upper-bidiagonal H1*A: a negligible
    off-diagonal splits, else xLASV2 rotates; singular values sorted by one selection-sort swap).  Only U's sign
    convention needs all this - the reference multiplies U by sqrt(S) (LFAE/modules/region_predictor.py:16-25)."""
    one, zero = torch.ones_like(a), torch.zeros_like(a)
    r = torch.sqrt(a * a + b * b)
    beta = -_sign1(a) * r
    noref = b == 0                                        # xLARFG: nothing to annihilate -> H1 = I
    safe = torch.where(noref, one, beta)
    h00, h01 = torch.where(noref, one, a / safe), torch.where(noref, zero, b / safe)
    h11 = torch.where(noref, one, -a / safe)
    d1 = torch.where(noref, a, beta)
    e1 = h00 * b + h01 * c
    d2 = h01 * b + h11 * c
    ssmin, ssmax, snl, csl = _lasv2(d1, e1, d2)
    eps = torch.finfo(a.dtype).eps / 2
    tol = max(10.0, min(100.0, eps ** (-0.125))) * eps
    ad1, ad2, ae = d1.abs(), d2.abs(), e1.abs()
    mu = ad2 * (ad1 / torch.where(ad1 + ae == 0, one, ad1 + ae))
    sminoa = torch.where(ad1 == 0, zero, torch.minimum(ad1, mu)) / (2.0 ** 0.5)
    split = ae <= torch.clamp(tol * sminoa, min=24 * torch.finfo(a.dtype).tiny)
    csl, snl = torch.where(split, one, csl), torch.where(split, zero, snl)
    s1, s2 = torch.where(split, ad1, ssmax), torch.where(split, ad2, ssmin)
    u00, u01 = h00 * csl + h01 * snl, -h00 * snl + h01 * csl
    u10, u11 = h01 * csl + h11 * snl, -h01 * snl + h11 * csl
    sw = s2 > s1
    u = torch.stack((torch.stack((torch.where(sw, u01, u00), torch.where(sw, u00, u01)), dim=-1),
                     torch.stack((torch.where(sw, u11, u10), torch.where(sw, u10, u11)), dim=-1)), dim=-2)
    return u, torch.stack((torch.where(sw, s2, s1), torch.where(sw, s1, s2)), dim=-1)


class RegionPredictorExec:
    def __init__(self, tree, num_blocks=5, temperature=0.1, scale_factor=0.25, pca_based=False, pad=3, estimate_affine=False):      # (the reference constructor's defaults, region_predictor.py:33-35)
        self.tree, self.temperature, self.scale_factor, self.pca_based, self.pad = tree, temperature, scale_factor, pca_based, pad
        self.regression = bool(estimate_affine) and not pca_based          # FOMM-like `jacobian` head (region_predictor.py:43-49, 98-108)
        self.hg = HourglassExec(tree, "predictor.", num_blocks, in_pad_to=32)      # 3-channel image in a 32-wide buffer (the
        #                                                     head's fast schedules want both concatenated sources in 32s)

    @torch.no_grad()
    def __call__(self, x):
        """x (N, 3, H, W) -> dict(shift (N,K,2), covar, affine (N,K,2,2), heatmap (N,K,h,w), u, d)."""
        if x.shape[0] > 65535 and self.pca_based:       # (grid limit of the fused statistics launch: more frames than any LFDM batch - slices)
            parts = [self(x_part) for x_part in x.split(32768)]
            return {key: torch.cat([q[key] for q in parts], dim=0) for key in parts[0]}
        if self.scale_factor != 1:
            x = antialias_down(x.float(), self.tree.get("down.weight"), self.scale_factor)
        n, _, h, w = x.shape
        out, skip = self.hg.forward(_image_rows(x, pad_to=32, full=True), n, h, w)
        k_regions = self.tree.get("regions.weight").shape[0]
        ho, wo = h + 2 * self.pad - 6, w + 2 * self.pad - 6                    # 7x7 head
        if not self.pca_based:
            # No LFDM yaml selects these two modes (all set pca_based: true); the reference accepts them.  The 7x7 heads run on the native
            # convolution (region and jacobian logits as ONE launch); the spatial softmax and the heat-map weighted sums over the few-KB
            # (N, K, h, w) maps are torch ops on the device.
            heads = _head_conv(self.tree, ("regions.", "jacobian.") if self.regression else "regions.", out, skip, n, h, w, self.pad)
            logits = heads[:, :k_regions]
            region = torch.softmax(logits.reshape(n, k_regions, -1) / self.temperature, dim=2).view(n, k_regions, ho, wo)
            ys, xs = torch.meshgrid(torch.linspace(-1, 1, ho, device=x.device), torch.linspace(-1, 1, wo, device=x.device), indexing="ij")
            grid = torch.stack((xs, ys), dim=-1).view(1, 1, ho, wo, 2)
            res = {"shift": (region.unsqueeze(-1) * grid).sum(dim=(2, 3)), "heatmap": region}
            if self.regression:
                jac = heads[:, k_regions:k_regions + 4].reshape(n, 1, 4, ho * wo)
                jac = (region.reshape(n, k_regions, 1, ho * wo) * jac).sum(dim=-1).view(n, k_regions, 2, 2)
                res["affine"] = jac
                res["covar"] = torch.matmul(jac, jac.transpose(-1, -2))
            return res
        if ho * wo > 4096:
            # frames above 256x256 at scale 0.25 (no LFDM yaml): the fused statistics launch holds one map per workgroup in LDS and stops at
            # 4096 pixels.  Head on the native convolution, softmax / centre / covariance as torch ops on the device (region_predictor.py:
            # 77-97), the 2x2 SVD with LAPACK's sign convention on the native kernel (:16-25 runs torch.svd on the host).
            logits = _head_conv(self.tree, "regions.", out, skip, n, h, w, self.pad)
            region = torch.softmax(logits.reshape(n, k_regions, -1) / self.temperature, dim=2).view(n, k_regions, ho, wo)
            ys, xs = torch.meshgrid(torch.linspace(-1, 1, ho, device=x.device), torch.linspace(-1, 1, wo, device=x.device), indexing="ij")
            grid = torch.stack((xs, ys), dim=-1).view(1, 1, ho, wo, 2)
            r = region.unsqueeze(-1)
            mean = (r * grid).sum(dim=(2, 3))
            sub = grid - mean.view(n, k_regions, 1, 1, 2)
            covar = ((sub.unsqueeze(-1) * sub.unsqueeze(-2)) * r.unsqueeze(-1)).sum(dim=(2, 3))
            cv = covar.reshape(-1, 4)
            u, sv = ops.svd2x2_sym(cv[:, 0], cv[:, 1], cv[:, 3])
            d = torch.diag_embed(sv ** 0.5)
            return {"shift": mean, "heatmap": region, "covar": covar, "affine": torch.matmul(u, d).view(n, k_regions, 2, 2), "u": u, "d": d}
        # one launch: spatial softmax, centre, covariance, U sqrt(S) with LAPACK's sign convention (region_predictor.py:16-25 does
        # torch.svd(covar.cpu()) per frame; svd2x2_sym_lapack above is the same closed form in tensor ops, kept for tests/test_svd2x2.py)
        rows = _head_conv(self.tree, "regions.", out, skip, n, h, w, self.pad, planar=False)
        return ops.lfae_region_stats(rows, n, k_regions, ho, wo, self.temperature)


class BGMotionPredictorExec:
    def __init__(self, tree, num_blocks=5, bg_type="zero"):
        if bg_type not in ("zero", "shift", "affine", "perspective"):
            raise ValueError("bg_type %r (the reference accepts 'zero', 'shift', 'affine', 'perspective')" % (bg_type,))
        self.tree, self.bg_type = tree, bg_type
        self.enc = None if bg_type == "zero" else HourglassExec(tree, "", num_blocks, decoder=False, in_pad_to=16)   # cat(source, driving): 6 channels

    @torch.no_grad()
    def __call__(self, source, driving):
        """(N,3,H,W) x2 -> (N,3,3) background transform (bg_motion_predictor.py:42-57): identity ('zero'), translation ('shift'), affine or
        perspective."""
        n, _, h, w = source.shape
        if self.bg_type == "zero":
            return torch.eye(3, device=source.device).unsqueeze(0).repeat(n, 1, 1)
        x = torch.cat((source, driving), dim=1).float()
        feats = self.enc.encode(_image_rows(x, pad_to=16, full=True), n, h, w)
        last, hh, ww = feats[-1]
        pooled = last.view(n, hh * ww, -1).mean(dim=1)
        pred = F.linear(pooled, self.tree.get("fc.weight"), self.tree.get("fc.bias"))
        from .params import bg_matrix
        return bg_matrix(pred, n, self.bg_type)


class PixelwiseFlowPredictorExec:
    """`tree` is the Generator (keys pixelwise_flow_predictor.*)."""
    MAX_FRAMES = 65535          # frames per library launch (grid limit); larger calls run in slices of whole videos

    def __init__(self, tree, num_regions, num_blocks=5, scale_factor=0.25, use_covar_heatmap=True,
                 use_deformed_source=True, revert_axis_swap=True, region_var=0.01):
        self.tree, self.k = tree, num_regions
        self.scale_factor, self.use_covar_heatmap = scale_factor, use_covar_heatmap
        self.use_deformed_source, self.revert_axis_swap, self.region_var = use_deformed_source, revert_axis_swap, region_var
        self.hg = HourglassExec(tree, "pixelwise_flow_predictor.hourglass.", num_blocks, in_pad_to=32)    # 44 channels in a 64-wide buffer

    @torch.no_grad()
    def __call__(self, source_image, driving, source, bg_params=None, frames=None):
        """frames=None: the reference's signature - source_image / source params carry one entry per driving frame (N).
        frames=T: source_image (B, ...) and source params (B, K, ...) are per VIDEO, the driving params / bg_params per frame
        (N = B*T, n = b*T + t): nothing is repeated T times (the down-sampled source image in particular)."""
        p = "pixelwise_flow_predictor."
        t_ = 1 if frames is None else frames
        if source_image.shape[0] * t_ > self.MAX_FRAMES and t_ <= self.MAX_FRAMES:
            # more frames than the library launches' grid limit (no LFDM batch comes close): slices of whole videos
            step = self.MAX_FRAMES // t_
            parts = []
            for lo in range(0, source_image.shape[0], step):
                vid, frm = slice(lo, lo + step), slice(lo * t_, (lo + step) * t_)
                parts.append(self(source_image[vid], {key: v[frm] for key, v in driving.items()}, {key: v[vid] for key, v in source.items()},
                                  None if bg_params is None else bg_params[frm], frames))
            return {key: torch.cat([q[key] for q in parts], dim=0) for key in parts[0]}
        if self.scale_factor != 1:
            source_image = antialias_down(source_image.float(), self.tree.get(p + "down.weight"), self.scale_factor)
        k = self.k
        if source_image.shape[1] == 3 and k <= 32:
            # two library launches around the hourglass instead of ~120 element-wise ATen launches over (N, K+1, h, w[, 2])
            t = 1 if frames is None else frames
            b, c, h, w = source_image.shape
            n = b * t
            if n > 65535 or h * w > 4096:
                raise ValueError("PixelwiseFlowPredictor: %d frames per video / %dx%d maps per call (the library launches take at most 65535 "
                                 "frames of at most 4096 pixels)" % (n, h, w))
            rows, sparse = ops.lfae_motion_inputs(source_image, driving, source, bg_params, t, region_var=self.region_var,
                                                  revert_axis_swap=self.revert_axis_swap, use_covar=self.use_covar_heatmap)
            if not self.use_deformed_source:
                # pixelwise_flow_predictor.py:116-119: the hourglass sees the K+1 heat-maps only (channels 0, 4, 8, ... of the launch's
                # [heat | deformed RGB] groups), in a zero-padded buffer of the width its first filter was padded to
                heat = rows.new_zeros(rows.shape[0], (k + 1 + 31) // 32 * 32)
                heat[:, :k + 1] = rows[:, 0:4 * (k + 1):4]
                rows = heat
            out, skip = self.hg.forward(rows, n, h, w)
            has_occ = self.tree.has(p + "occlusion.weight")
            heads = _head_conv(self.tree, (p + "mask.", p + "occlusion.") if has_occ else (p + "mask.",), out, skip, n, h, w, 3,
                               planar=False)
            flow, occ = ops.lfae_motion_combine(heads, sparse, has_occ)
            res = {"optical_flow": flow}
            if has_occ:
                res["occlusion_map"] = occ
            return res
        # (a pure-ATen formulation of this head / tail - F.grid_sample and element-wise tensors over (N, K+1, h, w[, 2]) - used to stand
        #  here for other configurations: a silent non-native route on the product path.  No LFDM configuration reaches it.)
        raise NotImplementedError("PixelwiseFlowPredictor: RGB sources and num_regions <= 32 are built; got %d channels, %d regions"
                                  % (source_image.shape[1], k))
