"""Host-side mirror of the reference `Unet3D` (DM/modules/video_flow_diffusion.py:368-588):
same constructor arguments, same state-dict keys, same `forward` / `forward_with_cond_scale`
signatures - but the forward pass is a static launch plan over the HIP kernels of
liblfdm_hip.so (channels-last activations, weights repacked once, skip concatenation /
residual adds / GroupNorm affine fused into kernel arguments).  No torch compute op runs on the
activation path; torch only provides device memory and the stream.
"""
import os

import torch
from torch import nn

from . import ops
from .params import ParamTree, build_tree, unet_spec, weights_epoch

BERT_MODEL_DIM = 768
# Default-on fusions keep ONE switch each for A/B profiling and for the bit-compare tests (LFDM_x=0 -> the separate launches):
_LOWRES_ATTN = os.environ.get("LFDM_LOWRES_ATTN", "1") != "0"       # one-launch attention blocks at <= 64 pixels per frame
# measured (tools/bench_attn_lowres.py, profiles/r03_f_bench_attn_lowres.txt): the one-launch kernels win at <= 64 pixels per frame
# (4x4: 19 vs 31 us linear, 19 vs 26 temporal; 8x8: 33 vs 35 / 23 vs 29) and lose at 16x16 (69 vs 64 / 63 vs 52: 2048 workgroups of one
# head each re-read the frame's rows eight times and hold 78-110 KB of LDS)
_LOWRES_MAX_HW = 64
# Split-K Winograd convolutions reduce their slabs inside the launch (lfdm_conv_params.tile_counters: fence-free hand-off, conv_wino.hip FUSE):
# no conv_splitk_reduce launch behind the 16x16 / 8x8 / 4x4 convolutions of a B = 1 step.  LFDM_WINO_FUSE_REDUCE=0: the separate reduce pass.
_WINO_FUSE_REDUCE = os.environ.get("LFDM_WINO_FUSE_REDUCE", "1") != "0"
# to_out + the residual add inside the fused temporal-attention launch at C = 64 (ops.temporal_attention_fused_out_cl); LFDM_TATTN_OUT=0: separate
_TATTN_OUT = os.environ.get("LFDM_TATTN_OUT", "1") != "0"
_LINATTN_OUT = os.environ.get("LFDM_LINATTN_OUT", "1") != "0"      # to_out + bias + residual inside the fused linear attention's output pass at C = 64
_RES_GN_MAX_ROWS = 65536         # (the pointwise schedule takes a res_gn launch up to this many rows: the level-0 res_conv of a B = 1 step included, -0.7 ms)
_RES_GN = os.environ.get("LFDM_RES_GN", "1") != "0"                # block2's GroupNorm + SiLU inside the res_conv launch (blocks that change their channel count)
_HEADS_GN = os.environ.get("LFDM_HEADS_GN", "1") != "0"            # the heads block's last GroupNorm + SiLU inside the heads kernel
_HEADS_FOLD = os.environ.get("LFDM_HEADS_FOLD", "1") != "0"      # the output heads' res_conv folded into the 1x1 heads (exact by linearity)
# Built, measured slower and REMOVED in round 6 (records in HISTORY.md, rounds 1-5): res_conv on a second stream (LFDM_RES_STREAM), GroupNorm
# straight from the raw split-K slabs (LFDM_GN_SPLITK) and its chip-wide cooperative form (LFDM_GN_COOP), the in-launch slab reduction on
# the KSW schedule (LFDM_KSW_FUSE_REDUCE), block1's GroupNorm inside block2's convolution (LFDM_GN_FUSE), the channel-streaming fused
# temporal attention at C >= 128 (LFDM_TATTN_WIDE).


def prob_mask_like(shape, prob, device):
    """Reference prob_mask_like (:55-61): consumes the RNG only for 0 < prob < 1."""
    if prob == 1:
        return torch.ones(shape, device=device, dtype=torch.bool)
    if prob == 0:
        return torch.zeros(shape, device=device, dtype=torch.bool)
    return torch.zeros(shape, device=device).float().uniform_(0, 1) < prob


def rel_pos_bias_table(emb_weight, n, num_buckets=32, max_distance=32):
    """RelativePositionBias (:72-111) evaluated once per (weights, n): (heads, n, n).
    Input independent, so it is precomputed at plan time instead of on every forward."""
    import math
    dev = emb_weight.device
    pos = torch.arange(n, device=dev)
    rel = pos[None, :] - pos[:, None]
    neg = -rel
    half = num_buckets // 2
    ret = (neg < 0).long() * half
    a = neg.abs()
    max_exact = half // 2
    large = max_exact + (torch.log(a.float() / max_exact) / math.log(max_distance / max_exact)
                         * (half - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, half - 1))
    bucket = ret + torch.where(a < max_exact, a, large)
    return emb_weight[bucket].permute(2, 0, 1).contiguous()


class Unet3D(ParamTree):
    def __init__(self, dim, cond_dim=None, out_grid_dim=2, out_conf_dim=1, dim_mults=(1, 2, 4, 8),
                 channels=3, attn_heads=8, attn_dim_head=32, use_bert_text_cond=False, init_dim=None,
                 init_kernel_size=7, use_sparse_linear_attn=True, resnet_groups=8,
                 use_final_activation=False, learn_null_cond=False, use_deconv=True,
                 padding_mode="zeros"):
        super().__init__()
        if attn_heads != 8 or attn_dim_head != 32 or resnet_groups != 8 or not use_sparse_linear_attn \
                or use_final_activation or init_dim not in (None, dim) or init_kernel_size != 7:
            raise NotImplementedError("Unet3D: only the configuration the LFDM scripts use is built "
                                      "(heads=8, dim_head=32, groups=8, sparse linear attention)")
        self.null_cond_mask = None
        self.channels = channels
        self.dim = dim
        self.dim_mults = tuple(dim_mults)
        self.out_grid_dim, self.out_conf_dim = out_grid_dim, out_conf_dim
        self.has_cond = (cond_dim is not None) or use_bert_text_cond
        self.cond_dim = BERT_MODEL_DIM if use_bert_text_cond else cond_dim
        self.learn_null_cond = learn_null_cond
        self.use_deconv = use_deconv
        self.padding_mode = padding_mode
        if not use_deconv and padding_mode not in ("zeros", "reflect"):
            raise NotImplementedError("padding_mode %r" % padding_mode)
        spec = unet_spec(dim=dim, dim_mults=self.dim_mults, channels=channels, out_grid_dim=out_grid_dim,
                         out_conf_dim=out_conf_dim, cond_dim=self.cond_dim or 0,
                         learn_null_cond=learn_null_cond and self.has_cond, use_deconv=use_deconv)
        build_tree(self, spec)
        if self.has_cond and not learn_null_cond:
            # a plain tensor in the reference too (not in the state dict, :440)
            self.null_cond_emb = torch.zeros(1, self.cond_dim)
        self._pk = None
        self._pk_sig = None
        self._bufs = {}
        self._buf_gen = 0

    # ------------------------------------------------------------------ plumbing
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        if isinstance(getattr(self, "null_cond_emb", None), torch.Tensor) and not isinstance(
                self.null_cond_emb, nn.Parameter):
            self.null_cond_emb = fn(self.null_cond_emb)
        self._pk = None
        self._bufs = {}
        self._buf_gen = getattr(self, "_buf_gen", 0) + 1
        return out

    def _signature(self):
        """Identity of the weights the pack was built from.  `_version` sees every torch-side write (copy_, optimizer
        foreach ops, load_state_dict); writes through raw pointers (FlatAdam's fused HIP step) are invisible to it, so
        they bump `params.weights_epoch()` instead."""
        sig = 0
        for p in self.parameters():
            sig += p._version
        return (sig, weights_epoch(), next(self.parameters()).device)

    def _buf(self, name, rows, ch, dtype=torch.float32):
        need = rows * ch
        cur = self._bufs.get(name)
        dev = next(self.parameters()).device
        if cur is None or cur.numel() < need or cur.device != dev or cur.dtype != dtype:
            cur = torch.empty(need, dtype=dtype, device=dev)
            self._bufs[name] = cur
            # a captured hipGraph holds raw pointers into these arenas: a (re)allocation invalidates every plan built
            # on the old ones (diffusion._plans keys carry this generation)
            self._buf_gen += 1
        return cur[:need].view(rows, ch)

    @property
    def levels(self):
        dims = [self.dim] + [self.dim * m for m in self.dim_mults]
        return list(zip(dims[:-1], dims[1:]))

    # ------------------------------------------------------------------ weight packing
    def packed(self):
        sig = self._signature()
        if self._pk is not None and self._pk_sig == sig:
            return self._pk
        with torch.no_grad():
            self._pk = self._pack()
        self._pk_sig = sig
        return self._pk

    def _pack(self):
        g = lambda k: self.get(k).detach().float().contiguous()
        pk = {}

        def block(prefix):
            pk[prefix + "proj.w"] = ops.pack_conv_weight(g(prefix + "proj.weight"))
            wp = g(prefix + "proj.weight")
            # the same 3x3 filter in Winograd F(2x2,3x3) form: the library picks the 16/36-multiplication schedule
            pk[prefix + "proj.ww"] = ops.pack_wino_weight(wp) if wp.shape[1] % 16 == 0 else None
            pk[prefix + "proj.b"] = g(prefix + "proj.bias")
            pk[prefix + "norm.w"] = g(prefix + "norm.weight")
            pk[prefix + "norm.b"] = g(prefix + "norm.bias")

        cond_w, cond_b, off = [], [], 0

        def resblock(prefix):
            nonlocal off
            block(prefix + "block1.")
            block(prefix + "block2.")
            if self.has(prefix + "res_conv.weight"):
                pk[prefix + "res.w"] = ops.pack_conv_weight(g(prefix + "res_conv.weight"))
                pk[prefix + "res.b"] = g(prefix + "res_conv.bias")
            if self.has(prefix + "mlp.1.weight"):
                w = g(prefix + "mlp.1.weight")
                cond_w.append(w)
                cond_b.append(g(prefix + "mlp.1.bias"))
                pk[prefix + "ss_off"] = off
                off += w.shape[0]

        def temporal(prefix):
            wq, gam = g(prefix + "fn.fn.fn.to_qkv.weight"), g(prefix + "fn.norm.gamma")
            # LayerNorm + to_qkv + attention as ONE kernel (qkv never materialised): W' = W * gamma row-major, for the finest level's
            # wave-per-sequence kernel and the low-resolution levels' workgroup-per-(sequence, head) kernel
            pk[prefix + "qkv.wf"] = (wq.reshape(wq.shape[0], -1) * gam.reshape(1, -1)).contiguous()
            pk[prefix + "qkv.w"], pk[prefix + "qkv.wsum"] = ops.pack_ln_conv_weight(wq, gam)
            pk[prefix + "out.w"] = ops.pack_conv_weight(g(prefix + "fn.fn.fn.to_out.weight"))
            if wq.shape[1] == 64:      # the one-launch block at C = 64: both weights in MFMA-operand order
                pk[prefix + "qkv.wp"], pk[prefix + "out.wp"] = ops.pack_tattn_weights(pk[prefix + "qkv.wf"], g(prefix + "fn.fn.fn.to_out.weight").reshape(-1, 256))

        def spatial_linear(prefix):
            wq, gam = g(prefix + "fn.fn.to_qkv.weight"), g(prefix + "fn.norm.gamma")
            pk[prefix + "qkv.wf"] = (wq.reshape(wq.shape[0], -1) * gam.reshape(1, -1)).contiguous()      # (see temporal)
            if wq.shape[1] == 64:      # the fused three-launch form at C = 64: weight fragments in MFMA-operand order
                pk[prefix + "qkv.wp"] = ops.pack_linattn_weights(pk[prefix + "qkv.wf"])
                pk[prefix + "out.wp"] = ops.pack_linattn_out_weight(g(prefix + "fn.fn.to_out.weight").reshape(64, 256))
            pk[prefix + "qkv.w"], pk[prefix + "qkv.wsum"] = ops.pack_ln_conv_weight(wq, gam)
            pk[prefix + "out.w"] = ops.pack_conv_weight(g(prefix + "fn.fn.to_out.weight"))
            pk[prefix + "out.b"] = g(prefix + "fn.fn.to_out.bias")

        w0 = g("init_conv.weight")                      # (dim, 3+256, 1, 7, 7)
        n_dyn = self.channels - 256 if self.channels > 256 else self.channels
        pk["n_dyn"] = n_dyn
        pk["init.dyn_w"] = ops.pack_planar_in_weight(w0[:, :n_dyn].contiguous())
        if self.channels > n_dyn:
            pk["init.fea_w"] = ops.pack_conv_weight(w0[:, n_dyn:].contiguous())
        pk["init.b"] = g("init_conv.bias")
        temporal("init_temporal_attn.")
        pk["time.w1"], pk["time.b1"] = g("time_mlp.1.weight"), g("time_mlp.1.bias")
        pk["time.w3"], pk["time.b3"] = g("time_mlp.3.weight"), g("time_mlp.3.bias")
        pk["time.freqs"] = ops.sinusoidal_freqs(self.dim, w0.device)
        nl = len(self.levels)
        for lvl in range(nl):
            p = "downs.%d." % lvl
            resblock(p + "0.")
            resblock(p + "1.")
            spatial_linear(p + "2.")
            temporal(p + "3.")
            if lvl < nl - 1:
                pk[p + "4.w"] = ops.pack_conv_weight(g(p + "4.weight"))
                pk[p + "4.b"] = g(p + "4.bias")
        resblock("mid_block1.")
        pk["mid_spatial_attn.qkv.w"], pk["mid_spatial_attn.qkv.wsum"] = ops.pack_ln_conv_weight(
            g("mid_spatial_attn.fn.fn.fn.to_qkv.weight"), g("mid_spatial_attn.fn.norm.gamma"))
        pk["mid_spatial_attn.qkv.wf"] = (g("mid_spatial_attn.fn.fn.fn.to_qkv.weight").reshape(768, -1)
                                         * g("mid_spatial_attn.fn.norm.gamma").reshape(1, -1)).contiguous()
        pk["mid_spatial_attn.out.w"] = ops.pack_conv_weight(g("mid_spatial_attn.fn.fn.fn.to_out.weight"))
        temporal("mid_temporal_attn.")
        resblock("mid_block2.")
        for lvl in range(nl):
            p = "ups.%d." % lvl
            resblock(p + "0.")
            resblock(p + "1.")
            spatial_linear(p + "2.")
            temporal(p + "3.")
            if lvl < nl - 1:
                if self.use_deconv:
                    pk[p + "4.packs"] = ops.pack_deconv4_weight(g(p + "4.weight"))
                    pk[p + "4.b"] = g(p + "4.bias")
                else:
                    pk[p + "4.w"] = ops.pack_conv_weight(g(p + "4.1.weight"))
                    pk[p + "4.b"] = g(p + "4.1.bias")
        for head in ("final_conv.", "occlusion_map."):
            resblock(head + "0.")
            pk[head + "1.w"] = g(head + "1.weight").reshape(-1, self.dim).contiguous()
            pk[head + "1.b"] = g(head + "1.bias")
        if self.dim % 32 == 0 and self.has("final_conv.0.res_conv.weight"):      # the two heads as one 2*dim-channel block
            hp = ("final_conv.0.", "occlusion_map.0.")
            cat = lambda k: torch.cat([g(h + k) for h in hp], dim=0).contiguous()
            w1 = cat("block1.proj.weight")
            pk["heads.block1.w"], pk["heads.block1.ww"] = ops.pack_conv_weight(w1), ops.pack_wino_weight(w1)
            pk["heads.block1.b"] = cat("block1.proj.bias")
            pk["heads.block2.ww"] = ops.pack_wino_weight_grouped([g(h + "block2.proj.weight")[:, :, 0] for h in hp])
            pk["heads.block2.b"] = cat("block2.proj.bias")
            pk["heads.res.w"], pk["heads.res.b"] = ops.pack_conv_weight(cat("res_conv.weight")), cat("res_conv.bias")
            for i in (1, 2):
                pk["heads.norm%d.w" % i], pk["heads.norm%d.b" % i] = cat("block%d.norm.weight" % i), cat("block%d.norm.bias" % i)
            # res_conv folded into the (linear) 1x1 heads: W1 (h + Wres u + bres) + b1 = W1 h + (W1 Wres) u + (W1 bres + b1), u = cat(x, r)
            wf, wo = pk["final_conv.1.w"].double(), pk["occlusion_map.1.w"].double()                    # (2, dim), (1, dim)
            rf, ro = g(hp[0] + "res_conv.weight").reshape(self.dim, -1).double(), g(hp[1] + "res_conv.weight").reshape(self.dim, -1).double()
            pk["heads.fold.w"] = torch.cat((wf @ rf, wo @ ro), dim=0).float().contiguous()                # (3, 2 * dim)
            pk["heads.fold.bf"] = (pk["final_conv.1.b"].double() + wf @ g(hp[0] + "res_conv.bias").double()).float().contiguous()
            pk["heads.fold.bo"] = (pk["occlusion_map.1.b"].double() + wo @ g(hp[1] + "res_conv.bias").double()).float().contiguous()
        pk["cond.w"] = torch.cat(cond_w, dim=0).contiguous()     # (sum 2C, time_dim + cond_dim)
        pk["cond.b"] = torch.cat(cond_b, dim=0).contiguous()
        pk["cond.n"] = off
        pk["rel_emb"] = g("time_rel_pos_bias.relative_attention_bias.weight")
        pk["freqs"] = g("init_temporal_attn.fn.fn.fn.rotary_emb.freqs")
        pk["tables"] = {}
        return pk

    def _tables(self, pk, frames):
        t = pk["tables"].get(frames)
        if t is None:
            bias = rel_pos_bias_table(pk["rel_emb"], frames)
            ang = torch.arange(frames, device=bias.device).float()[:, None] * pk["freqs"][None, :]
            t = (bias, ang.cos().contiguous(), ang.sin().contiguous())
            pk["tables"][frames] = t
        return t

    # ------------------------------------------------------------------ conditioning
    def time_embedding(self, pk, t_dev, batch, t_stride=1):
        """SinusoidalPosEmb + time_mlp (:141-153, :423-428); t_dev int32 on device."""
        e = ops.sinusoidal(t_dev, pk["time.freqs"], batch, self.dim, t_stride=t_stride)
        e = ops.linear_small(e, pk["time.w1"], pk["time.b1"], act_out=ops.ACT_GELU)
        return ops.linear_small(e, pk["time.w3"], pk["time.b3"])

    def merge_cond(self, cond, null_mask):
        """where(null_mask, null_cond_emb, cond) (:556-561)."""
        null = self.null_cond_emb.to(cond.device, cond.dtype)
        return torch.where(null_mask.view(-1, 1), null, cond).contiguous()

    def cond_scale_shift(self, pk, temb, cond):
        """All ResnetBlock.mlp projections in one launch: Linear(SiLU(cat(temb, cond)))."""
        tc = torch.cat((temb, cond), dim=-1).contiguous()
        return ops.linear_small(tc, pk["cond.w"], pk["cond.b"], act_in=ops.ACT_SILU)

    def cond_tables(self, pk, temb_steps, cond):
        """Split form for graph replay: (per-step part, per-sample part incl. bias)."""
        td = temb_steps.shape[1]
        w = pk["cond.w"]
        step_part = ops.linear_small(temb_steps, w[:, :td], None, act_in=ops.ACT_SILU)
        sample_part = ops.linear_small(cond, w[:, td:], pk["cond.b"], act_in=ops.ACT_SILU)
        return step_part, sample_part

    # ------------------------------------------------------------------ building blocks
    def _conv(self, src0, w, cout, k, n_img, s, *, src1=None, bias=None, residual=None, out=None, gn=None,
              scratch="splitk", ww=None, **kw):
        # (measured and removed: res_conv on a second stream - slower; split-K slabs reduced inside the launch with a release / acquire
        #  fence pair, KSW schedule - neutral; the fence-free form on the Winograd schedule is `_WINO_FUSE_REDUCE` below)
        """One lfdm_conv2d_cl_f32 launch; tile shape / split-K come from the library's plan.
        gn = (batch,) asks for fused GroupNorm statistics; then returns (out, (partial, nchunk) or None)."""
        if ((k == 1 or (k == 4 and kw.get("stride") == 2)) and self._pk is not None and w.dim() == 3 and w.shape[1] % 32 == 0 and
                not kw.get("deconv4")):
            # 1x1 projections: the operand-order pack for the pointwise schedule, built once per weight pack
            cache = self._pk.setdefault("_pw", {})
            wpw = cache.get(w.data_ptr())
            if wpw is None:
                wpw = cache[w.data_ptr()] = ops.pack_pw_weight(w)
            kw["weight_pw"] = wpw
        p, y = ops.conv_params(src0, w, cout, k, k, n_img, s, s, src1=src1, bias=bias, residual=residual,
                               out=out, weight_wino=ww if (src1 is None or src0.shape[1] % 16 == 0) else None, **kw)
        coutp = p.coutp
        sched = ops.conv_schedule(p) if (_WINO_FUSE_REDUCE and kw.get("deconv4") is None) else -1
        # (KSW: fused statistics need the group inside a 32-column tile - wider groups keep the reduce pass, whose statistics take any width)
        if sched == 2:
            counters = self._tile_counters(src0.device)
            p.tile_counters, p.tile_counters_len = counters.data_ptr(), counters.numel()
        if gn is not None:
            p.gn_partial = 1      # (placeholder: "fused statistics wanted" changes the plan - schedules 3 / 4 have none; include/lfdm_hip.h)
        tile_rows, ksplit = ops.conv_plan(p)
        # (sized while the placeholder is still set: the slab buffer must belong to the plan the launch will re-derive with the real
        # gn_partial - without it schedules 3 / 4 would plan ksplit = 1 and size it 0)
        # (asked for ksplit = 1 plans too: the balanced Winograd launch halves the K range of 128 of a launch's 640 jobs and needs slabs for those)
        partial_floats = ops.conv_partial_floats(p) if (ksplit > 1 or sched == 2) else 0
        p.gn_partial = None
        m = n_img * p.hq * p.wq
        if partial_floats > 0:
            part = self._buf(scratch, 1, partial_floats)      # slabs (+ LayerNorm row statistics)
            p.partial = part.data_ptr()
        stats = None
        if gn is not None:
            batch, groups = gn[0], (gn[1] if len(gn) > 1 else 8)
            pixels = m // batch
            cg = cout // groups
            fused = ksplit > 1 and tile_rows != 16                        # slabs reduced inside the launch (KSW: 160-row tiles, Winograd: 128)
            # the fused Winograd form always runs 32-column workgroups: a group wider than that (512 channels / 8) gets one chunk slot per
            # column part (conv_wino.hip)
            parts = cg // 32 if (fused and tile_rows == 128 and cg > 32 and cg % 32 == 0) else 1
            in_tile = (ksplit == 1 or fused) and (32 % cg == 0 or parts > 1)   # statistics from the conv epilogue
            in_reduce = ksplit > 1 and not fused and 256 % (coutp // 4) == 0 and coutp == cout    # ... from the split-K reduce pass (any group width)
            if pixels % tile_rows == 0 and cg % 4 == 0 and (in_tile or in_reduce):
                nchunk = pixels // tile_rows * (parts if in_tile else 1)
                stats = (self._buf("gn.partial", batch * nchunk, 2 * groups), nchunk)
                p.gn_partial, p.gn_groups, p.gn_pixels = stats[0].data_ptr(), groups, pixels
        ops.conv_launch(p)
        return (y, stats) if gn is not None else y

    def _tile_counters(self, dev):
        """Ticket words of the in-launch split-K reduction: zeroed once, every launch leaves them zeroed (launches of a step are serialised)."""
        cur = getattr(self, "_tile_cnt", None)
        if cur is None or cur.device != torch.device(dev):
            cur = self._tile_cnt = torch.zeros(8192, dtype=torch.int32, device=dev)
            self._buf_gen += 1          # (a captured graph holds this pointer)
        return cur

    def _gn(self, x, batch, gamma, beta, stats, groups=8, **kw):
        gws = self._buf("gn.ws", batch, 256 * 128 + 2 * 1024)
        if stats is not None:
            return ops.groupnorm_apply_cl(x, batch, gamma, beta, stats[0], stats[1], out=x, ws=gws, groups=groups, **kw)
        return ops.groupnorm_silu_cl(x, batch, gamma, beta, out=x, ws=gws, groups=groups, **kw)

    def _resblock(self, pk, prefix, x, skip, batch, frames, s, ss, cout, outname):
        n_img, rows = batch * frames, batch * frames * s * s
        h1 = self._buf("rb.h1", rows, cout)
        _, st = self._conv(x, pk[prefix + "block1.proj.w"], cout, 3, n_img, s, src1=skip,
                           bias=pk[prefix + "block1.proj.b"], out=h1, gn=(batch,), ww=pk[prefix + "block1.proj.ww"])
        sshift = None
        if ss is not None and (prefix + "ss_off") in pk:
            o = pk[prefix + "ss_off"]
            sshift = ss[:, o:o + 2 * cout]
        self._gn(h1, batch, pk[prefix + "block1.norm.w"], pk[prefix + "block1.norm.b"], st, scale_shift=sshift)
        out = self._buf(outname, rows, cout)
        _, st = self._conv(h1, pk[prefix + "block2.proj.w"], cout, 3, n_img, s, bias=pk[prefix + "block2.proj.b"],
                           out=out, gn=(batch,), ww=pk[prefix + "block2.proj.ww"])
        has_res = (prefix + "res.w") in pk
        if has_res and _RES_GN and st is not None and rows <= _RES_GN_MAX_ROWS and (rows // batch) % 32 == 0 and cout % 32 == 0:
            # h + res_conv(x) with block2's GroupNorm + SiLU applied to the RAW `out` in the res_conv launch's epilogue (pointwise schedule;
            # lfdm_conv_params.res_gn_*): no GroupNorm launch for the blocks that change their channel count
            res_gn = dict(partial=st[0], nchunk=st[1], pixels=rows // batch, gamma=pk[prefix + "block2.norm.w"], beta=pk[prefix + "block2.norm.b"], groups=8)
            try:
                return self._conv(x, pk[prefix + "res.w"], cout, 1, n_img, s, src1=skip, bias=pk[prefix + "res.b"], residual=out, out=out,
                                  res_gn=res_gn)
            except RuntimeError:
                pass          # (a geometry the pointwise schedule does not take - M > 16 384 rows: the separate GroupNorm launch below)
        self._gn(out, batch, pk[prefix + "block2.norm.w"], pk[prefix + "block2.norm.b"], st,
                 residual=None if has_res else x)
        if has_res:
            self._conv(x, pk[prefix + "res.w"], cout, 1, n_img, s, src1=skip, bias=pk[prefix + "res.b"],
                       residual=out, out=out)
        return out

    def _attn_common(self, pk, prefix, x, n_img, s, c):
        """PreNorm + to_qkv as ONE kernel: the channel LayerNorm is folded into the 1x1 projection
        (lfdm_conv_params.ln_wsum), so the normalised tensor is never written."""
        rows = n_img * s * s
        qkv = self._buf("at.qkv", rows, 768)
        self._conv(x, pk[prefix + "qkv.w"], 768, 1, n_img, s, out=qkv, ln_wsum=pk[prefix + "qkv.wsum"])
        return qkv, self._buf("at.o", rows, 256)

    def _temporal_attn(self, pk, prefix, x, batch, frames, s, c, outname, tables, focus=None):
        bias, cos, sin = tables
        if focus is not None and any(focus):
            # focus_present_mask (Attention.forward :313-317 / :342-352): a focused sample attends only to itself - its softmax row is
            # exactly one-hot, so its attention output IS its value rows: to_out runs on the V columns of the LayerNorm-folded
            # projection for those samples, on the attention output for the others (library launches on row ranges; never captured)
            qkv, att = self._attn_common(pk, prefix, x, batch * frames, s, c)
            if not all(focus):
                ops.attention_cl(qkv, batch, frames, s * s, 0, bias=bias, rot_cos=cos, rot_sin=sin, out=att)
            out = self._buf(outname, x.shape[0], c)
            rows = frames * s * s
            for b in range(batch):
                sl = slice(b * rows, (b + 1) * rows)
                src = qkv[sl, 512:768] if focus[b] else att[sl]
                self._conv(src, pk[prefix + "out.w"], c, 1, frames, s, residual=x[sl], out=out[sl])
            return out
        if c == 64 and frames <= 64 and _TATTN_OUT:      # the whole block (LayerNorm, to_qkv, attention, to_out, residual) in one launch
            return ops.temporal_attention_fused_out_cl(x, pk[prefix + "qkv.wp"], pk[prefix + "out.wp"], batch, frames, s * s, bias=bias,
                                                       rot_cos=cos, rot_sin=sin, out=self._buf(outname, x.shape[0], c))
        if c == 64 and frames <= 64:
            att = self._buf("at.o", x.shape[0], 256)
            ops.temporal_attention_fused_cl(x, pk[prefix + "qkv.wf"], batch, frames, s * s, bias=bias, rot_cos=cos,
                                            rot_sin=sin, out=att)
        elif _LOWRES_ATTN and c % 64 == 0 and frames <= 64 and s * s <= _LOWRES_MAX_HW:      # one launch: workgroup = (pixel sequence, head)
            att = self._buf("at.o", x.shape[0], 256)
            ops.attention_lowres_cl(x, pk[prefix + "qkv.wf"], pk[prefix + "qkv.wsum"], batch, frames, s * s, 0, bias=bias,
                                    rot_cos=cos, rot_sin=sin, out=att)
        else:
            qkv, att = self._attn_common(pk, prefix, x, batch * frames, s, c)
            ops.attention_cl(qkv, batch, frames, s * s, 0, bias=bias, rot_cos=cos, rot_sin=sin, out=att)
        out = self._buf(outname, x.shape[0], c)
        return self._conv(att, pk[prefix + "out.w"], c, 1, batch * frames, s, residual=x, out=out)

    def _spatial_attn(self, pk, prefix, x, batch, frames, s, c, outname):
        if _LOWRES_ATTN and c % 64 == 0 and s * s <= 64:
            att = self._buf("at.o", x.shape[0], 256)
            ops.attention_lowres_cl(x, pk[prefix + "qkv.wf"], pk[prefix + "qkv.wsum"], batch, frames, s * s, 1, out=att)
        else:
            qkv, att = self._attn_common(pk, prefix, x, batch * frames, s, c)
            ops.attention_cl(qkv, batch, frames, s * s, 1, out=att)
        out = self._buf(outname, x.shape[0], c)
        return self._conv(att, pk[prefix + "out.w"], c, 1, batch * frames, s, residual=x, out=out)

    def _linear_attn(self, pk, prefix, x, batch, frames, s, c, outname):
        n_img = batch * frames
        if c == 64 and _LINATTN_OUT:      # the whole block in three launches: context partials, merge, output + to_out + bias + residual
            ws = self._buf("la.wsf", 1, ops.linear_attention_fused_ws_floats(n_img, s * s))
            return ops.linear_attention_fused_out_cl(x, pk[prefix + "qkv.wp"], pk[prefix + "out.wp"], pk[prefix + "out.b"], n_img, s * s,
                                                     out=self._buf(outname, x.shape[0], c), ws=ws)
        if c == 64:
            att = self._buf("at.o", x.shape[0], 256)
            ws = self._buf("la.wsf", 1, ops.linear_attention_fused_ws_floats(n_img, s * s))
            ops.linear_attention_fused_cl(x, pk[prefix + "qkv.wp"], n_img, s * s, out=att, ws=ws)
        elif _LOWRES_ATTN and ops.linear_attention_lowres_ok(s * s, c) and s * s <= _LOWRES_MAX_HW:      # one launch: workgroup = (frame, head)
            att = self._buf("at.o", x.shape[0], 256)
            ops.linear_attention_lowres_cl(x, pk[prefix + "qkv.wf"], pk[prefix + "qkv.wsum"], n_img, s * s, out=att)
        else:
            qkv, att = self._attn_common(pk, prefix, x, n_img, s, c)
            ws = self._buf("la.ws", n_img, 8 * 32 * 32)
            ops.linear_attention_cl(qkv, n_img, s * s, out=att, ws=ws)
        out = self._buf(outname, x.shape[0], c)
        return self._conv(att, pk[prefix + "out.w"], c, 1, n_img, s, bias=pk[prefix + "out.b"], residual=x, out=out)

    # ------------------------------------------------------------------ the network
    def fea_term(self, pk, fea_cl, batch, s):
        """Step-invariant part of init_conv: conv7x7 over the 256 `fea` channels + bias,
        (B*S*S, dim) CL; exact split of the 259-channel convolution by linearity (:410, :713, :789)."""
        return self._conv(fea_cl, pk["init.fea_w"], self.dim, 7, batch, s, bias=pk["init.b"],
                          out=self._buf("fea_term", batch * s * s, self.dim))

    def stem(self, pk, x_dyn, add_term, batch, frames, s):
        """Step-dependent part of init_conv: x_dyn planar (B, >=n_dyn, T, S, S) (first n_dyn channels
        are read) + add_term ((B*S*S, dim) CL, broadcast over T, already holding the bias) -> r."""
        r = self._buf("r", batch * frames * s * s, self.dim)
        return ops.conv_planar_in_cl(x_dyn, batch, pk["n_dyn"], x_dyn.shape[1], frames, s, s, pk["init.dyn_w"],
                                     7, 7, self.dim, bias=None if add_term is not None else pk["init.b"],
                                     add_term=add_term, out=r)

    def run_trunk(self, pk, r, ss, batch, frames, s, out, focus=None):
        """r: init_conv output (B*T*S*S, dim) CL; ss: (B, sum 2C) scale/shift rows;
        out: planar (B, 3, T, S, S); focus: None or one bool per sample (focus_present_mask: every temporal attention of the
        down / mid / up path, not init_temporal_attn - :547 vs :570, :575, :584)."""
        n_img = batch * frames
        tables = self._tables(pk, frames)
        dim = self.dim
        x = self._temporal_attn(pk, "init_temporal_attn.", r, batch, frames, s, dim, "x.init", tables)
        skips = []
        res = s
        nl = len(self.levels)
        for lvl, (ci, co) in enumerate(self.levels):
            p = "downs.%d." % lvl
            x = self._resblock(pk, p + "0.", x, None, batch, frames, res, ss, co, "d%d.a" % lvl)
            x = self._resblock(pk, p + "1.", x, None, batch, frames, res, ss, co, "d%d.b" % lvl)
            x = self._linear_attn(pk, p + "2.", x, batch, frames, res, co, "d%d.c" % lvl)
            x = self._temporal_attn(pk, p + "3.", x, batch, frames, res, co, "d%d.skip" % lvl, tables, focus)
            skips.append(x)
            if lvl < nl - 1:
                out_d = self._buf("d%d.down" % lvl, n_img * (res // 2) ** 2, co)
                x = self._conv(x, pk[p + "4.w"], co, 4, n_img, res, bias=pk[p + "4.b"], pad=(1, 1), stride=2, out=out_d)
                res //= 2
        mid = self.levels[-1][1]
        x = self._resblock(pk, "mid_block1.", x, None, batch, frames, res, ss, mid, "m.a")
        x = self._spatial_attn(pk, "mid_spatial_attn.", x, batch, frames, res, mid, "m.b")
        x = self._temporal_attn(pk, "mid_temporal_attn.", x, batch, frames, res, mid, "m.c", tables, focus)
        x = self._resblock(pk, "mid_block2.", x, None, batch, frames, res, ss, mid, "m.d")
        for lvl, (ci, co) in enumerate(reversed(self.levels)):
            p = "ups.%d." % lvl
            x = self._resblock(pk, p + "0.", x, skips.pop(), batch, frames, res, ss, ci, "u%d.a" % lvl)
            x = self._resblock(pk, p + "1.", x, None, batch, frames, res, ss, ci, "u%d.b" % lvl)
            x = self._linear_attn(pk, p + "2.", x, batch, frames, res, ci, "u%d.c" % lvl)
            x = self._temporal_attn(pk, p + "3.", x, batch, frames, res, ci, "u%d.d" % lvl, tables, focus)
            if lvl < nl - 1:
                out_u = self._buf("u%d.up" % lvl, n_img * (res * 2) ** 2, ci)
                if self.use_deconv:      # ConvTranspose (1,4,4) s2 p1 = four 2x2 parity convolutions, ONE launch
                    w4 = pk[p + "4.packs"]
                    self._conv(x, w4[0], ci, 2, n_img, res, bias=pk[p + "4.b"], pad=(1, 1), out=out_u,
                               hq=res, wq=res, ho=2 * res, wo=2 * res, out_scale=2, deconv4=w4)
                    x = out_u
                else:
                    x = self._conv(x, pk[p + "4.w"], ci, 3, n_img, res, bias=pk[p + "4.b"], upsample=True,
                                   reflect=(self.padding_mode == "reflect"), out=out_u)
                res *= 2
        if "heads.block1.w" in pk:
            # final_conv.0 and occlusion_map.0 (:493-509) are two ResnetBlocks(2*dim -> dim) on the SAME input cat(x, r):
            # run as ONE block with 2*dim mid channels - block1 / res_conv dense with the filters concatenated along the
            # output channels, block2 a 2-group convolution, GroupNorm 16 groups of dim/8 - 5 launches instead of 10
            rows, c2 = n_img * res * res, 2 * dim
            h1 = self._buf("h.h1", rows, c2)
            _, st = self._conv(x, pk["heads.block1.w"], c2, 3, n_img, res, src1=r, bias=pk["heads.block1.b"], out=h1,
                               gn=(batch, 16), ww=pk["heads.block1.ww"])
            self._gn(h1, batch, pk["heads.norm1.w"], pk["heads.norm1.b"], st, groups=16)
            y = self._buf("h.y", rows, c2)
            _, st = self._conv(h1, pk["heads.block2.ww"], c2, 3, n_img, res, bias=pk["heads.block2.b"], out=y,
                               gn=(batch, 16), ww=pk["heads.block2.ww"], groups=2)
            fold = _HEADS_FOLD and x.shape[1] % 4 == 0 and r.shape[1] % 4 == 0 and x.shape[1] + r.shape[1] == pk["heads.fold.w"].shape[1]
            if fold and _HEADS_GN and st is not None and c2 <= 512:
                # ... and the block's last GroupNorm + SiLU applied by the heads kernel itself on the raw convolution output: one launch and
                # 2 x 21 MB of traffic less at 40 frames of 32x32
                return ops.heads_gn_res_cl_to_planar(y, st[0], st[1], pk["heads.norm2.w"], pk["heads.norm2.b"], pk["final_conv.1.w"],
                                                     pk["heads.fold.bf"], pk["occlusion_map.1.w"], pk["heads.fold.bo"], x, r, pk["heads.fold.w"],
                                                     batch, frames, res * res, groups=16, out=out)
            self._gn(y, batch, pk["heads.norm2.w"], pk["heads.norm2.b"], st, groups=16)
            if fold:
                # the blocks' res_conv(cat(x, r)) folded into the linear heads: one 1x1 convolution launch less
                return ops.heads_res_cl_to_planar(y[:, :dim], y[:, dim:], pk["final_conv.1.w"], pk["heads.fold.bf"], pk["occlusion_map.1.w"],
                                                  pk["heads.fold.bo"], x, r, pk["heads.fold.w"], batch, frames, res * res, out=out)
            self._conv(x, pk["heads.res.w"], c2, 1, n_img, res, src1=r, bias=pk["heads.res.b"], residual=y, out=y)
            yf, yo = y[:, :dim], y[:, dim:]
        else:
            yf = self._resblock(pk, "final_conv.0.", x, r, batch, frames, res, None, dim, "h.flow")
            yo = self._resblock(pk, "occlusion_map.0.", x, r, batch, frames, res, None, dim, "h.occ")
        ops.heads_cl_to_planar(yf, yo, pk["final_conv.1.w"], pk["final_conv.1.b"], pk["occlusion_map.1.w"],
                               pk["occlusion_map.1.b"], batch, frames, res * res, out=out)
        return out

    # ------------------------------------------------------------------ reference-compatible API
    def forward_with_cond_scale(self, *args, cond_scale=2., **kwargs):
        """:511-526."""
        if cond_scale == 0:
            return self.forward(*args, null_cond_prob=1., **kwargs)
        logits = self.forward(*args, null_cond_prob=0., **kwargs)
        if cond_scale == 1 or not self.has_cond:
            return logits
        null_logits = self.forward(*args, null_cond_prob=1., **kwargs)
        return null_logits + (logits - null_logits) * cond_scale

    def forward(self, x, time, cond=None, null_cond_prob=0., none_cond_mask=None, focus_present_mask=None,
                prob_focus_present=0.):
        """Reference signature (:528-538).  x: (B, channels, T, S, S) planar = [noisy 3 | fea 256]
        (fea may differ per frame here; the samplers use the cheaper split path where it is constant)."""
        if self.has_cond and cond is None:
            raise AssertionError("cond must be passed in if cond_dim specified")
        # :542-543: drawn BEFORE the null-condition mask (it consumes the RNG for 0 < prob < 1)
        if focus_present_mask is None and prob_focus_present in (0, 1):      # (no random draw, no device round trip)
            focus = [True] * x.shape[0] if prob_focus_present == 1 else None
        else:
            if focus_present_mask is None:
                focus_present_mask = prob_mask_like((x.shape[0],), prob_focus_present, device=x.device)
            focus = [bool(v) for v in torch.as_tensor(focus_present_mask).reshape(-1).tolist()]
            if len(focus) != x.shape[0]:
                raise ValueError("focus_present_mask: one entry per sample expected")
            focus = focus if any(focus) else None
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and self.training:
            # training mode under autograd: the differentiable executor (native forward AND backward kernels).  It takes
            # the reference image features as ONE (B,256,S,S) map - which is what the LFDM pipeline feeds (:901 repeats
            # one frame's features over T); per-frame features are refused rather than silently averaged
            from .unet_train import unet_train_forward
            n_dyn = self.channels - 256 if self.channels > 256 else self.channels
            if self.channels == n_dyn:
                raise NotImplementedError("Unet3D.forward under autograd needs the [x | fea] input of the LFDM pipeline")
            fea = x[:, n_dyn:]
            if fea.shape[2] > 1 and not bool((fea == fea[:, :, :1]).all()):
                raise NotImplementedError("Unet3D.forward under autograd: `fea` must be constant over the frame axis "
                                          "(video_flow_diffusion.py:901)")
            return unet_train_forward(self, x[:, :n_dyn].float(), fea[:, :, 0].contiguous().float(), time, cond,
                                      null_cond_prob=null_cond_prob, none_cond_mask=none_cond_mask, focus=focus)
        pk = self.packed()
        x = x.contiguous().float()
        batch, _, frames, s, _ = x.shape
        dev = x.device
        self.null_cond_mask = prob_mask_like((batch,), null_cond_prob, device=dev)
        if none_cond_mask is not None:
            self.null_cond_mask = torch.logical_or(self.null_cond_mask,
                                                   torch.as_tensor(none_cond_mask, device=dev))
        temb = self.time_embedding(pk, time.to(torch.int32).contiguous(), batch)
        if self.has_cond:
            ss = self.cond_scale_shift(pk, temb, self.merge_cond(cond.float(), self.null_cond_mask))
        else:
            ss = ops.linear_small(temb, pk["cond.w"], pk["cond.b"], act_in=ops.ACT_SILU)
        n_dyn = pk["n_dyn"]
        out = torch.empty(batch, self.out_grid_dim + self.out_conf_dim, frames, s, s, device=dev)
        if self.channels > n_dyn:
            # general case: the fea channels may differ per frame -> evaluate their 7x7 conv on all
            # B*T frames and run the 3-channel part with (B*T) folded into the batch axis
            nf = self.channels - n_dyn
            fea = x[:, n_dyn:].permute(0, 2, 1, 3, 4).reshape(batch * frames, nf, s * s).contiguous()
            fea_cl = ops.planar_to_cl(fea, batch * frames, nf, s * s)
            term = self._conv(fea_cl, pk["init.fea_w"], self.dim, 7, batch * frames, s, bias=pk["init.b"])
            xd = x[:, :n_dyn].permute(0, 2, 1, 3, 4).reshape(batch * frames, n_dyn, 1, s, s).contiguous()
            r = self.stem(pk, xd, term, batch * frames, 1, s)
        else:
            r = self.stem(pk, x, None, batch, frames, s)
        return self.run_trunk(pk, r, ss, batch, frames, s, out, focus)
