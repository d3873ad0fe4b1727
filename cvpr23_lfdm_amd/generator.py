"""Host-side mirror of the reference LFAE `Generator` (LFAE/modules/generator.py:17-166): same
constructor, same state-dict keys, same `compute_fea` / `forward_with_flow` signatures and outputs.

MI355X-first re-plan of the decode step:
  * the source-image encoder is evaluated ONCE per video (the reference re-runs it for each of the
    T frames, generator.py:137-141 - the source image does not depend on the frame);
  * all T frames of a video batch are decoded together (N = B*T images per kernel launch) on
    channels-last rows, with eval-BatchNorm folded into the adjacent convolution;
  * every `deform_input` + `apply_optical` pair is one fused warp kernel that reads the 32x32 flow /
    occlusion prediction of the diffusion model in place (no up-sampled grid is materialised).
"""
import torch

from . import ops
from .params import ParamTree, build_tree, generator_spec

BN_EPS = 1e-5


class Generator(ParamTree):
    def __init__(self, num_channels, num_regions, block_expansion, max_features, num_down_blocks,
                 num_bottleneck_blocks, pixelwise_flow_predictor_params=None, skips=False,
                 revert_axis_swap=True):
        super().__init__()
        fp = pixelwise_flow_predictor_params or {}
        build_tree(self, generator_spec(
            num_channels=num_channels, block_expansion=block_expansion, max_features=max_features,
            num_down_blocks=num_down_blocks, num_bottleneck_blocks=num_bottleneck_blocks,
            num_regions=num_regions, with_flow_predictor=pixelwise_flow_predictor_params is not None,
            fp_block_expansion=fp.get("block_expansion", 64), fp_max_features=fp.get("max_features", 1024),
            fp_num_blocks=fp.get("num_blocks", 5), use_deformed_source=fp.get("use_deformed_source", True)))
        self.num_channels = num_channels
        self.num_down_blocks = num_down_blocks
        self.num_bottleneck_blocks = num_bottleneck_blocks
        self.block_expansion = block_expansion
        self.max_features = max_features
        self.skips = skips
        self.num_regions = num_regions
        self.flow_predictor_cfg = dict(
            num_blocks=fp.get("num_blocks", 5), scale_factor=fp.get("scale_factor", 1),
            use_covar_heatmap=fp.get("use_covar_heatmap", False), use_deformed_source=fp.get("use_deformed_source", True),
            revert_axis_swap=revert_axis_swap)
        self._fp = None
        self.frames_per_chunk = 160      # decode working set bound (images per launch)
        self._pk = None
        self._pk_sig = None
        self._bufs = {}

    # ------------------------------------------------------------------ plumbing
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._pk = None
        self._bufs = {}
        return out

    def _buf(self, name, rows, ch):
        need = rows * ch
        cur = self._bufs.get(name)
        dev = next(self.parameters()).device
        if cur is None or cur.numel() < need or cur.device != dev:
            cur = torch.empty(need, dtype=torch.float32, device=dev)
            self._bufs[name] = cur
        return cur[:need].view(rows, ch)

    def packed(self):
        sig = (sum(p._version for p in self.parameters()) + sum(b._version for b in self.buffers()),
               next(self.parameters()).device)
        if self._pk is None or self._pk_sig != sig:
            with torch.no_grad():
                self._pk = self._pack()
            self._pk_sig = sig
        return self._pk

    def _pack(self):
        g = lambda k: self.get(k).detach().float()
        pk = {}

        def bn_affine(prefix):
            a = g(prefix + "weight") / torch.sqrt(g(prefix + "running_var") + BN_EPS)
            return a, g(prefix + "bias") - g(prefix + "running_mean") * a

        def conv_bn(cprefix, nprefix):
            """conv followed by eval BatchNorm -> one conv (util.py:107-150)."""
            a, b = bn_affine(nprefix)
            w = g(cprefix + "weight") * a.view(-1, 1, 1, 1)
            return w.contiguous(), (g(cprefix + "bias") * a + b).contiguous()

        w, b = conv_bn("first.conv.", "first.norm.")
        pk["first.w"], pk["first.b"] = ops.pack_planar_in_weight(w), b
        for i in range(self.num_down_blocks):
            w, b = conv_bn("down_blocks.%d.conv." % i, "down_blocks.%d.norm." % i)
            pk["down%d.w" % i], pk["down%d.b" % i] = ops.pack_conv_weight(w), b
            pk["down%d.ww" % i] = ops.pack_wino_weight(w) if w.shape[1] % 16 == 0 else None
        for i in range(self.num_down_blocks):
            w, b = conv_bn("up_blocks.%d.conv." % i, "up_blocks.%d.norm." % i)
            pk["up%d.w" % i], pk["up%d.b" % i] = ops.pack_conv_weight(w), b
            pk["up%d.ww" % i] = ops.pack_wino_weight(w) if w.shape[1] % 16 == 0 else None
            pk["up%d.w4" % i] = ops.pack_wino4_weight(w) if w.shape[1] % 16 == 0 else None      # F(4x4,3x3): batched launches only
        for i in range(self.num_bottleneck_blocks):
            p = "bottleneck.r%d." % i
            a1, b1 = bn_affine(p + "norm1.")                 # pre-activation BN + ReLU (util.py:85-86)
            pk["r%d.a1" % i], pk["r%d.b1" % i] = a1.contiguous(), b1.contiguous()
            w, b = conv_bn(p + "conv1.", p + "norm2.")       # conv1 -> norm2 folded, ReLU in the epilogue
            pk["r%d.w1" % i], pk["r%d.bb1" % i] = ops.pack_conv_weight(w), b
            pk["r%d.w2" % i] = ops.pack_conv_weight(g(p + "conv2.weight").contiguous())
            # Winograd F(2x2,3x3) forms of the same filters (the library chooses the schedule)
            pk["r%d.ww1" % i] = ops.pack_wino_weight(w) if w.shape[1] % 16 == 0 else None
            pk["r%d.ww2" % i] = ops.pack_wino_weight(g(p + "conv2.weight").contiguous()) if w.shape[1] % 16 == 0 else None
            # ... and the F(4x4,3x3) forms: the library takes them only for launches of >= 2048 workgroups (the 320-frame decode of a
            # training step / throughput mode), never for the 40 frames of a B = 1 sample (lfdm_conv_params.weight_wino4)
            pk["r%d.w41" % i] = ops.pack_wino4_weight(w) if w.shape[1] % 16 == 0 else None
            pk["r%d.w42" % i] = ops.pack_wino4_weight(g(p + "conv2.weight").contiguous()) if w.shape[1] % 16 == 0 else None
            pk["r%d.b2" % i] = g(p + "conv2.bias").contiguous()
        # output channels padded to a multiple of 4 (zero filters): float4 epilogue -> the 32-column KSW tile instead of
        # a 64-column tile for 3 real channels
        wf, bf = g("final.weight"), g("final.bias")
        cpad = (4 - wf.shape[0] % 4) % 4
        pk["final.w"] = ops.pack_conv_weight(torch.cat((wf, wf.new_zeros(cpad, *wf.shape[1:])), dim=0).contiguous())
        pk["final.b"] = torch.cat((bf, bf.new_zeros(cpad))).contiguous()
        pk["final.cout"] = wf.shape[0] + cpad
        # <= 4 output channels: sixteen 4x4 MFMA blocks per instruction (lane = pixel) instead of a 32-column tile
        pk["final.small"] = None
        if wf.shape[0] <= 4 and wf.shape[1] % 16 == 0 and wf.shape[2] == wf.shape[3] and wf.shape[2] % 2 == 1 and wf.shape[2] <= 7:
            pk["final.small"] = ops.pack_smalln_weight(wf, bf)
        return pk

    def _feat(self, i):
        return min(self.max_features, self.block_expansion * (2 ** i))

    # ------------------------------------------------------------------ encoder (once per video)
    def encode(self, source_image):
        """first + down blocks (generator.py:137-141): returns CL skips
        [(B*H*W, 64), (B*H/2*W/2, 128), (B*H/4*W/4, 256)] for a 128^2 MUG config."""
        pk = self.packed()
        img = source_image.float().contiguous()
        b, c, h, w = img.shape
        out = ops.conv_planar_in_cl(img, b, c, c, 1, h, w, pk["first.w"], 7, 7, self._feat(0), bias=pk["first.b"],
                                    act=ops.ACT_RELU, out=self._buf("enc0", b * h * w, self._feat(0)))
        skips = [out]
        res_h, res_w = h, w
        for i in range(self.num_down_blocks):
            co = self._feat(i + 1)
            pooled = self._buf("enc%d" % (i + 1), b * (res_h // 2) * (res_w // 2), co)
            try:          # DownBlock2d's 2x2 average pool in the convolution's epilogue (Winograd schedule)
                ops.conv2d_cl(out, pk["down%d.w" % i], co, 3, 3, b, res_h, res_w, bias=pk["down%d.b" % i], act=ops.ACT_RELU,
                              out=pooled, weight_wino=pk["down%d.ww" % i], pool2=True)
            except ops.WinogradUnavailable:
                y = ops.conv2d_cl(out, pk["down%d.w" % i], co, 3, 3, b, res_h, res_w, bias=pk["down%d.b" % i],
                                  act=ops.ACT_RELU, out=self._buf("enc.t", b * res_h * res_w, co), weight_wino=pk["down%d.ww" % i])
                ops.avgpool2_cl(y, b, res_h, res_w, out=pooled)
            res_h, res_w = res_h // 2, res_w // 2
            out = pooled
            skips.append(out)
        return skips

    def compute_fea_from_skips(self, skips, b, lh, lw):
        co = self._feat(self.num_down_blocks)
        return ops.cl_to_planar(skips[-1], b, co, lh * lw).view(b, co, lh, lw)

    def compute_fea(self, source_image):
        """Reference :130-134 -> planar (B, 256, H/4, W/4)."""
        with torch.no_grad():
            skips = self.encode(source_image)
            b, _, h, w = source_image.shape
            d = 2 ** self.num_down_blocks
            return self.compute_fea_from_skips(skips, b, h // d, w // d)

    # ------------------------------------------------------------------ decode
    def decode_video(self, source_image, skips, flow_x, flow_y, occ, frames, fh, fw, fsb, fst,
                     occ_scale=1.0, occ_bias=0.0):
        """All frames of a video batch: returns planar (prediction, deformed) of shape (B, C, T, H, W).
        flow_x/flow_y/occ are base tensors of low-res maps addressed b*fsb + t*fst + y*fw + x."""
        pk = self.packed()
        img = source_image.float().contiguous()
        b, c, h, w = img.shape
        n = b * frames
        wk = dict(fh=fh, fw=fw, fsb=fsb, fst=fst, occ_scale=occ_scale, occ_bias=occ_bias)
        deformed = ops.warp_planar(img, frames, flow_x, flow_y, None, fh, fw, fsb, fst)
        d = 2 ** self.num_down_blocks
        lh, lw = h // d, w // d
        cb = self._feat(self.num_down_blocks)
        # bottleneck input: warped + masked latent (generator.py:149)
        out = ops.warp_cl(skips[-1], b, frames, lh, lw, flow_x, flow_y, occ, out=self._buf("dec.x", n * lh * lw, cb), **wk)
        for i in range(self.num_bottleneck_blocks):          # ResBlock2d (util.py:84-92)
            t0 = ops.affine_act_cl(out, pk["r%d.a1" % i], pk["r%d.b1" % i], ops.ACT_RELU,
                                   out=self._buf("dec.t0", n * lh * lw, cb))
            t1 = ops.conv2d_cl(t0, pk["r%d.w1" % i], cb, 3, 3, n, lh, lw, bias=pk["r%d.bb1" % i], act=ops.ACT_RELU,
                               out=self._buf("dec.t1", n * lh * lw, cb), weight_wino=pk["r%d.ww1" % i], weight_wino4=pk["r%d.w41" % i])
            out = ops.conv2d_cl(t1, pk["r%d.w2" % i], cb, 3, 3, n, lh, lw, bias=pk["r%d.b2" % i], residual=out,
                                out=out, weight_wino=pk["r%d.ww2" % i], weight_wino4=pk["r%d.w42" % i])
        res_h, res_w = lh, lw
        for i in range(self.num_down_blocks):                # apply_optical(skip, prev) + UpBlock2d (:152-155)
            skip = skips[-(i + 1)]
            ci = skip.shape[1]
            blended = out                                    # Generator(skips=False): no skip blending (:152-153)
            if self.skips:
                blended = ops.warp_cl(skip, b, frames, res_h, res_w, flow_x, flow_y, occ, prev=out,
                                      out=self._buf("dec.w%d" % i, n * res_h * res_w, ci), **wk)
            co = self._feat(self.num_down_blocks - i - 1)
            out = ops.conv2d_cl(blended, pk["up%d.w" % i], co, 3, 3, n, res_h, res_w, bias=pk["up%d.b" % i],
                                upsample=True, act=ops.ACT_RELU, weight_wino=pk["up%d.ww" % i], weight_wino4=pk["up%d.w4" % i],
                                out=self._buf("dec.u%d" % i, n * 4 * res_h * res_w, co))
            res_h, res_w = res_h * 2, res_w * 2
        blended = out
        if self.skips:
            blended = ops.warp_cl(skips[0], b, frames, res_h, res_w, flow_x, flow_y, occ, prev=out,
                                  out=self._buf("dec.wf", n * res_h * res_w, skips[0].shape[1]), **wk)
        rgb_buf = self._buf("dec.rgb", n * res_h * res_w, pk["final.cout"])
        if pk["final.small"] is not None:
            wsm, bsm = pk["final.small"]
            rgb = ops.conv2d_smalln_cl(blended, wsm, bsm, c, int(round(wsm.shape[0] ** 0.5)), n, res_h, res_w,
                                       act=ops.ACT_SIGMOID, out=rgb_buf)[:, :c]
        else:
            rgb = ops.conv2d_cl(blended, pk["final.w"], pk["final.cout"], 7, 7, n, res_h, res_w, bias=pk["final.b"],
                                act=ops.ACT_SIGMOID, out=rgb_buf)[:, :c]
        if not self.skips:          # (:160-161: no final blend with the warped source - only the layout change remains; no LFDM config)
            return rgb.reshape(b, frames, res_h, res_w, c).permute(0, 4, 1, 2, 3).contiguous(), deformed
        prediction = ops.warp_planar(img, frames, flow_x, flow_y, occ, fh, fw, fsb, fst, prev=rgb, prev_is_cl=True,
                                     occ_scale=occ_scale, occ_bias=occ_bias)
        return prediction, deformed

    def forward_with_flow(self, source_image, optical_flow, occlusion_map):
        """Reference :136-166.  optical_flow (B, h, w, 2) absolute sampling grid, occlusion_map
        (B, 1, h, w) in [0,1]  ->  {'prediction', 'deformed'} (B, C, H, W)."""
        with torch.no_grad():
            b = source_image.shape[0]
            fh, fw = optical_flow.shape[1], optical_flow.shape[2]
            maps = torch.empty(b, 3, fh, fw, dtype=torch.float32, device=source_image.device)
            maps[:, 0] = optical_flow[..., 0]
            maps[:, 1] = optical_flow[..., 1]
            maps[:, 2] = occlusion_map[:, 0]
            skips = self.encode(source_image)
            pred, deformed = self.decode_video(source_image, skips, maps[:, 0], maps[:, 1], maps[:, 2], 1, fh, fw,
                                               3 * fh * fw, 0)
            return {"prediction": pred[:, :, 0], "deformed": deformed[:, :, 0]}

    # ------------------------------------------------------------------ training pseudo ground truth (a35)
    def flow_predictor(self):
        if getattr(self, "_fp", None) is None:
            from .lfae_predictors import PixelwiseFlowPredictorExec
            if not self.has("pixelwise_flow_predictor.mask.weight"):
                raise RuntimeError("Generator was built without pixelwise_flow_predictor_params")
            self._fp = PixelwiseFlowPredictorExec(self, self.num_regions, **self.flow_predictor_cfg)
        return self._fp

    def forward_frames(self, source_image, frames, driving_region_params, source_region_params, bg_params=None, decode=True):
        """Batched Generator.forward for `frames` driving frames per source image (the training loop of
        video_flow_diffusion_model.py:124-137 in one pass).  source_image (B,C,H,W); driving params / bg_params have
        leading dimension B*frames (n = b*frames + t); source params leading dimension B.
        -> dict: optical_flow (B,2,T,h,w), occlusion_map (B,1,T,h,w), prediction / deformed (B,C,T,H,W),
                 bottle_neck_feat (B,256,h,w).  decode=False: instead of prediction / deformed the dict carries `decode`, a
                 callable producing that pair on demand (FlowDiffusion.lazy_real_decode)."""
        with torch.no_grad():
            img = source_image.float().contiguous()
            b, c, h, w = img.shape
            n = b * frames
            rep = lambda v: v.unsqueeze(1).expand(b, frames, *v.shape[1:]).reshape(n, *v.shape[1:])
            src = {k: v for k, v in source_region_params.items() if k in ("shift", "covar", "affine")}
            motion = self.flow_predictor()(img, driving_region_params, src, bg_params=bg_params, frames=frames)
            flow, occ = motion["optical_flow"], motion["occlusion_map"]          # (N,h,w,2), (N,1,h,w)
            fh, fw = flow.shape[1], flow.shape[2]
            maps = torch.empty(b, 3, frames, fh, fw, dtype=torch.float32, device=img.device)
            maps[:, 0] = flow[..., 0].reshape(b, frames, fh, fw)
            maps[:, 1] = flow[..., 1].reshape(b, frames, fh, fw)
            maps[:, 2] = occ[:, 0].reshape(b, frames, fh, fw)
            skips = self.encode(img)
            d = 2 ** self.num_down_blocks
            fea = self.compute_fea_from_skips(skips, b, h // d, w // d).clone()
            if not decode:
                # the deferred decode must not read the executor's reusable arenas (skips are views of them): any other generator
                # pass before real_out_vid / real_warped_vid is first read would silently change the video
                skips = [t.clone() for t in skips]

            def run_decode():
                with torch.no_grad():
                    return self.decode_video(img, skips, maps[:, 0], maps[:, 1], maps[:, 2], frames, fh, fw,
                                             3 * frames * fh * fw, fh * fw)
            if not decode:
                return {"optical_flow": maps[:, :2], "occlusion_map": maps[:, 2:3], "decode": run_decode, "bottle_neck_feat": fea}
            pred, deformed = run_decode()
            return {"optical_flow": maps[:, :2], "occlusion_map": maps[:, 2:3], "prediction": pred,
                    "deformed": deformed, "bottle_neck_feat": fea}

    def forward(self, source_image, driving_region_params, source_region_params, bg_params=None):
        """Reference :90-128 (one driving frame per source image): dict with `prediction`, `deformed` (B,C,H,W),
        `optical_flow` (B,h,w,2), `occlusion_map` (B,1,h,w), `bottle_neck_feat` (B,256,h,w)."""
        out = self.forward_frames(source_image, 1, driving_region_params, source_region_params, bg_params)
        return {"prediction": out["prediction"][:, :, 0], "deformed": out["deformed"][:, :, 0],
                "optical_flow": out["optical_flow"][:, :, 0].permute(0, 2, 3, 1).contiguous(),
                "occlusion_map": out["occlusion_map"][:, :, 0], "bottle_neck_feat": out["bottle_neck_feat"]}
