"""Host-side mirror of the reference pipeline wrapper `FlowDiffusion`
(DM/modules/video_flow_diffusion_model.py:17-253): same constructor keywords, attributes,
setters and `sample_one_video`, so demo_*.py / test_*.py style callers can switch over.
The sampling path (a28) is fully native: LFAE encoder once per video -> hipGraph-replayed
DDIM/DDPM loop -> batched LFAE decode of all T frames.
The DM training step (a29) = frozen-LFAE pseudo ground truth batched over all frames (lfae_predictors.py) ->
diffusion loss through the native UNet forward/backward (unet_train.py, autograd.py) -> fused Adam (optim.py), with
an optional one-process-per-GPU gradient all-reduce (`enable_data_parallel`).
"""
import os

import torch
import yaml
from torch import nn

from .diffusion import GaussianDiffusion
from .generator import Generator
from .params import ParamTree, bg_predictor_spec, build_tree, region_predictor_spec
from .optim import FlatAdam, GradAllReduce
from .unet import Unet3D


class RegionPredictor(ParamTree):
    """LFAE/modules/region_predictor.py: same state-dict keys; forward = lfae_predictors.RegionPredictorExec
    (any number of frames per call)."""

    def __init__(self, num_regions, num_channels, estimate_affine=False, **params):      # (region_predictor.py:34 default)
        super().__init__()
        pca_based = params.get("pca_based", False)       # (region_predictor.py:36 default; every LFDM yaml sets pca_based: true)
        build_tree(self, region_predictor_spec(num_regions=num_regions, num_channels=num_channels, estimate_affine=estimate_affine,
                                               **dict(params, pca_based=pca_based)))
        if self.has("jacobian.weight"):                  # region_predictor.py:46-47: the regression head starts at the identity
            with torch.no_grad():
                self.get("jacobian.weight").zero_()
                self.get("jacobian.bias").copy_(torch.tensor([1, 0, 0, 1], dtype=torch.float))
        from .lfae_predictors import RegionPredictorExec
        self._exec = RegionPredictorExec(self, num_blocks=params.get("num_blocks", 5),
                                         temperature=params.get("temperature", 0.1),
                                         scale_factor=params.get("scale_factor", 0.25),
                                         pca_based=pca_based, pad=params.get("pad", 3), estimate_affine=estimate_affine)

    def forward(self, x):
        return self._exec(x)


class BGMotionPredictor(ParamTree):
    """Parameter holder for LFAE/modules/bg_motion_predictor.py."""

    def __init__(self, num_channels, **params):
        super().__init__()
        build_tree(self, bg_predictor_spec(num_channels=num_channels, **params))
        self.bg_type = params.get("bg_type", "zero")          # (bg_motion_predictor.py:20 default)
        if self.bg_type != "zero":
            from .params import BG_FC_BIAS
            with torch.no_grad():   # reference initialises fc to the identity transform (bg_motion_predictor.py:27-40)
                self.get("fc.bias").copy_(torch.tensor(BG_FC_BIAS[self.bg_type], dtype=torch.float))

        from .lfae_predictors import BGMotionPredictorExec
        self._exec = BGMotionPredictorExec(self, num_blocks=params.get("num_blocks", 5),
                                           bg_type=self.bg_type)

    def forward(self, source_image, driving_image):
        return self._exec(source_image, driving_image)


class FlowDiffusion(nn.Module):
    def __init__(self, img_size=32, num_frames=40, sampling_timesteps=250, null_cond_prob=0.1,
                 ddim_sampling_eta=1., timesteps=1000, dim_mults=(1, 2, 4, 8), lr=1e-4,
                 adam_betas=(0.9, 0.99), is_train=True, only_use_flow=True, use_residual_flow=False,
                 learn_null_cond=False, use_deconv=True, padding_mode="zeros", pretrained_pth="",
                 config_pth="", bert_path=None):
        """Reference signature (video_flow_diffusion_model.py:19-37) + `bert_path`: a local Hugging Face directory of
        bert-base-cased for `cond=list[str]` (the reference downloads it with torch.hub; see text.py).  LFDM_BERT_PATH in
        the environment is the default, so unchanged caller scripts pick it up."""
        super().__init__()
        self.use_residual_flow = use_residual_flow
        self.only_use_flow = only_use_flow
        checkpoint = torch.load(pretrained_pth, map_location="cpu") if pretrained_pth != "" else None
        with open(config_pth) as f:
            mp = yaml.safe_load(f)['model_params']
        self.generator = Generator(num_regions=mp['num_regions'], num_channels=mp['num_channels'],
                                   revert_axis_swap=mp['revert_axis_swap'], **mp['generator_params'])
        self.region_predictor = RegionPredictor(num_regions=mp['num_regions'], num_channels=mp['num_channels'],
                                                estimate_affine=mp['estimate_affine'],
                                                **mp['region_predictor_params'])
        self.bg_predictor = BGMotionPredictor(num_channels=mp['num_channels'], **mp['bg_predictor_params'])
        for name in ('generator', 'region_predictor', 'bg_predictor'):
            net = getattr(self, name)
            if checkpoint is not None:
                net.load_state_dict(checkpoint[name])
                net.eval()
                self.set_requires_grad(net, False)
        self.unet = Unet3D(dim=64, channels=3 + 256, out_grid_dim=2, out_conf_dim=1, dim_mults=dim_mults,
                           use_bert_text_cond=True, learn_null_cond=learn_null_cond,
                           use_final_activation=False, use_deconv=use_deconv, padding_mode=padding_mode)
        self.diffusion = GaussianDiffusion(self.unet, image_size=img_size, num_frames=num_frames,
                                           sampling_timesteps=sampling_timesteps, timesteps=timesteps,
                                           loss_type='l2', use_dynamic_thres=True,
                                           null_cond_prob=null_cond_prob, ddim_sampling_eta=ddim_sampling_eta)
        bert_path = bert_path or os.environ.get("LFDM_BERT_PATH")
        if bert_path:
            from .text import BertTextEncoder
            self.diffusion.text_encoder = BertTextEncoder(bert_path, use_cls=self.diffusion.text_use_bert_cls)
        for attr in ('ref_img', 'ref_img_fea', 'real_vid', 'real_out_vid', 'real_warped_vid', 'real_vid_grid',
                     'real_vid_conf', 'fake_out_vid', 'fake_warped_vid', 'fake_vid_grid', 'fake_vid_conf',
                     'sample_out_vid', 'sample_warped_vid', 'sample_vid_grid', 'sample_vid_conf'):
            setattr(self, attr, None)
        self.is_train = is_train
        if self.is_train:
            self.unet.train()
            self.diffusion.train()
            self.lr = lr
            self.loss = torch.tensor(0.0)
            self.rec_loss = torch.tensor(0.0)
            self.rec_warp_loss = torch.tensor(0.0)
            # a torch.optim.Optimizer (state_dict / param_groups / lr schedulers work) whose step is one fused HIP launch
            self.optimizer_diff = FlatAdam(self.diffusion.parameters(), lr=lr, betas=adam_betas)
        self._dp = None
        self.lazy_real_decode = os.environ.get("LFDM_LAZY_REAL_DECODE", "0") == "1"     # see the real_out_vid property
        self._real_decode, self._real_out_vid, self._real_warped_vid = None, None, None
        self._shard = None            # (rank, world) once data parallelism is on: set_train_input keeps this rank's videos
        self._slice = None            # (lo, hi, global batch) of the current step's shard
        # Launched by torchrun (one process per GPU) from an UNCHANGED training script: there is nobody to call
        # enable_data_parallel(), so the wrapper does it itself at the first optimize_parameters() - the process takes the
        # GPU of its LOCAL_RANK here, before the script's `.cuda()`.  LFDM_AUTO_DP=0 switches this off.
        self._auto_dp = (is_train and int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("LFDM_AUTO_DP", "1") != "0")
        if self._auto_dp and torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())

    # ------------------------------------------------------------------ sampling (a28)
    def sample_one_video(self, cond_scale):
        """Reference :190-216.  Results land in sample_vid_grid (B,2,T,S,S), sample_vid_conf (B,1,T,S,S),
        sample_out_vid / sample_warped_vid (B,3,T,H,W)."""
        gen = self.generator
        with torch.no_grad():
            img = self.sample_img.float().contiguous()
            skips = gen.encode(img)                                   # encoder ONCE per video
            b, _, h, w = img.shape
            d = 2 ** gen.num_down_blocks
            fea_cl = skips[-1]
            fea = gen.compute_fea_from_skips(skips, b, h // d, w // d)
            self.sample_img_fea = fea
            pred = self.diffusion.sample(fea, cond=self.sample_text, batch_size=1, cond_scale=cond_scale)
            nf, s = pred.shape[2], pred.shape[3]
            if self.use_residual_flow:
                grid = pred[:, :2] + self.get_grid(b, nf, s, s, normalize=True).to(pred.device)
                maps = torch.cat((grid, pred[:, 2:3]), dim=1).contiguous()
            else:
                maps = pred
            self.sample_vid_grid = maps[:, :2]
            self.sample_vid_conf = (pred[:, 2, :, :, :].unsqueeze(dim=1) + 1) * 0.5
            out, warped = gen.decode_video(img, skips, maps[:, 0], maps[:, 1], maps[:, 2], nf, s, s,
                                           3 * nf * s * s, s * s, occ_scale=0.5, occ_bias=0.5)
            self.sample_out_vid = out
            self.sample_warped_vid = warped

    def set_sample_input(self, sample_img, sample_text):
        dev = next(self.unet.parameters()).device
        self.sample_img = sample_img.to(dev)
        self.sample_text = sample_text

    # ------------------------------------------------------------------ training (a29) - next row
    def set_train_input(self, ref_img, real_vid, ref_text):
        """Reference :218-221.  Under data parallelism every rank is handed the SAME global batch (the script's DataLoader
        is seeded identically everywhere) and keeps its own contiguous slice - what nn.DataParallel's scatter does in the
        reference (DM/train_video_flow_diffusion_mhad_multiGPU.py:207,249)."""
        dev = next(self.unet.parameters()).device
        self._start_auto_dp()
        self._slice = None
        if self._shard is not None:
            # torch.tensor_split semantics, like nn.DataParallel's scatter: any batch size works (the reference scripts use
            # BATCH_SIZE = 5 and no drop_last); the first b % world ranks get one video more, a rank may get none
            rank, world = self._shard
            b = real_vid.shape[0]
            lo = rank * (b // world) + min(rank, b % world)
            hi = lo + b // world + (1 if rank < b % world else 0)
            self._slice = (lo, hi, b)
            self.diffusion.rank_shard = self._slice
            ref_img, real_vid = ref_img[lo:hi], real_vid[lo:hi]
            ref_text = ref_text[lo:hi] if isinstance(ref_text, torch.Tensor) else list(ref_text)[lo:hi]
        self.ref_img = ref_img.to(dev)
        self.real_vid = real_vid.to(dev)
        self.ref_text = ref_text

    def _train_forward(self, real_vid, ref_img, ref_text, lazy_real=False):
        """Shared body of `forward` (reference :116-179) and of the functional *_multiGPU flavour
        (video_flow_diffusion_model_multiGPU.py:89-157): pseudo ground-truth flow / occlusion of every frame from
        the frozen LFAE (all B*T frames in one batched pass), the diffusion loss on it (native UNet forward under
        autograd), and - for the logged reconstructions - the decode of the denoised prediction.  -> dict."""
        b, _, nf, H, W = real_vid.shape
        gen = self.generator
        out = {}
        with torch.no_grad():
            ref = ref_img.float().contiguous()
            frames = real_vid.float().permute(0, 2, 1, 3, 4).reshape(b * nf, -1, H, W).contiguous()
            source_region_params = self.region_predictor(ref)
            driving_region_params = self.region_predictor(frames)
            ref_rep = ref.unsqueeze(1).expand(b, nf, *ref.shape[1:]).reshape(b * nf, *ref.shape[1:])
            bg_params = self.bg_predictor(ref_rep, frames)
            generated = gen.forward_frames(ref, nf, driving_region_params, source_region_params, bg_params, decode=not lazy_real)
        out["real_vid_grid"] = generated["optical_flow"]
        out["real_vid_conf"] = generated["occlusion_map"]
        if lazy_real:
            out["real_decode"] = generated["decode"]
        else:
            out["real_out_vid"] = generated["prediction"]
            out["real_warped_vid"] = generated["deformed"]
        out["ref_img_fea"] = generated["bottle_neck_feat"].clone().detach()
        if self.is_train:
            h, w = out["real_vid_grid"].shape[-2:]
            identity_grid = self.get_grid(b, nf, h, w, normalize=True).to(ref.device) if self.use_residual_flow else None
            grid = out["real_vid_grid"] - identity_grid if self.use_residual_flow else out["real_vid_grid"]
            res = self.diffusion(torch.cat((grid, out["real_vid_conf"] * 2 - 1), dim=1), out["ref_img_fea"], ref_text)
            if isinstance(res, tuple):
                out["loss"], out["null_cond_mask"] = res
            else:
                out["loss"] = res
            with torch.no_grad():
                pred = self.diffusion.pred_x0
                out["fake_vid_grid"] = pred[:, :2] + identity_grid if self.use_residual_flow else pred[:, :2]
                out["fake_vid_conf"] = (pred[:, 2].unsqueeze(dim=1) + 1) * 0.5
                maps = torch.cat((out["fake_vid_grid"], pred[:, 2:3]), dim=1).contiguous()
                skips = gen.encode(ref)
                fo, fw = gen.decode_video(ref, skips, maps[:, 0], maps[:, 1], maps[:, 2], nf, h, w,
                                          3 * nf * h * w, h * w, occ_scale=0.5, occ_bias=0.5)
                out["fake_out_vid"], out["fake_warped_vid"] = fo, fw
        return out

    def forward(self):
        """Reference :116-179 (inputs from set_train_input, results as attributes)."""
        out = self._train_forward(self.real_vid, self.ref_img, self.ref_text, lazy_real=self.lazy_real_decode)
        for k in ("real_vid_grid", "real_vid_conf", "ref_img_fea"):
            setattr(self, k, out[k])
        if self.lazy_real_decode:
            self._real_decode, self._real_out_vid, self._real_warped_vid = out["real_decode"], None, None
        else:
            self._real_decode, self._real_out_vid, self._real_warped_vid = None, out["real_out_vid"], out["real_warped_vid"]
        if self.is_train:
            for k in ("loss", "fake_vid_grid", "fake_vid_conf", "fake_out_vid", "fake_warped_vid"):
                setattr(self, k, out[k])
            with torch.no_grad():
                self.rec_loss = (self.real_vid - self.fake_out_vid).abs().mean()
                self.rec_warp_loss = (self.real_vid - self.fake_warped_vid).abs().mean()

    # real_out_vid / real_warped_vid (reference :139-140): the LFAE decode of the pseudo ground truth feeds no loss - the
    # training scripts only write it to their sample images every `save_img_freq` steps.  By default it is computed in
    # forward() like the reference does; with `lazy_real_decode = True` (or LFDM_LAZY_REAL_DECODE=1) the 320-frame decode
    # (a sixth of the B = 8 step) runs when one of the two attributes is first read.
    def _materialise_real(self):
        if self._real_out_vid is None and self._real_decode is not None:
            self._real_out_vid, self._real_warped_vid = self._real_decode()
            self._real_decode = None

    @property
    def real_out_vid(self):
        self._materialise_real()
        return self._real_out_vid

    @real_out_vid.setter
    def real_out_vid(self, v):
        self._real_out_vid = v

    @property
    def real_warped_vid(self):
        self._materialise_real()
        return self._real_warped_vid

    @real_warped_vid.setter
    def real_warped_vid(self, v):
        self._real_warped_vid = v

    def enable_data_parallel(self, bucket_bytes=64 << 20, shard_inputs=True):
        """One process per GPU (torch.distributed initialised by the launcher): average the DM gradients over the
        ranks with a bucketed RCCL all-reduce that overlaps backward.  No-op for world size 1.
        shard_inputs: every rank receives the same global batch and keeps its slice (set_train_input); the step's random
        draws (t, noise, null-condition mask) are made for the global batch and sliced, so N ranks reproduce the
        single-process step on the same batch exactly.  shard_inputs=False: the caller already feeds rank-local batches
        (tools/train_dm.py with a DistributedSampler)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            self._dp = GradAllReduce(self.optimizer_diff, bucket_bytes=bucket_bytes)
            self._dp.sync_replicas()          # rank 0's parameters + Adam moments -> all ranks, once
            if shard_inputs:
                self._shard = (dist.get_rank(), dist.get_world_size())
                self.diffusion.rank_shard = self._shard
        return self

    def _start_auto_dp(self):
        if not self._auto_dp or self._dp is not None:
            return
        import torch.distributed as dist
        if not dist.is_initialized():
            backend = os.environ.get("LFDM_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")     # "nccl" = RCCL
            dist.init_process_group(backend)
        self._auto_dp = False
        self.enable_data_parallel()

    def optimize_parameters(self):
        """:181-188.  Under sharded data parallelism the rank's loss is the mean over ITS videos: it is weighted by
        shard_size * world / global_batch before backward, so that the all-reduced sum times 1/world (folded into the Adam
        kernel) is the gradient of the mean over the global batch for any split; a rank without videos this step skips the
        model, advances the random generator like everybody else and contributes zero gradients."""
        weight, empty = 1.0, False
        if self._dp is not None and getattr(self, "_slice", None) is not None:
            lo, hi, total = self._slice
            weight, empty = (hi - lo) * self._shard[1] / float(total), hi == lo
        if empty:
            dev = next(self.unet.parameters()).device
            s = self.diffusion.image_size
            # (forward() calls self.diffusion(x, fea, text) without focus arguments: prob_focus_present = 0, no focus-mask draw to replay)
            self.diffusion.skip_step_draws(self._slice[2], (3, self.diffusion.num_frames, s, s), dev, prob_focus_present=0.)
            self.unet.null_cond_mask = torch.zeros(0, dtype=torch.bool, device=dev)
            self.loss = torch.zeros((), device=dev)
            self.rec_loss, self.rec_warp_loss = torch.zeros((), device=dev), torch.zeros((), device=dev)
            # what the training scripts read after a step (sample images, logs): zero-video tensors of the right rank, so an empty
            # rank's first step does not meet attributes that only forward() would have created
            nf, hw = self.diffusion.num_frames, self.real_vid.shape[-1] if getattr(self, "real_vid", None) is not None else 4 * s
            vid = torch.zeros(0, 3, nf, hw, hw, device=dev)
            self.real_vid_grid = self.fake_vid_grid = torch.zeros(0, 2, nf, s, s, device=dev)
            self.real_vid_conf = self.fake_vid_conf = torch.zeros(0, 1, nf, s, s, device=dev)
            self.ref_img_fea = torch.zeros(0, 256, s, s, device=dev)
            self.fake_out_vid = self.fake_warped_vid = vid
            self._real_decode, self._real_out_vid, self._real_warped_vid = None, vid, vid
            self.optimizer_diff.zero_grad()
            self._dp.prepare()
            self._dp.finish()
            self.optimizer_diff.step()
            return
        self.forward()
        self.optimizer_diff.zero_grad()
        if self._dp is not None:
            self._dp.prepare()
        total_loss = self.loss if self.only_use_flow else self.loss + self.rec_loss + self.rec_warp_loss
        (total_loss * weight if weight != 1.0 else total_loss).backward()
        if self._dp is not None:
            self._dp.finish()
        self.optimizer_diff.step()

    # ------------------------------------------------------------------ misc (reference :227-253)
    def print_learning_rate(self):
        lr = self.optimizer_diff.param_groups[0]['lr']
        assert lr > 0
        print('lr= %.7f' % lr)

    def get_grid(self, b, nf, H, W, normalize=True):
        """linspace(-1,1) identity grid in (x, y) order, (B, 2, nf, H, W) (:232-240)."""
        ys = torch.linspace(-1, 1, H) if normalize else torch.arange(0, H).float()
        xs = torch.linspace(-1, 1, W) if normalize else torch.arange(0, W).float()
        gy, gx = ys.view(H, 1).expand(H, W), xs.view(1, W).expand(H, W)
        grid = torch.stack((gx, gy), dim=0).float()
        return grid.view(1, 2, 1, H, W).repeat(b, 1, nf, 1, 1)

    def set_requires_grad(self, nets, requires_grad=False):
        if not isinstance(nets, list):
            nets = [nets]
        for net in nets:
            if net is not None:
                for p in net.parameters():
                    p.requires_grad = requires_grad


class FlowDiffusionFunctional(FlowDiffusion):
    """The *_multiGPU.py flavour of the wrapper (DM/modules/video_flow_diffusion_model_multiGPU.py): functional
    `forward(real_vid, ref_img, ref_text) -> dict` with an un-reduced `loss` and `null_cond_mask`, functional
    `sample_one_video(sample_img, sample_text, cond_scale) -> dict`, optimizer owned by the training script
    (DM/train_video_flow_diffusion_mhad_multiGPU.py:182,249-299,357).  Data parallelism is one process per GPU here
    (wrap the script's optimizer step with `GradAllReduce`, or use `FlowDiffusion.enable_data_parallel`)."""

    def __init__(self, *args, **kwargs):
        kwargs.pop("lr", None)
        super().__init__(*args, **kwargs)
        self.diffusion.per_element_loss = True

    def forward(self, real_vid, ref_img, ref_text):
        out = self._train_forward(real_vid, ref_img, ref_text)
        if self.is_train:
            with torch.no_grad():
                out["rec_loss"] = (real_vid - out["fake_out_vid"]).abs()
                out["rec_warp_loss"] = (real_vid - out["fake_warped_vid"]).abs()
        out.pop("ref_img_fea", None)
        return out

    def sample_one_video(self, sample_img, sample_text, cond_scale):
        self.set_sample_input(sample_img=sample_img, sample_text=sample_text)
        FlowDiffusion.sample_one_video(self, cond_scale)
        return {k: getattr(self, k) for k in ("sample_vid_grid", "sample_vid_conf", "sample_out_vid", "sample_warped_vid")}
