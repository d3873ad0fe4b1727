"""Optimizer + data-parallel gradient exchange of the DM training step.

`FlatAdam` is a real `torch.optim.Optimizer` (the training scripts call `.state_dict()`, `.load_state_dict()`,
`.param_groups[0]['lr']` and wrap it in `MultiStepLR`: DM/train_video_flow_diffusion_mug.py:181,210-211,367) whose
step is ONE fused HIP launch (lfdm_adam_step_f32) over flat fp32 buffers: parameters, gradients and both moments
live contiguously, each tensor being a view.  The same flat gradient buffer is what `GradAllReduce` hands to RCCL -
bucketed sum all-reduce over xGMI launched from autograd hooks while backward is still running; the 1/world of the
mean is folded into the Adam kernel.  (Reference multi-GPU: nn.DataParallel threads + NCCL reduce to GPU 0,
SURVEY.md 3.4; here: one process per GPU, every rank applies the identical update, no parameter broadcast.)
"""
import torch

from .ops import _chk, _lib, _p, _stream
from .params import bump_weights_epoch


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._flat = None            # per group: dict(params, offs, p, g, m, v, gviews)
        self._staged = False         # gradients already gathered into the flat buffer (GradAllReduce.finish)
        self.grad_scale = 1.0        # GradAllReduce sets 1/world

    # ------------------------------------------------------------------ flat storage
    def _build(self):
        flats = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            dev, n = ps[0].device, 0
            offs = []
            for p in ps:
                offs.append(n)
                n += (p.numel() + 3) // 4 * 4          # keep every view 16-byte aligned
            fp = torch.zeros(n, dtype=torch.float32, device=dev)
            fg = torch.zeros_like(fp)
            fm = torch.zeros_like(fp)
            fv = torch.zeros_like(fp)
            gviews = []
            for pi, (p, o) in enumerate(zip(ps, offs)):
                k = p.numel()
                fp[o:o + k].copy_(p.data.reshape(-1))
                p.data = fp[o:o + k].view(p.shape)
                gviews.append(fg[o:o + k].view(p.shape))
                # autograd.grad_out: the native backward kernels write the step's first gradient of `p` straight into this slot
                p._lfdm_grad_slot, p._lfdm_grad_slot_busy = gviews[-1], False
                st = self.state[p]
                if "exp_avg" in st:                    # state loaded before the first step
                    if st["exp_avg"].numel() != k or st["exp_avg_sq"].numel() != k:
                        raise ValueError("FlatAdam.load_state_dict: the moment of parameter #%d has %d elements, the parameter %s has %d - "
                                         "the state was saved for a different parameter ORDER (torch indexes optimizer state by "
                                         "position: it must come from a model with the same registration order)"
                                         % (pi, st["exp_avg"].numel(), tuple(p.shape), k))
                    fm[o:o + k].copy_(st["exp_avg"].reshape(-1))
                    fv[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
                st.setdefault("step", torch.tensor(0.0))
                st["exp_avg"] = fm[o:o + k].view(p.shape)
                st["exp_avg_sq"] = fv[o:o + k].view(p.shape)
            flats.append(dict(params=ps, offs=offs, p=fp, g=fg, m=fm, v=fv, gviews=gviews))
        self._flat = flats
        self._staged = False

    def _valid(self):
        if self._flat is None:
            return False
        for fl in self._flat:
            base = fl["p"].data_ptr()
            for p, o in zip(fl["params"], fl["offs"]):
                if p.data_ptr() != base + 4 * o:
                    return False
        return True

    def ensure_flat(self):
        """(Re)build the flat buffers if parameters were moved / replaced (e.g. `.cuda()` after construction)."""
        if not self._valid():
            self._build()
        return self._flat

    def flat_grads(self):
        return [fl["g"] for fl in self.ensure_flat()]

    def zero_grad(self, set_to_none=True):
        """torch semantics.  With set_to_none (the default, what the training scripts get) autograd hands every parameter its
        gradient tensor as is - no `grad += new` kernel per parameter (302 of them per step when the gradients were kept as
        views of the flat buffer) - and `stage_grads` gathers them into the flat buffer with a few multi-tensor copies."""
        self._staged = False
        for fl in self.ensure_flat():
            for p in fl["params"]:
                p._lfdm_grad_slot_busy = False
            if set_to_none:
                for p in fl["params"]:
                    p.grad = None
            else:
                gs = [p.grad for p in fl["params"] if p.grad is not None]
                if gs:
                    torch._foreach_zero_(gs)

    @torch.no_grad()
    def stage_grads(self, fl=None, idxs=None):
        """Gather p.grad of the given parameters (default: all) into their slots of the flat gradient buffer and re-point
        p.grad at the slot (so p.grad shows what the update / the all-reduce sees; a backward without zero_grad then
        accumulates in place).  Parameters without a gradient this step get a zero slot."""
        todo = [(f, range(len(f["params"]))) for f in self.ensure_flat()] if fl is None else [(fl, idxs)]
        for f, ids in todo:
            src, dst = [], []
            for i in ids:
                p, slot = f["params"][i], f["gviews"][i]
                g = p.grad
                if g is None:
                    slot.zero_()
                elif g.data_ptr() != slot.data_ptr():
                    src.append(g if g.shape == slot.shape else g.reshape(slot.shape))
                    dst.append(slot)
                p.grad = slot
                p._lfdm_grad_slot_busy = False
            if dst:
                torch._foreach_copy_(dst, src)
        if fl is None:
            self._staged = True

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat = None             # loaded moments are folded into fresh flat buffers at the next step

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib()
        self.ensure_flat()
        if not self._staged:
            self.stage_grads()
        self._staged = False
        for group, fl in zip(self.param_groups, self.ensure_flat()):
            st0 = self.state[fl["params"][0]]
            step = int(st0["step"]) + 1
            for p in fl["params"]:
                self.state[p]["step"] = torch.tensor(float(step))
            _chk(lib, fl["p"], fl["g"], fl["m"], fl["v"])
            b1, b2 = group["betas"]
            lib.check(lib.lfdm_adam_step_f32(_p(fl["p"]), _p(fl["g"]), _p(fl["m"]), _p(fl["v"]), fl["p"].numel(),
                                             float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                             float(group["weight_decay"]), step, float(self.grad_scale), _stream(lib)),
                      "lfdm_adam_step_f32")
        bump_weights_epoch()      # the kernel wrote the parameters behind torch's back: packed weight caches must rebuild
        return loss


class GradAllReduce:
    """Sum all-reduce of FlatAdam's gradient buffer across ranks (torch.distributed; backend "nccl" = RCCL over xGMI
    on the GPU box, "gloo" in CPU tests).  Buckets are contiguous slices of the flat buffer, walked from the END
    (the last layers finish their gradients first); a bucket becomes READY in the autograd hook of the last of its
    parameters, so the exchange overlaps the remaining backward.  Collectives are issued STRICTLY IN BUCKET-LIST ORDER on
    every rank: a ready bucket is launched only once all buckets before it have been launched (RCCL / gloo match
    collectives by issue order and the buckets differ in size - a rank whose hooks fire in another order, or a rank with
    an empty shard that launches everything from `finish()`, must not permute them).  `finish()` launches what is left
    (parameters without gradient this step, an empty shard) in the same order and waits for all buckets."""

    def __init__(self, optimizer, bucket_bytes=64 << 20, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.opt = optimizer
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        optimizer.grad_scale = 1.0 / self.world
        self.bucket_elems = max(bucket_bytes // 4, 1)
        self._hooks = []
        self._buckets = None
        self._flat_id = None
        self._next = 0
        self.launch_order = []
        # profile = True: finish() brackets its waits with events on the compute stream; exposed_ms() then reports how long the
        # compute stream stood still for the exchange per step (the part of the all-reduce NOT hidden under backward)
        self.profile = False
        self._wait_events = []

    def sync_replicas(self, src=0):
        """Broadcast rank `src`'s flat parameter buffer and Adam moments (one collective each per group) so that the
        replicas start from identical state whatever seeds / restores happened before; afterwards identical averaged
        gradients keep them identical (checked by `replica_checksum`).  The reference's nn.DataParallel re-broadcasts
        the weights from GPU 0 on every forward (DM/train_video_flow_diffusion_mhad_multiGPU.py:207) - here once."""
        if self.world <= 1:
            return
        for group, fl in zip(self.opt.param_groups, self.opt.ensure_flat()):
            for key in ("p", "m", "v"):
                self.dist.broadcast(fl[key], src=src, group=self.group)
            step = torch.tensor([float(self.opt.state[fl["params"][0]]["step"])], device=fl["p"].device)
            self.dist.broadcast(step, src=src, group=self.group)
            for p in fl["params"]:
                self.opt.state[p]["step"] = torch.tensor(float(step.item()))
        bump_weights_epoch()

    def replica_checksum(self):
        """(max - min) over ranks of the fp64 sum of the flat parameter buffer: 0.0 iff the replicas agree."""
        tot = torch.stack([fl["p"].double().sum() for fl in self.opt.ensure_flat()]).sum().reshape(1)
        if self.world <= 1:
            return 0.0
        hi, lo = tot.clone(), tot.clone()
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX, group=self.group)
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN, group=self.group)
        return float((hi - lo).item())

    def _setup(self):
        flats = self.opt.ensure_flat()
        if self._flat_id == id(flats):
            return
        for h in self._hooks:
            h.remove()
        self._hooks, self._buckets = [], []
        for fl in flats:
            n = fl["p"].numel()
            ends = list(fl["offs"][1:]) + [n]
            # buckets from the end of the buffer
            cur_hi, members = n, []
            items = list(zip(range(len(fl["params"])), fl["params"], fl["offs"], ends))
            idxs = []
            for i, p, lo, hi in reversed(items):
                members.append(p)
                idxs.append(i)
                if cur_hi - lo >= self.bucket_elems or lo == 0:
                    self._buckets.append(dict(view=fl["g"][lo:cur_hi], params=members, fl=fl, idxs=idxs, pending=0, handle=None,
                                              staged=False))
                    cur_hi, members, idxs = lo, [], []
        for bi, b in enumerate(self._buckets):
            for p in b["params"]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self._flat_id = id(flats)

    def _make_hook(self, bi):
        def hook(_param):
            b = self._buckets[bi]
            b["pending"] -= 1
            if b["pending"] == 0:
                self._stage(b)                    # copy now (overlaps backward) ...
                b["ready"] = True
                self._launch_ready()              # ... launch as soon as every earlier bucket has been launched
        return hook

    def _stage(self, b):
        if not b["staged"]:                       # the bucket's gradients -> its slice of the flat buffer (one multi-tensor copy)
            self.opt.stage_grads(b["fl"], b["idxs"])
            b["staged"] = True

    def _launch_ready(self, force=False):
        """Issue the all-reduces of the leading run of ready buckets (all remaining ones with force=True), in list order."""
        while self._next < len(self._buckets):
            b = self._buckets[self._next]
            if not (b["ready"] or force):
                return
            self._stage(b)
            if self.world > 1 and b["handle"] is None:
                b["handle"] = self.dist.all_reduce(b["view"], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.launch_order.append(self._next)
            self._next += 1

    def prepare(self):
        """Call before backward (after zero_grad)."""
        self._setup()
        for b in self._buckets:
            b["pending"] = len(b["params"])
            b["handle"] = None
            b["staged"] = False
            b["ready"] = False
        self._next = 0
        self.launch_order = []                    # bucket indices in the order their collectives were issued (tests)

    def finish(self):
        """Call after backward, before optimizer.step()."""
        self._launch_ready(force=True)
        timed = self.profile and self.world > 1 and torch.cuda.is_available() and self._buckets and self._buckets[0]["view"].is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for b in self._buckets:
            if b["handle"] is not None:
                b["handle"].wait()
                b["handle"] = None
        if timed:
            e1.record()
            self._wait_events.append((e0, e1))
        self.opt._staged = True

    def describe(self):
        """Bucket layout of the exchange: count and bytes (after the first prepare())."""
        if not self._buckets:
            return {"buckets": 0, "bytes": []}
        return {"buckets": len(self._buckets), "bytes": [int(b["view"].numel()) * 4 for b in self._buckets]}

    def exposed_ms(self, last=None):
        """Per-step milliseconds the compute stream waited in finish() (profile = True), newest `last` steps."""
        ev = self._wait_events[-last:] if last else self._wait_events
        if not ev:
            return None
        torch.cuda.synchronize()
        return [e0.elapsed_time(e1) for e0, e1 in ev]
