#!/usr/bin/env python
"""bench.py - headline metric of BASELINE.json on MI355X.

  metric   : 40-frame 128x128 videos/sec (DDIM-100)          [BASELINE.json "metric"]
  workload : configs[1] = MUG 128x128, 40-frame DDIM-100 sample, batch=1 per GPU
  a "step" : one FlowDiffusion.sample_one_video() call = LFAE encode + 100 UNet/sampler steps
             + 40-frame LFAE decode, on synthetic random-init weights / inputs (no network).

`python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches one rank per GPU with
torch.distributed.run (sampling shards by video: no data-path collective, "weak" scaling).
Rank 0 prints ONE JSON line with `roofline` (dominant kernel = fp32-MFMA implicit-GEMM conv) and
`cpu_baseline` (the CPU oracle timed on this host on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO_ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO_ROOT)
sys.path.insert(0, os.path.join(REPO_ROOT, "tests"))

WORKLOAD = dict(name="MUG 128x128, 40-frame DDIM-100 sample, batch=1 per GPU (BASELINE.json configs[1])",
                batch=1, frames=40, latent=32, image=128, sampling_timesteps=100, timesteps=1000)
# fp32 matrix peak of MI355X (MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s)
FP32_MFMA_PEAK_TFLOPS = 157.3
# reference-dataflow algorithmic work per C2 video: 100*235.10 + 5.14 + 40*24.78 GFLOP
GFLOP_PER_VIDEO_REFERENCE = 24506.0
# one training step of the unmodified reference per 40-frame 128x128 video, counted by torch.utils.flop_counter on the CPU run
# (oracle/make_golden.py --train-flops): 2423.2 forward (per-frame pseudo ground truth incl. the encoder it re-runs per frame,
# UNet, logged decode of the denoised flow) + 403.7 backward GFLOP
TRAIN_GFLOP_PER_VIDEO_REFERENCE = 2826.9


def log(msg):
    """Progress on stderr (stdout carries exactly ONE JSON line)."""
    print("[bench %7.1f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def _read_num(path):
    try:
        with open(path) as f:
            return float(f.read().split()[0])
    except Exception:
        return None


def _hwmon_dirs(local):
    """hwmon directories of the amdgpu devices in sysfs; the one whose PCI address matches torch's device `local` first."""
    import glob
    dirs = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    want = None
    try:
        pr = torch.cuda.get_device_properties(local)
        want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        pass
    if want:
        hit = [d for d in dirs if want in os.path.realpath(os.path.join(d, "..", ".."))]
        if hit:
            return hit[:1], True
    return dirs[:8], False


def gpu_state(local=0):
    """One snapshot of the GPU's clocks / power / temperature: sysfs hwmon when the box exposes it (file reads, no subprocess),
    else `rocm-smi --json`.  Never raises: the box state is a diagnostic beside the timing, not part of it."""
    try:
        dirs, matched = _hwmon_dirs(local)
        rows = []
        for d in dirs:
            row = {"sclk_mhz": _read_num(os.path.join(d, "freq1_input")), "mclk_mhz": _read_num(os.path.join(d, "freq2_input")),
                   "power_w": _read_num(os.path.join(d, "power1_average")) or _read_num(os.path.join(d, "power1_input")),
                   "temp_c": _read_num(os.path.join(d, "temp1_input"))}
            for k, div in (("sclk_mhz", 1e6), ("mclk_mhz", 1e6), ("power_w", 1e6), ("temp_c", 1e3)):
                if row[k] is not None:
                    row[k] = round(row[k] / div, 1)
            if any(v is not None for v in row.values()):
                rows.append(row)
        if rows:
            return {"source": "sysfs hwmon" + ("" if matched else " (device not identified: all visible cards)"),
                    "cards": rows if not matched else None, **(rows[0] if matched or len(rows) == 1 else {})}
    except Exception:
        pass
    try:
        import subprocess
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, timeout=20, text=True)
        j = json.loads(r.stdout)
        card = j.get("card%d" % local) or next(iter(j.values()))
        keep = {k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "mclk", "power", "temperature (sensor junction)"))}
        return {"source": "rocm-smi", **keep}
    except Exception as e:
        return {"source": "unavailable", "error": "%s" % type(e).__name__}


class StateSampler:
    """Samples gpu_state() from a thread every `period` s while a timed block runs (sysfs only - a subprocess per sample would
    perturb the host); summary() = min / median / max of the shader clock and the power seen DURING the block."""

    def __init__(self, local, period=0.25):
        import threading
        self.local, self.period, self.rows, self._stop = local, period, [], threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self.enabled = gpu_state(local).get("source", "").startswith("sysfs")

    def _run(self):
        while not self._stop.wait(self.period):
            self.rows.append(gpu_state(self.local))

    def __enter__(self):
        if self.enabled:
            self._thread.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self.enabled:
            self._thread.join(timeout=2)

    def summary(self):
        out = {"samples": len(self.rows)}
        for key in ("sclk_mhz", "power_w", "temp_c", "mclk_mhz"):
            v = sorted(r[key] for r in self.rows if r.get(key) is not None)
            if v:
                out[key] = {"min": v[0], "median": v[len(v) // 2], "max": v[-1]}
        return out


def dist_setup(n_gpus):
    """One process per GPU; returns (rank, world, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        # "nccl" = RCCL over xGMI on the GPU box; LFDM_DIST_BACKEND=gloo lets the N>1 path be exercised with several
        # ranks sharing ONE GPU (RCCL refuses duplicate devices) - a test hook, never used by the driver
        backend = os.environ.get("LFDM_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        # stdout carries exactly one line - rank 0's JSON: whatever a backend's C++ side prints while connecting (gloo's
        # "[Gloo] Rank 0 is connected to ..." goes to fd 1) is sent to stderr
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend=backend)
            if backend == "gloo":
                dist.barrier()      # (gloo connects its pairs lazily: do it while fd 1 is diverted)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    return rank, world, local


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (backend nccl = RCCL
    over xGMI), the same command line; rank 0's ONE JSON line passes through on stdout.  With fewer visible GPUs than ranks (a
    one-GPU test box) the ranks share devices and the collective backend falls back to gloo (RCCL refuses duplicate devices)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < n_gpus:
        env.setdefault("LFDM_DIST_BACKEND", "gloo")
        log("self-launch: %d ranks on %d visible GPU(s): ranks share devices, backend gloo (NOT a scaling measurement)" % (n_gpus, ndev))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("self-launch: " + " ".join(cmd))
    return subprocess.run(cmd, env=env).returncode


def timed_region(run_step, steps, warmup, world, device_sync, per_rank=None):
    """W untimed + exactly K timed steps, bracketed by barrier + device sync; returns max-over-ranks seconds
    (per_rank, if a list, receives every rank's own seconds)."""
    import torch.distributed as dist
    for _ in range(warmup):
        run_step()
    device_sync()
    if world > 1:
        dist.barrier()
    device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        run_step()
    device_sync()
    if world > 1:
        dist.barrier()
    device_sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if per_rank is not None:
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            per_rank[:] = [float(v.item()) for v in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    elif per_rank is not None:
        per_rank[:] = [elapsed]
    return elapsed


def sampler_step_convs(model):
    """The lfdm_conv2d_cl_f32 launches of ONE sampler step (stem excluded: it is not a convolution launch), collected from
    an eager run of the loop the sampler captures (LFDM_NO_GRAPH=1): the parameter structs stay valid afterwards because
    every pointer in them is a weight pack or one of the executor's persistent arenas."""
    from cvpr23_lfdm_amd import ops
    records, state = [], {"step": 0}
    orig_conv, orig_step = ops.conv_launch, ops.sampler_step

    def conv_launch(p):
        if state["step"] == 0:
            records.append(p)
        orig_conv(p)

    def sampler_step(*a, **k):
        state["step"] += 1
        return orig_step(*a, **k)

    os.environ["LFDM_NO_GRAPH"] = "1"
    ops.conv_launch, ops.sampler_step = conv_launch, sampler_step
    try:
        model.sample_one_video(cond_scale=1.0)
    finally:
        ops.conv_launch, ops.sampler_step = orig_conv, orig_step
        os.environ.pop("LFDM_NO_GRAPH", None)
    torch.cuda.synchronize()
    # before the loop: the LFAE encoder's convolutions and the per-video `fea` term (7x7 over the 256 feature channels) - once
    # per video, not part of a step
    fea = [i for i, p in enumerate(records) if p.kh == 7 and p.c0 == 256]
    assert len(fea) == 1, "unexpected launch order"
    return records[fea[0] + 1:]


def conv_work(p):
    """(algorithmic direct-form FLOPs, algorithmic bytes = input once + filters once + output once) of one launch."""
    probs = 4 if p.deconv4 else 1                                       # deconv4: four parity problems in the launch
    kdim = (p.c0 + p.c1) // max(1, p.groups) * p.kh * p.kw
    m = p.n_img * p.hq * p.wq
    return (2.0 * m * p.cout * kdim * probs,
            4.0 * (p.n_img * p.hi * p.wi * (p.c0 + p.c1) + kdim * p.cout * probs + p.n_img * p.ho * p.wo * p.cout))


def graph_time(launch_all, replays=30):
    """Average duration of a launch sequence under the conditions of the real step: captured once, replayed back to back."""
    launch_all()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        launch_all()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / replays


def in_situ_family_us(model, is_member, ms_per_video_full, videos=3):
    """Time one launch family INSIDE the real captured step, by subtraction: the sampler is made to capture its step graph
    once more with that family's launches left out (every other kernel, buffer and dependency unchanged; the samples are
    garbage and discarded), and (video time with) - (video time without) over the 100 steps is what the family costs where
    it actually runs - inputs still warm from the producing kernel, filters cold.  Returns microseconds per step."""
    from cvpr23_lfdm_amd import ops
    orig = ops.conv_launch
    ops.conv_launch = lambda p: None if is_member(p) else orig(p)
    dif = model.diffusion
    saved_plans, dif._plans = dif._plans, {}
    try:
        model.sample_one_video(cond_scale=1.0)          # captures the reduced graph
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(videos):
            model.sample_one_video(cond_scale=1.0)
        torch.cuda.synchronize()
        ms_without = 1e3 * (time.perf_counter() - t0) / videos
    finally:
        ops.conv_launch = orig
        dif._plans = saved_plans
    return (ms_per_video_full - ms_without) * 1e3 / WORKLOAD["sampling_timesteps"], ms_without


def evidence_files():
    """The committed rocprofv3 evidence of the current build: profiles/LATEST holds the tag of the last tools/gpu_final.sh run
    (<tag>_kernel_stats.txt, <tag>_step_sequence.txt) and, on its second line, the counter file of tools/prof_step_pmc.sh."""
    out = {"kernel_stats": None, "step_sequence": None, "pmc": None}
    latest = os.path.join(REPO_ROOT, "profiles", "LATEST")
    if not os.path.exists(latest):
        return out
    lines = open(latest).read().split()
    tag = lines[0] if lines else None
    for key, name in (("kernel_stats", "%s_kernel_stats.txt" % tag), ("step_sequence", "%s_step_sequence.txt" % tag)):
        if tag and os.path.exists(os.path.join(REPO_ROOT, "profiles", name)):
            out[key] = "profiles/" + name
    for name in lines[1:2] + ["r02_traffic.json"]:
        if os.path.exists(os.path.join(REPO_ROOT, "profiles", name)):
            out["pmc"] = "profiles/" + name
            break
    return out


def conv_roofline(model, ms_per_sampler_step):
    """Roofline of the dominant kernel, measured live: the Winograd convolution launches (conv_wino_kernel) of one sampler
    step are re-captured as their own hipGraph - same parameter structs, same arenas, 40 different filter sets so the
    weights are as cold as in the real step - and replayed; achieved = sum of their ALGORITHMIC (direct-form) FLOPs / that
    time.  The direct-form schedules (conv_ksw / conv_igemm + split-K reduce) get the same treatment as a second row."""
    from cvpr23_lfdm_amd import ops
    convs = sampler_step_convs(model)
    fam = {"winograd": [], "direct": []}
    for p in convs:
        fam["winograd" if (p.weight_wino and p.kh == 3 and os.environ.get("LFDM_WINO", "1") != "0") else "direct"].append(p)
    rows = {}
    for name, ps in fam.items():
        if not ps:
            continue
        sec = graph_time(lambda ps=ps: [ops.conv_launch(p) for p in ps])
        flops = sum(conv_work(p)[0] for p in ps)
        nbytes = sum(conv_work(p)[1] for p in ps)
        rows[name] = {"launches": len(ps), "us_per_step": round(sec * 1e6, 1), "us_per_launch": round(sec * 1e6 / len(ps), 2),
                      "gflop_per_step": round(flops / 1e9, 2), "tflops": round(flops / sec / 1e12, 2),
                      "frac": round(flops / sec / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4), "algorithmic_bytes_per_step": round(nbytes)}
    # in-situ (the headline roofline number): the same launches timed inside the real step, by subtraction
    key = lambda p: (p.src0, p.out, p.weight_wino, p.cout)            # (the LFAE's Winograd launches before / after the loop stay)
    members = {key(p) for p in fam["winograd"]}
    is_wino = lambda p: key(p) in members
    ms_video = ms_per_sampler_step * WORKLOAD["sampling_timesteps"]
    t0 = time.perf_counter()
    model.sample_one_video(cond_scale=1.0)
    model.sample_one_video(cond_scale=1.0)
    torch.cuda.synchronize()
    ms_video_now = 1e3 * (time.perf_counter() - t0) / 2          # same un-barriered host timing as the subtraction run
    wino_us, ms_without = in_situ_family_us(model, is_wino, ms_video_now)
    w = dict(rows["winograd"])
    w_iso = rows["winograd"]
    w.update(us_per_step=round(wino_us, 1), us_per_launch=round(wino_us / w["launches"], 2),
             tflops=round(w["gflop_per_step"] / wino_us * 1e3, 2), frac=round(w["gflop_per_step"] / wino_us * 1e3 / FP32_MFMA_PEAK_TFLOPS, 4))
    rows["winograd_in_situ"] = {"us_per_step": w["us_per_step"], "us_per_launch": w["us_per_launch"], "tflops": w["tflops"], "frac": w["frac"],
                                "ms_per_video_with": round(ms_video_now, 2), "ms_per_video_without": round(ms_without, 2)}
    executed = w["gflop_per_step"] * 16.0 / 36.0
    prof = evidence_files()
    # PMC counters cannot be read from inside the process: the committed rocprofv3 --pmc passes of this build (tools/prof_step_pmc.sh)
    traffic, traffic_src, step_traffic, direct_traffic, traffic_lower, traffic_build = None, None, None, None, None, None
    if prof["pmc"]:
        with open(os.path.join(REPO_ROOT, prof["pmc"])) as f:
            tj = json.load(f)
        traffic_build = tj.get("build", "unrecorded (taken before round 4)")
        traffic, traffic_src = tj.get("wino_bytes_per_step"), prof["pmc"] + ": " + tj.get("source", "")
        direct_traffic = tj.get("direct_bytes_per_step")
        wf = tj.get("families", {}).get("winograd", {})
        # FETCH_SIZE counts 64 B per request: whole-line (128 B) requests need the guide's x2, isolated 64-byte segments (the patch gathers)
        # are counted exactly (profiles/r03_x_fetch_calib.txt) - the x2 figure above is an UPPER bound for a kernel that mixes both
        traffic_lower = round(wf["fetch_bytes"] / 2 + wf["write_bytes"]) if wf.get("fetch_bytes") else None
        if tj.get("step_bytes"):
            step_traffic = {"bytes": tj["step_bytes"], "fetch_bytes": tj.get("step_fetch_bytes"), "write_bytes": tj.get("step_write_bytes"),
                            "launches": tj.get("launches_per_step"), "mfma_util": tj.get("step_mfma_util"),
                            "unit": "HBM-side bytes of ALL launches of one sampler step (FETCH_SIZE x2 + WRITE_SIZE); mfma_util = "
                                    "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE)", "file": prof["pmc"]}
    # cross-check against the rocprofv3 kernel table + dispatch sequence committed for this build (tools/gpu_final.sh -> profiles/<LATEST>_*)
    n_split = sum(1 for p in fam["winograd"] if ops.conv_plan(p)[1] > 1)
    rocprof = None
    if prof["kernel_stats"]:
        # all conv_wino_kernel<false, ...> rows of the table: the plain instantiation and - since round 5 - the one that reduces its split-K
        # slabs inside the launch (..., true>): calls / total_us are the 6th / 5th fields from the end
        calls, total = 0, 0.0
        for ln in open(os.path.join(REPO_ROOT, prof["kernel_stats"])):
            f = ln.split()
            if ln.startswith("conv_wino_kernel<false") and len(f) >= 7:
                calls, total = calls + int(f[-6]), total + float(f[-5])
        if calls:
            avg = total / calls
            rocprof = {"file": prof["kernel_stats"], "avg_us_per_launch": round(avg, 2),
                       "tflops_from_avg": round(w["gflop_per_step"] / (avg * w["launches"]) * 1e3, 2),
                       "frac_from_avg": round(w["gflop_per_step"] / (avg * w["launches"]) * 1e3 / FP32_MFMA_PEAK_TFLOPS, 4),
                       "what": "the conv_wino_kernel<false, ...> rows of the table (call-weighted): the kernel as launched - with the split-K "
                               "slabs reduced inside the launch where the sequence below shows no conv_splitk_reduce behind it"}
    if rocprof and prof["step_sequence"]:
        # the dispatch sequence of ONE sampler step from the same rocprofv3 trace: exactly the step's launches (the table's rows also hold the
        # LFAE encoder / decoder launches of the same instantiation, which are larger) - the figure that must agree with the live in-situ one
        red_us, red_n, prev, wn, wus = 0.0, 0, "", 0, 0.0
        for ln in open(os.path.join(REPO_ROOT, prof["step_sequence"])):
            f = ln.split()
            if len(f) < 5 or not f[0].isdigit():
                continue
            if f[1].startswith("conv_wino_kernel"):
                wn, wus = wn + 1, wus + float(f[-2])
            if f[1].startswith("conv_splitk_reduce") and prev.startswith("conv_wino_kernel"):      # (incl. conv_splitk_reduce_vec_kernel<KS>)
                red_us, red_n = red_us + float(f[-2]), red_n + 1
            prev = f[1]
        if wn:
            rocprof.update(table_avg_us_per_launch=rocprof["avg_us_per_launch"], avg_us_per_launch=round(wus / wn, 2), launches_in_sequence=wn,
                           tflops_from_avg=round(w["gflop_per_step"] / wus * 1e3, 2),
                           frac_from_avg=round(w["gflop_per_step"] / wus * 1e3 / FP32_MFMA_PEAK_TFLOPS, 4),
                           what="avg_us_per_launch = the conv_wino_kernel launches of ONE sampler step in the committed dispatch sequence "
                                "(rocprofv3 --kernel-trace); table_avg_us_per_launch = the call-weighted conv_wino_kernel<false, ...> rows of the "
                                "kernel table, which also hold the larger LFAE encoder / decoder launches")
        with_red = (wus if wn else avg * w["launches"]) + red_us
        rocprof.update(split_k_reduce_launches=red_n, split_k_reduce_us_per_step=round(red_us, 1), sequence_file=prof["step_sequence"],
                       frac_with_reduce_passes=round(w["gflop_per_step"] / with_red * 1e3 / FP32_MFMA_PEAK_TFLOPS, 4))
    try:
        sat = wino_saturated(next(model.unet.parameters()).device)
    except Exception as e:                       # an extra, never a reason to lose the bench line
        sat = {"error": repr(e)}
    all_flops = sum(rows[k]["gflop_per_step"] for k in ("winograd", "direct") if k in rows)
    all_us = sum(rows[k]["us_per_step"] for k in ("winograd", "direct") if k in rows)
    if direct_traffic and "direct" in rows:
        rows["direct"]["traffic"] = direct_traffic
        rows["direct"]["traffic_over_algorithmic"] = round(direct_traffic / max(1, rows["direct"]["algorithmic_bytes_per_step"]), 2)
    return {"bound": "mfma", "kernel": "conv_wino_kernel (3x3 convolutions as Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32) - the %d launches of one "
                                       "sampler step; %d of them split K - their slabs are reduced inside the launch by the workgroup that draws a tile's last "
                                       "ticket (conv_wino.hip FUSE), or by a conv_splitk_reduce_kernel pass where LFDM_WINO_FUSE_REDUCE=0" % (w["launches"], n_split),
            "achieved": w["tflops"], "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": w["frac"],
            "frac_definition": "direct-form FLOPs of the family / in-situ time of the launches incl. their split-K reduction (in-launch, or the reduce "
                               "passes where there are any) / peak; `rocprofv3` carries the figure from the committed kernel table",
            "traffic": traffic, "traffic_unit": "HBM-side bytes per sampler step over the same launches incl. their reduce passes (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE passes, not live)",
            "traffic_source": traffic_src, "traffic_build": traffic_build,
            "traffic_is_of_this_build": bool(traffic_build) and traffic_build == __import__("cvpr23_lfdm_amd._build", fromlist=["x"]).source_fingerprint(),
            "algorithmic_bytes_per_step": w["algorithmic_bytes_per_step"],
            "traffic_over_algorithmic": round(traffic / max(1, w["algorithmic_bytes_per_step"]), 2) if traffic else None,
            "traffic_lower_bound": traffic_lower,
            "traffic_bounds_note": "traffic = FETCH_SIZE x2 + WRITE_SIZE (the guide's correction, exact for whole-line requests); traffic_lower_bound = FETCH_SIZE x1 "
                                   "+ WRITE_SIZE (exact for isolated 64-byte segments such as the patch gathers); calibration: profiles/r03_x_fetch_calib.txt",
            "step_traffic": step_traffic,
            "launches": w["launches"], "split_k_launches": n_split, "us_per_launch": w["us_per_launch"], "gflop_per_step": w["gflop_per_step"],
            "executed_mfma_tflops": round(executed / (w["us_per_step"] * 1e-6) / 1e3, 2),
            "executed_mfma_frac": round(executed / (w["us_per_step"] * 1e-6) / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4),
            "families": rows,
            "all_convolutions": {"gflop_per_step": round(all_flops, 2), "us_per_step": round(all_us, 1),
                                 "tflops": round(all_flops / all_us * 1e3, 2), "frac": round(all_flops / all_us * 1e3 / FP32_MFMA_PEAK_TFLOPS, 4)},
            "sampler_step_us": round(ms_per_sampler_step * 1e3, 1),
            "isolated_replay": {"us_per_launch": w_iso["us_per_launch"], "tflops": w_iso["tflops"], "frac": w_iso["frac"]},
            "rocprofv3": rocprof,
            "kernel_at_saturating_size": sat,
            "note": "achieved / frac count the reference's direct-form FLOPs (SURVEY.md 8d); the Winograd launches execute 16/36 of theirs "
                    "on the matrix pipe (executed_*; kernel_at_saturating_size.frac can exceed 1 for that reason - read its executed_mfma_frac).  "
                    "us_per_launch = in situ: (video time of the captured step) - (video time of the same step captured without these launches "
                    "and their reduce passes), / 100 steps / launches; the rocprofv3 kernel table's average is the kernel alone and is lower by the "
                    "reduce passes (rocprofv3.split_k_reduce_us_per_step); `isolated_replay` = the same launches replayed alone as a hipGraph "
                    "(inputs cold: slower)"}


def wino_saturated(dev):
    """The same Winograd kernel where its launch is not floor-bound: the LFAE bottleneck convolution of a B = 8 training step
    (3x3, 256 -> 256 channels, 320 frames of 32x32 = 386.5 GFLOP direct form, 10 240 workgroups), timed with events around
    back-to-back launches - what the kernel does when every CU has work for the whole launch."""
    from cvpr23_lfdm_amd import ops
    n_img, hw, c = 320, 32, 256
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n_img * hw * hw, c, generator=g).to(dev)
    w = (torch.randn(c, c, 3, 3, generator=g) * 0.02).to(dev)
    bias = torch.zeros(c, device=dev)
    ww = ops.pack_wino_weight(w)
    out = torch.empty_like(x)
    fn = lambda: ops.conv2d_cl(x, None, c, 3, 3, n_img, hw, hw, bias=bias, weight_wino=ww, out=out)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    gflop = 2.0 * n_img * hw * hw * c * c * 9 / 1e9
    return {"shape": "3x3 conv 256->256, 320 frames of 32x32 (LFAE bottleneck of a B=8 training step)", "us_per_launch": round(us, 1),
            "gflop": round(gflop, 1), "tflops": round(gflop / us * 1e3, 1), "frac": round(gflop / us * 1e3 / FP32_MFMA_PEAK_TFLOPS, 4),
            "executed_mfma_tflops": round(gflop * 16 / 36 / us * 1e3, 1),
            "executed_mfma_frac": round(gflop * 16 / 36 / us * 1e3 / FP32_MFMA_PEAK_TFLOPS, 4)}


def warp_bench(model, img, iters=20):
    """BASELINE.json's second metric, "LFAE warp Gpix/s": the six deform_input/apply_optical launches of one
    40-frame decode (flow = identity + 0.1 randn clipped to [-1.2,1.2], occ = rand).
    Gpix/s = 1e-9 * sum(C*H*W*frames) / time (channel-pixels); HBM roofline on the algorithmic bytes of 8d:
    8 B/elem for a pure warp (gathered input counted once + output), 12 B/elem for warp+blend (prev read)."""
    from cvpr23_lfdm_amd import ops
    gen = model.generator
    b, t, s = img.shape[0], WORKLOAD["frames"], WORKLOAD["latent"]
    g = torch.Generator(device="cpu").manual_seed(99)
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, s), torch.linspace(-1, 1, s), indexing="ij")
    ident = torch.stack((xs, ys), 0).view(1, 2, 1, s, s)
    flow = (ident + 0.1 * torch.randn(b, 2, t, s, s, generator=g)).clamp_(-1.2, 1.2)
    maps = torch.cat((flow, torch.rand(b, 1, t, s, s, generator=g)), 1).contiguous().to(img.device)
    fx, fy, occ = maps[:, 0], maps[:, 1], maps[:, 2]
    wk = dict(fh=s, fw=s, fsb=3 * t * s * s, fst=s * s)
    with torch.no_grad():
        skips = gen.encode(img)
    n = b * t
    h = img.shape[2]
    prevs = {k: torch.rand(n * r * r, c, device=img.device) for k, (r, c) in
             {"l1": (s, skips[2].shape[1]), "u0": (h // 2, skips[1].shape[1]), "u1": (h, skips[0].shape[1]), "rgb": (h, 4)}.items()}
    outs = {"lat": torch.empty(n * s * s, skips[2].shape[1], device=img.device), "l1": torch.empty_like(prevs["l1"]),
            "u0": torch.empty_like(prevs["u0"]), "u1": torch.empty_like(prevs["u1"]),
            "def": torch.empty(b, 3, t, h, h, device=img.device), "pred": torch.empty(b, 3, t, h, h, device=img.device)}

    def one_decode_warps():
        ops.warp_planar(img, t, fx, fy, None, s, s, wk["fsb"], wk["fst"], out=outs["def"])
        ops.warp_cl(skips[2], b, t, s, s, fx, fy, occ, out=outs["lat"], **wk)
        ops.warp_cl(skips[2], b, t, s, s, fx, fy, occ, prev=prevs["l1"], out=outs["l1"], **wk)     # (a34: the latent-resolution skip again, blended)
        ops.warp_cl(skips[1], b, t, h // 2, h // 2, fx, fy, occ, prev=prevs["u0"], out=outs["u0"], **wk)
        ops.warp_cl(skips[0], b, t, h, h, fx, fy, occ, prev=prevs["u1"], out=outs["u1"], **wk)
        ops.warp_planar(img, t, fx, fy, occ, s, s, wk["fsb"], wk["fst"], prev=prevs["rgb"][:, :3], prev_is_cl=True,
                        out=outs["pred"])

    # reference frame count: deform_input x6 per frame (the (3,128,128) source warp appears twice in a34)
    pure = n * (3 * h * h + skips[2].shape[1] * s * s)
    blend = n * (skips[2].shape[1] * s * s + skips[1].shape[1] * (h // 2) ** 2 + skips[0].shape[1] * h * h + 3 * h * h)
    elems, nbytes = pure + blend, 8 * pure + 12 * blend
    one_decode_warps()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        one_decode_warps()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / iters
    gbs = nbytes / sec / 1e9
    traffic, pmc, pmc_launches = None, evidence_files()["pmc"], None
    if pmc:
        with open(os.path.join(REPO_ROOT, pmc)) as f:
            tj = json.load(f)
        traffic, pmc_launches = tj.get("warp_bytes_per_video"), tj.get("warp_launches", 5)
    return {"value": round(elems / sec / 1e9, 2), "unit": "Gpix/s (channel-pixels, 6 launches = all warps of one 40-frame decode)",
            "us_per_video": round(sec * 1e6, 1), "elements": elems,
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(gbs / 8000.0, 4), "traffic": traffic, "algorithmic_bytes": nbytes,
                         "frac_of_counter_bytes": round(traffic / sec / 1e9 / 8000.0, 4) if traffic else None,
                         "traffic_source": "%s: rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE over the %s warp launches of one sample_one_video decode, not live"
                                           % (pmc, pmc_launches)}}


def dp_diagnostics(dp, steps, world):
    """N > 1 self-diagnosis of a data-parallel training leg (every rank calls it): the exchange's bucket layout, the exposed all-reduce time
    of EVERY rank (all-gathered) and the replica checksum after the timed steps (0.0 = the replicas hold identical parameters)."""
    import torch.distributed as dist
    exposed = dp.exposed_ms(last=steps) or []
    mine = sum(exposed) / max(1, len(exposed))
    worst = max(exposed) if exposed else 0.0
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([mine, worst], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    return dict(dp.describe(), exposed_allreduce_ms_per_step=round(mine, 3), exposed_allreduce_ms_max=round(worst, 3) if exposed else None,
                exposed_allreduce_ms_per_step_per_rank=[round(float(v[0]), 3) for v in every],
                exposed_allreduce_ms_max_per_rank=[round(float(v[1]), 3) for v in every],
                replica_checksum=dp.replica_checksum(), backend=dist.get_backend(),
                note="exposed = time the compute stream waited in GradAllReduce.finish() (events around the bucket waits); the un-suffixed figures are "
                     "rank 0's; replica_checksum = (max - min) over ranks of the fp64 parameter sum after the timed steps (0.0 = replicas identical)")


def train_bench(dev, rank, world, steps, warmup, batch, lazy_extra=True):
    """BASELINE.json configs[3] per GPU: one DM training step = frozen-LFAE pseudo ground truth of all B*T frames +
    UNet forward/backward (native kernels under autograd) + [RCCL gradient all-reduce when world > 1] + fused Adam,
    on `batch` 40-frame 128x128 videos per GPU (the reference script uses 64 videos over 8 GPUs)."""
    import contextlib
    import synth
    from cvpr23_lfdm_amd import FlowDiffusion
    torch.manual_seed(4321)
    with contextlib.redirect_stdout(sys.stderr):
        m = FlowDiffusion(img_size=WORKLOAD["latent"], num_frames=WORKLOAD["frames"], sampling_timesteps=1000,
                          null_cond_prob=0.1, is_train=True, lr=1e-4, config_pth=synth.CONFIG, pretrained_pth="")
    m.unet.load_state_dict(synth.unet_state())
    m.generator.load_state_dict(synth.generator_state())
    m.region_predictor.load_state_dict(synth.region_state())
    m.bg_predictor.load_state_dict(synth.bg_state())
    for net in (m.generator, m.region_predictor, m.bg_predictor):
        net.eval()
        m.set_requires_grad(net, False)
    m.to(dev)
    m.enable_data_parallel(shard_inputs=False)      # weak scaling: every rank draws its own per-GPU batch
    ref_img, real_vid, cond, _, _ = synth.train_inputs(batch, WORKLOAD["frames"], WORKLOAD["image"], seed=100 + rank)
    m.set_train_input(ref_img=ref_img.to(dev), real_vid=real_vid.to(dev), ref_text=cond.to(dev))
    losses = []

    def step():
        m.optimize_parameters()
        losses.append(m.loss.detach())

    if m._dp is not None:
        m._dp.profile = True
    per_rank = []
    elapsed = timed_region(step, steps, warmup, world, torch.cuda.synchronize, per_rank)
    vals = [float(v) for v in losses]
    comm = None
    if m._dp is not None:       # what the gradient exchange looks like and how much of it backward did not hide, per rank
        comm = dp_diagnostics(m._dp, steps, world)
        m._dp.profile = False
    # not the headline: the same step with the pseudo-ground-truth decode (real_out_vid / real_warped_vid: read by no loss, only by
    # the scripts' sample images) deferred until it is read (FlowDiffusion.lazy_real_decode)
    lazy_steps, lazy_elapsed = max(2, steps // 2), None
    if lazy_extra:
        m.lazy_real_decode = True
        lazy_elapsed = timed_region(step, lazy_steps, 1, world, torch.cuda.synchronize)
        m.lazy_real_decode = False
    return {"value": round(steps * batch * world / elapsed, 3), "unit": "training videos/s (40 frames, 128x128)",
            "ms_per_step": round(1e3 * elapsed / steps, 1), "batch_per_gpu": batch, "global_batch": batch * world,
            "steps": steps, "warmup": warmup, "ms_per_step_per_rank": [round(1e3 * v / steps, 1) for v in per_rank],
            "allreduce": comm, "grad_allreduce": ("%s, bucketed, overlapped with backward" % __import__("torch.distributed").distributed.get_backend()) if world > 1 else "none (1 GPU)",
            "gflop_per_video_reference_dataflow": TRAIN_GFLOP_PER_VIDEO_REFERENCE,
            "tflops_reference_dataflow": round(steps * batch * world / elapsed * TRAIN_GFLOP_PER_VIDEO_REFERENCE / 1e3, 1),
            "ms_per_step_lazy_real_decode": round(1e3 * lazy_elapsed / lazy_steps, 1) if lazy_elapsed else None,
            "loss_first": round(vals[0], 5), "loss_last": round(vals[warmup + steps - 1], 5),
            "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}


def cpu_baseline():
    """The CPU oracle (oracle/lfdm_oracle.py, a torch-CPU port of the reference dataflow; the reference tree itself is not on
    the GPU box) on this host, bounded sample: the thread count is swept over {8, 16, 32, 64, all} on one UNet forward at the
    C2 shape, the best setting is then timed on 3 forwards + compute_fea + 3 decoded frames, and extrapolated linearly to
    one video (per-step cost is step independent)."""
    sys.path.insert(0, os.path.join(REPO_ROOT, "oracle"))
    import lfdm_oracle as O
    import synth
    torch.manual_seed(0)
    dsd = {"denoise_fn." + k: v for k, v in synth.unet_state().items()}
    gsd = synth.generator_state()
    b, t, s, hw = WORKLOAD["batch"], WORKLOAD["frames"], WORKLOAD["latent"], WORKLOAD["image"]
    img, cond = synth.inputs(b, hw)
    x = torch.randn(b, 259, t, s, s)
    tt = torch.full((b,), 500, dtype=torch.long)
    ncpu = os.cpu_count() or 1
    sweep = {}
    budget_t0 = time.perf_counter()
    xs = x[:, :, :8].contiguous()               # the thread-count sweep runs on 8 of the 40 frames (bounded CPU time)
    with torch.no_grad():
        O.unet_forward(dsd, xs, tt, cond)       # warm-up (thread pools, allocator)
        for n in sorted({min(v, ncpu) for v in (8, 16, 32, 64, ncpu)}):
            torch.set_num_threads(n)
            O.unet_forward(dsd, xs, tt, cond)
            t0 = time.perf_counter()
            O.unet_forward(dsd, xs, tt, cond)
            sweep[n] = time.perf_counter() - t0
            log("cpu_baseline: %d threads -> %.2f s per 8-frame UNet forward" % (n, sweep[n]))
            # torch's CPU convolutions get SLOWER with more threads on this host (measured on the 128-CPU GPU box: 0.22 s at 8
            # threads, 1.1 s at 64, minutes at 128): stop climbing once a setting is clearly worse than the best so far
            if sweep[n] > 2.0 * min(sweep.values()) or time.perf_counter() - budget_t0 > 45:
                break
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.unet_forward(dsd, x, tt, cond)
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - budget_t0 > 150:       # bounded sample: at least one full-shape forward
                break
        n_unet = len(times)
        t_unet = sum(times) / n_unet
        log("cpu_baseline: %d full-shape forwards, %.2f s each" % (n_unet, t_unet))
        t0 = time.perf_counter()
        O.generator_compute_fea(gsd, img)
        t_fea = time.perf_counter() - t0
        flow = torch.rand(b, s, s, 2) * 2 - 1
        occ = torch.rand(b, 1, s, s)
        O.generator_forward_with_flow(gsd, img, flow, occ)
        t0 = time.perf_counter()
        n_dec = 3
        for _ in range(n_dec):
            O.generator_forward_with_flow(gsd, img, flow, occ)
        t_dec = (time.perf_counter() - t0) / n_dec
    per_video = WORKLOAD["sampling_timesteps"] * t_unet + t_fea + t * t_dec
    calib = None                 # the port timed beside the reference itself on the same cores (tests/test_oracle_vs_reference.py, build container)
    try:
        with open(os.path.join(REPO_ROOT, "profiles", "port_vs_reference_cpu.json")) as f:
            calib = json.load(f)
    except (OSError, ValueError):
        pass
    return {"value": round(b / per_video, 6), "unit": "videos/s", "cores": best, "kind": "port", "host_cpus": ncpu,
            "port_over_reference": calib["port_over_reference"] if calib else None,
            "port_over_reference_note": ("one UNet forward at this shape, port / unmodified reference, best of 3 each, interleaved on the same %d torch threads of "
                                         "the %d-CPU build container (profiles/port_vs_reference_cpu.json): the port's videos/s times this ratio is what the "
                                         "reference itself would give on these cores" % (calib["threads"], calib["host_cpus"])) if calib else None,
            "thread_sweep_s_per_8_frame_unet_forward": {str(k): round(v, 3) for k, v in sweep.items()},
            "reference_figure": "SURVEY.md 6: the reference itself (its own code, torch CPU) takes 1.56 s per UNet forward of this shape on the 8 cores of "
                                "the build container (0.0053-0.0061 videos/s); /root/reference does not travel to the GPU box, so what is timed here is the PORT (kind = 'port')",
            "sample": "the PORT oracle/lfdm_oracle.py (a restatement of the reference, not the reference) on host CPU, %d threads (best of the sweep): %d UNet fwd @ (%d,259,%d,%d,%d) = %.2f s each "
                      "(min %.2f, max %.2f), compute_fea %.3f s, %d decode frames = %.3f s each; extrapolated to %d steps + %d frames"
                      % (best, n_unet, b, t, s, s, t_unet, min(times), max(times), t_fea, n_dec, t_dec, WORKLOAD["sampling_timesteps"], t)}


def gpu_eager_baseline(dev, budget_s=90.0):
    """The "context baseline" BASELINE.md section 3 planned: the same torch restatement that `cpu_baseline` times (oracle/lfdm_oracle.py =
    the reference dataflow, video_flow_diffusion_model.py:190-216: per-step 3-D convolutions, einops-style transposes, per-frame decode)
    run as stock eager PyTorch-ROCm on this GPU - MIOpen / rocBLAS / ATen kernels, no code of this repository on the device.  It is what
    "just use the GPU" gives; the headline over it is what the hand-written kernels buy.  Outside the timed region, rank 0, bounded:
    1 warm-up + up to 5 UNet forwards at the C2 shape, compute_fea, 40 decoded frames (per frame, as the reference loops) and the same 40
    frames as one batch (the friendlier form for a GPU), extrapolated linearly to one video."""
    sys.path.insert(0, os.path.join(REPO_ROOT, "oracle"))
    import lfdm_oracle as O
    import synth
    torch.manual_seed(0)
    b, t, s, hw = WORKLOAD["batch"], WORKLOAD["frames"], WORKLOAD["latent"], WORKLOAD["image"]
    t_start = time.perf_counter()
    with torch.device(dev), torch.no_grad():
        dsd = {"denoise_fn." + k: v.to(dev) for k, v in synth.unet_state().items()}
        gsd = {k: v.to(dev) for k, v in synth.generator_state().items()}
        img, cond = (v.to(dev) for v in synth.inputs(b, hw))
        x = torch.randn(b, 259, t, s, s)
        tt = torch.full((b,), 500, dtype=torch.long)
        O.unet_forward(dsd, x, tt, cond)                  # warm-up: MIOpen solver selection, allocator
        torch.cuda.synchronize()
        times = []
        for _ in range(5):
            t0 = time.perf_counter()
            O.unet_forward(dsd, x, tt, cond)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget_s:
                break
        t_unet = sum(times) / len(times)
        log("gpu_eager_baseline: %d UNet forwards, %.1f ms each" % (len(times), 1e3 * t_unet))
        O.generator_compute_fea(gsd, img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        O.generator_compute_fea(gsd, img)
        torch.cuda.synchronize()
        t_fea = time.perf_counter() - t0
        flow = torch.rand(b, s, s, 2) * 2 - 1
        occ = torch.rand(b, 1, s, s)
        O.generator_forward_with_flow(gsd, img, flow, occ)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(t):
            O.generator_forward_with_flow(gsd, img, flow, occ)
        torch.cuda.synchronize()
        t_dec_loop = time.perf_counter() - t0
        imgs, flows, occs = img.repeat(t, 1, 1, 1), flow.repeat(t, 1, 1, 1), occ.repeat(t, 1, 1, 1)
        O.generator_forward_with_flow(gsd, imgs, flows, occs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        O.generator_forward_with_flow(gsd, imgs, flows, occs)
        torch.cuda.synchronize()
        t_dec_batched = time.perf_counter() - t0
    per_video = WORKLOAD["sampling_timesteps"] * t_unet + t_fea + t_dec_loop
    per_video_batched = WORKLOAD["sampling_timesteps"] * t_unet + t_fea + t_dec_batched
    return {"value": round(b / per_video, 4), "unit": "videos/s", "kind": "port on torch-rocm (eager MIOpen / rocBLAS / ATen; no kernel of this repository)",
            "ms_per_unet_forward": round(1e3 * t_unet, 2), "ms_per_unet_forward_min": round(1e3 * min(times), 2), "unet_forwards_timed": len(times),
            "ms_compute_fea": round(1e3 * t_fea, 2), "ms_decode_40_frames_per_frame_loop": round(1e3 * t_dec_loop, 2),
            "ms_decode_40_frames_one_batch": round(1e3 * t_dec_batched, 2),
            "value_with_batched_decode": round(b / per_video_batched, 4),
            "sample": "oracle/lfdm_oracle.py on %s through stock eager PyTorch-ROCm: %d UNet fwd @ (%d,259,%d,%d,%d) after one warm-up, compute_fea, 40 decode "
                      "frames (the reference's per-frame loop; the one-batch form beside it); extrapolated to %d steps" % (
                          dev, len(times), b, t, s, s, WORKLOAD["sampling_timesteps"]),
            "note": "no hipGraph, no fusion, the reference's transposes and per-step fea convolution included: the dataflow of "
                    "video_flow_diffusion_model.py:190-216 as written.  headline / this = what the hand-written path buys over merely using the GPU"}


def kernel_census(step, dev):
    """Kernel launches of ONE call of `step` (torch.profiler, device activities): count, and the share of the device time spent in this
    library's kernels vs vendor code (ATen / MIOpen / composable_kernel / Tensile).  A library kernel = a symbol liblfdm_hip.so defines."""
    import re
    import subprocess
    from torch.profiler import ProfilerActivity, profile
    from cvpr23_lfdm_amd._native import HIP_LIB_PATH
    native = set()
    out = subprocess.run(["nm", "-C", "--defined-only", HIP_LIB_PATH], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=60).stdout
    for ln in out.splitlines():
        m = re.match(r"^[0-9a-f]+ \w (?:void )?(?:\(anonymous namespace\)::)?([A-Za-z_][A-Za-z_0-9]*)", ln)
        if m and ("_kernel" in m.group(1) or m.group(1).startswith("lfdm_")):
            native.add(m.group(1))
    step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    rows = {}
    for ev in prof.events():
        if getattr(ev, "device_type", None) is not None and str(ev.device_type).endswith("CUDA"):
            r = rows.setdefault(ev.name, [0, 0.0])
            r[0] += 1
            r[1] += float(getattr(ev, "device_time", 0.0) or getattr(ev, "cuda_time", 0.0) or 0.0)
    launches = sum(r[0] for r in rows.values())
    total_us = sum(r[1] for r in rows.values())

    def base(name):
        m = re.match(r"^(?:void )?(?:\(anonymous namespace\)::)?([A-Za-z_][A-Za-z_0-9]*)", name)
        return m.group(1) if m else name

    is_native = lambda n: base(n) in native
    nat_us = sum(r[1] for n, r in rows.items() if is_native(n))
    nat_n = sum(r[0] for n, r in rows.items() if is_native(n))
    vendor = sorted(((n, r) for n, r in rows.items() if not is_native(n)), key=lambda kv: -kv[1][1])
    return {"launches_per_step": launches, "native_launches": nat_n, "device_us_per_step": round(total_us, 1),
            "native_time_share": round(nat_us / total_us, 4) if total_us else None,
            "vendor_top": [{"kernel": n[:160], "calls": r[0], "us": round(r[1], 1)} for n, r in vendor[:int(os.environ.get("LFDM_CENSUS_TOP", "8"))]],
            "native_top": [{"kernel": n[:100], "calls": r[0], "us": round(r[1], 1)}
                           for n, r in sorted(((n, r) for n, r in rows.items() if is_native(n)), key=lambda kv: -kv[1][1])[:int(os.environ.get("LFDM_CENSUS_TOP", "8"))]],
            "vendor_families": sorted({f for n, _ in vendor for f in ("miopen", "MIOpen", "ck::", "_ZN2ck", "grid_sampler_2d", "Cijk_")
                                       if f in n}),
            "how": "torch.profiler device events of one step; native = kernel symbols defined in liblfdm_hip.so"}


def lfae_train_bench(dev, rank, world, steps, warmup, batch):
    """one LFAE stage-1 training step (LFAE/train.py:96-104 on config mug128: region predictor x3, background
    predictor, generator, VGG-19 pyramid perceptual loss, equivariance losses, backward, Adam) on `batch` synthetic 128x128 frame pairs per
    GPU; gradients all-reduced over RCCL when world > 1."""
    import yaml
    from cvpr23_lfdm_amd import lfae_train, params as P
    with open(os.path.join(REPO_ROOT, "configs", "lfae_128.yaml")) as f:
        cfg = yaml.safe_load(f)
    torch.manual_seed(1234 + rank)
    gen, reg, bgp = lfae_train.build_from_config(cfg)
    vgg = lfae_train.Vgg19()
    vgg.load_state_dict(P.synthetic_vgg19_state())
    trainer = lfae_train.LFAETrainer(gen, reg, bgp, cfg["model_params"], cfg["train_params"], vgg=vgg).to(dev)
    trainer.enable_data_parallel()
    hw = 128
    g = torch.Generator().manual_seed(77 + rank)
    base = torch.rand(batch, 3, hw // 8, hw // 8, generator=g)
    src = torch.nn.functional.interpolate(base, size=(hw, hw), mode="bilinear", align_corners=False)
    drv = (0.8 * torch.roll(src, shifts=(hw // 16, -(hw // 16)), dims=(2, 3)) + 0.2 * torch.rand(batch, 3, hw, hw, generator=g)).clamp(0, 1)
    x = {"source": src.to(dev), "driving": drv.to(dev)}
    last = {}

    def step():
        last["loss"] = trainer.step(x)[0]["total"]

    per_rank = []
    if trainer._dp is not None:
        trainer._dp.profile = True
    elapsed = timed_region(step, steps, warmup, world, torch.cuda.synchronize, per_rank)
    comm = None
    if trainer._dp is not None:
        comm = dp_diagnostics(trainer._dp, steps, world)
        trainer._dp.profile = False
    out = {"allreduce": comm,
           "value": round(steps * batch * world / elapsed, 2), "unit": "LFAE stage-1 training frame pairs/s (128x128, config mug128)",
           "ms_per_step": round(1e3 * elapsed / steps, 1), "ms_per_step_per_rank": [round(1e3 * v / steps, 1) for v in per_rank],
           "batch_per_gpu": batch, "global_batch": batch * world, "steps": steps, "warmup": warmup,
           "loss_last": round(float(last["loss"]), 4), "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
    if rank == 0 and world == 1:
        try:
            out["kernels"] = kernel_census(step, dev)
        except Exception as e:
            out["kernels"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--train-steps", type=int, default=10, help="timed DM training steps for the extra `train` object (0 = skip)")
    ap.add_argument("--train-batch", type=int, default=8, help="training videos per GPU per step")
    ap.add_argument("--train-timeout", type=int, default=240, help="seconds before the training measurement is abandoned")
    ap.add_argument("--no-gpu-eager-baseline", action="store_true", help="skip the stock eager PyTorch-ROCm context baseline")
    ap.add_argument("--lfae-train-steps", type=int, default=6, help="timed LFAE stage-1 training steps for the extra `lfae_train` object (0 = skip)")
    ap.add_argument("--lfae-train-batch", type=int, default=32, help="frame pairs per GPU per LFAE training step")
    ap.add_argument("--blocks", type=int, default=3, help="timed blocks of K steps: block 0 is the headline, the others are reported beside it")
    ap.add_argument("--batch", type=int, default=1,
                    help="videos per GPU per step; 1 = BASELINE.json configs[1] (latency mode), >1 = throughput mode")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:      # plain `python bench.py --gpus N`: become the launcher
        sys.exit(self_launch(args.gpus))
    rank, world, local = dist_setup(args.gpus)
    if os.environ.get("LFDM_BENCH_DRYRUN") == "1":      # test hook (tests/test_host_and_abi.py): launcher + rendezvous + timing protocol only
        el = timed_region(lambda: time.sleep(0.005 * (rank + 1)), args.steps, args.warmup, world, lambda: None)
        if rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": world, "steps": args.steps, "elapsed": el}), flush=True)
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    if args.batch != 1:
        WORKLOAD["batch"] = args.batch
        WORKLOAD["name"] = WORKLOAD["name"].replace("batch=1 per GPU (BASELINE.json configs[1])",
                                                    "batch=%d per GPU (throughput mode, NOT configs[1])" % args.batch)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback path exists)")
    ndev = torch.cuda.device_count()
    local = local % ndev
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    rccl_ranks = None
    if world > 1:       # who sits where: with enough devices every rank must own its own GPU (RCCL refuses duplicates anyway)
        import torch.distributed as dist
        mine = torch.tensor([local, ndev], dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        devices = [int(v[0]) for v in every]
        rccl_ranks = {"backend": dist.get_backend(), "devices": devices, "visible_gpus": ndev,
                      "one_rank_per_gpu": len(set(devices)) == world}
        assert ndev < world or len(set(devices)) == world, "ranks share a GPU although %d are visible: %s" % (ndev, devices)

    import contextlib
    import synth
    torch.manual_seed(1234)
    with contextlib.redirect_stdout(sys.stderr):      # the reference-compatible ctor prints; stdout = ONE json line
        model, _, _ = synth.build_flow_diffusion(dev, img_size=WORKLOAD["latent"], num_frames=WORKLOAD["frames"],
                                                 sampling_timesteps=WORKLOAD["sampling_timesteps"],
                                                 timesteps=WORKLOAD["timesteps"])
    img, cond = synth.inputs(WORKLOAD["batch"], WORKLOAD["image"], seed=7 + rank)
    img, cond = img.to(dev), cond.to(dev)
    torch.manual_seed(1237 + rank)          # sampling noise seed
    model.set_sample_input(sample_img=img, sample_text=cond)

    def run_step():
        model.sample_one_video(cond_scale=1.0)

    log("model built; timing %d + %d sampling steps" % (args.warmup, args.steps))
    from cvpr23_lfdm_amd import ops as _ops
    from cvpr23_lfdm_amd._build import source_fingerprint
    box = {"state_before": gpu_state(local)}
    try:
        box["calib_before"] = _ops.calib_mfma(dev)
    except Exception as e:
        box["calib_before"] = {"error": repr(e)[:200]}
    per_rank = []
    elapsed = timed_region(run_step, args.steps, args.warmup, world, torch.cuda.synchronize, per_rank)
    log("headline: %.2f ms per video" % (1e3 * elapsed / args.steps))
    box["state_after"] = gpu_state(local)
    # Reconciliation aid (round-3 verdict: the driver's box ran the same commit 13 % slower than the evidence run with nothing on the
    # line to tell why).  The headline above is the contract's ONE timed block; two more blocks of the same K follow with the shader
    # clock / power sampled while they run, and the fp32-MFMA calibration kernel brackets the whole thing.
    blocks_ms = [1e3 * elapsed / args.steps]
    if args.blocks > 1:
        with StateSampler(local) as smp:
            for _ in range(args.blocks - 1):
                el = timed_region(run_step, args.steps, 0, world, torch.cuda.synchronize)
                blocks_ms.append(1e3 * el / args.steps)
        box["during_extra_blocks"] = smp.summary()
    try:
        box["calib_after"] = _ops.calib_mfma(dev)
    except Exception as e:
        box["calib_after"] = {"error": repr(e)[:200]}
    box["calib_note"] = ("lfdm_calib_mfma_f32: 256 workgroups x 4 waves x 4 chains of v_mfma_f32_32x32x2_f32 on pseudo-random operands; "
                         "tflops vs the 157.3 peak and mhz (shader cycles / 100 MHz real-time ticks inside the kernel) say what THIS box "
                         "delivers; compare across runs before comparing videos/s")
    srt = sorted(blocks_ms)
    blocks = {"ms_per_step": [round(v, 2) for v in blocks_ms], "median": round(srt[len(srt) // 2], 2), "min": round(srt[0], 2),
              "max": round(srt[-1], 2), "note": "block 0 is the headline (`value`, `ms_per_step`); every block = %d steps between barrier + sync" % args.steps}
    log("blocks: %s" % blocks["ms_per_step"])
    videos = args.steps * WORKLOAD["batch"] * world
    value = videos / elapsed
    out = model.sample_out_vid
    assert out.shape == (WORKLOAD["batch"], 3, WORKLOAD["frames"], WORKLOAD["image"], WORKLOAD["image"])
    assert bool(torch.isfinite(out).all()) and float(out.min()) >= 0.0 and float(out.max()) <= 1.0, "bad sample"

    line = {}
    if rank == 0:
        line = {
            "metric": "40-frame 128x128 videos/sec (DDIM-100)",
            "value": round(value, 4), "unit": "videos/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 2),
            "ms_per_step_per_rank": [round(1e3 * v / args.steps, 2) for v in per_rank],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (random-init weights, rand image, randn cond; seeds 1234/7/1237)",
            "config": {"workload": WORKLOAD["name"], "global_batch": WORKLOAD["batch"] * world,
                       "frames": WORKLOAD["frames"], "latent": WORKLOAD["latent"], "image": WORKLOAD["image"],
                       "sampler": "DDIM-100 eta=1, cond_scale=1", "parallelism": "replicas x%d (videos sharded, no collective)" % world},
            "gflop_per_video_reference_dataflow": GFLOP_PER_VIDEO_REFERENCE,
            "whole_job_tflops_reference_dataflow": round(value * GFLOP_PER_VIDEO_REFERENCE / 1e3, 2),
            "blocks": blocks, "box": box, "build": source_fingerprint(), "rccl_ranks": rccl_ranks,
        }
        def guarded(fn, *a):                 # the secondary measurements must never cost the headline line
            try:
                return fn(*a)
            except Exception as e:
                return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

        if not args.no_roofline:
            line["roofline"] = guarded(conv_roofline, model, 1e3 * elapsed / args.steps / WORKLOAD["sampling_timesteps"])
            log("roofline done")
            line["warp"] = guarded(warp_bench, model, img)
            log("warp done")
        if world == 1 and not args.no_gpu_eager_baseline:
            line["gpu_eager_baseline"] = guarded(gpu_eager_baseline, dev)
            if isinstance(line["gpu_eager_baseline"], dict) and line["gpu_eager_baseline"].get("value"):
                line["gpu_eager_baseline"]["headline_over_this"] = round(value / line["gpu_eager_baseline"]["value"], 2)
            torch.cuda.empty_cache()
            log("gpu eager baseline done")
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = guarded(cpu_baseline)
            log("cpu baseline done")
    train = None
    if args.train_steps > 0:                 # every rank takes part (gradient all-reduce); after the headline measurement
        del model
        torch.cuda.empty_cache()
        import threading

        def on_timeout():                    # a hung collective must not cost the headline line (watchdog thread:
            if rank == 0:                    # works while the main thread is blocked inside a C call)
                line["train"] = {"error": "timeout after %d s" % args.train_timeout}
                print(json.dumps(line), flush=True)
            os._exit(0)

        watchdog = threading.Timer(args.train_timeout, on_timeout)
        watchdog.daemon = True
        watchdog.start()
        try:
            train = train_bench(dev, rank, world, args.train_steps, 2, args.train_batch)
        except Exception as e:               # never lose the headline line to the secondary measurement
            train = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        watchdog.cancel()
        log("train done")
    lfae = None
    if args.lfae_train_steps > 0:            # SURVEY.md 8 row f4 under the same clock; every rank takes part
        torch.cuda.empty_cache()
        import threading

        def on_lfae_timeout():
            if rank == 0:
                if train is not None:
                    line["train"] = train
                line["lfae_train"] = {"error": "timeout after %d s" % args.train_timeout}
                print(json.dumps(line), flush=True)
            os._exit(0)

        watchdog = threading.Timer(args.train_timeout, on_lfae_timeout)
        watchdog.daemon = True
        watchdog.start()
        try:
            lfae = lfae_train_bench(dev, rank, world, args.lfae_train_steps, 2, args.lfae_train_batch)
        except Exception as e:
            lfae = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        watchdog.cancel()
        log("lfae_train done")
    if rank == 0:
        if train is not None:
            line["train"] = train
        if lfae is not None:
            line["lfae_train"] = lfae
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
