"""CPU-side checks: the C-ABI library loads and exports every symbol include/lfdm_hip.h declares
(no compute call without a GPU), host-side sampler tables equal the oracle's, state-dict layouts
are checkpoint-compatible, the product path refuses to run without the GPU library, and the
multi-process timing protocol of bench.py works over gloo (world_size 2)."""
import os
import re
import subprocess
import sys

import pytest
import torch

import lfdm_oracle as O
import synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    from cvpr23_lfdm_amd import _build, _native
    path = _build.build_hip()
    header = open(os.path.join(REPO, "include", "lfdm_hip.h")).read()
    declared = set(re.findall(r"\b(lfdm_[a-z0-9_]+)\s*\(", header))
    declared -= {"lfdm_stream_t"}
    assert len(declared) >= 25
    assert declared == set(_native.EXPORTED_SYMBOLS), declared ^ set(_native.EXPORTED_SYMBOLS)
    lib = _native.NativeLibrary(path, "hip")          # getattr on every symbol; raises if one is missing
    assert lib.lfdm_abi_version() == 12         # 12: defer_reduce / gn_in_* reserved, lfdm_groupnorm_splitk_* removed; 11: BatchNorm segments; 10: LFAE stage-1 training glue; 9: lfdm_wgrad_params.dw_layout / .dbias, lfdm_multi_linear_*; 8: lfdm_conv_params.gn_in_*; 7: lfdm_calib_mfma_f32; 2: lfdm_conv_params.deconv4 / .groups, heads ld; 3: .pool2; 4: *_lowres_cl_f32; 5: .weight_wino4; 6: .defer_reduce
    nm = subprocess.run(["nm", "-D", "--defined-only", path], stdout=subprocess.PIPE, text=True).stdout
    for sym in declared:
        assert re.search(r"\bT %s\b" % sym, nm), sym


def test_product_path_refuses_cpu_tensors():
    from cvpr23_lfdm_amd import _native, ops
    _native._set_library_for_tests(None)
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by the gpu tests")
    x = torch.zeros(4, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layernorm_cl(x, torch.ones(64))


def test_sampler_tables_match_oracle():
    from cvpr23_lfdm_amd import GaussianDiffusion
    d = GaussianDiffusion(torch.nn.Identity(), image_size=32, num_frames=40, timesteps=1000,
                          sampling_timesteps=100, loss_type="l2", use_dynamic_thres=True)
    sched = O.make_schedule(1000)
    for k in O.SCHEDULE_KEYS:
        assert torch.equal(getattr(d, k), sched[k]), k
    assert d.ddim_times() == O.ddim_time_pairs(1000, 100)
    times, coef, draws = d._step_tables(True)
    assert times[0] == 990 and times[-1] == 9 and len(times) == 100
    assert draws == [True] * 99 + [False]
    a, an = sched["alphas_cumprod_prev"][990], sched["alphas_cumprod_prev"][980]
    sigma = ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
    assert torch.equal(coef[0, 5], sigma) and torch.equal(coef[0, 2], an.sqrt())
    assert float(coef[-1, 5]) == 0.0
    times, coef, draws = d._step_tables(False)
    assert times == list(reversed(range(1000))) and all(draws) and float(coef[-1, 5]) == 0.0
    assert torch.equal(coef[0, 2], sched["posterior_mean_coef1"][999])


def test_sampler_tables_are_kept_per_schedule_and_follow_the_buffers():
    """`_step_tables_on` (the tables of `_step_tables` on the sampling device, kept between videos so that no per-video host read-back synchronises with
    the GPU): the same objects for the same schedule, rebuilt when a registered buffer is written (load_state_dict, an edited schedule), when the
    schedule changes, and never cached for an instance-level replacement of `_step_tables` (the teacher-forced tests)."""
    from cvpr23_lfdm_amd import GaussianDiffusion
    d = GaussianDiffusion(torch.nn.Identity(), image_size=8, num_frames=4, timesteps=50, sampling_timesteps=10, loss_type="l2")
    t1, c1, tt1, dr1 = d._step_tables_on(True, torch.device("cpu"))
    t2, c2, tt2, dr2 = d._step_tables_on(True, torch.device("cpu"))
    assert c1 is c2 and tt1 is tt2 and t1 == t2 and tt1.tolist() == t1
    ref_t, ref_c, _ = d._step_tables(True)
    assert torch.equal(c1, ref_c) and t1 == ref_t
    d.sqrt_recip_alphas_cumprod.mul_(2.0)                    # an in-place write bumps the buffer's version counter
    t3, c3, _, _ = d._step_tables_on(True, torch.device("cpu"))
    assert c3 is not c1 and torch.equal(c3, d._step_tables(True)[1]) and not torch.equal(c3, c1)
    t4, c4, _, _ = d._step_tables_on(False, torch.device("cpu"))
    assert len(t4) == 50 and c4 is not c3
    d._step_tables = lambda ddim: ([7], torch.zeros(1, 6), [True])
    t5, c5, tt5, _ = d._step_tables_on(True, torch.device("cpu"))
    assert t5 == [7] and tt5.tolist() == [7] and "_tables_cache" in d.__dict__ and d.__dict__["_tables_cache"][1] != [7]


def test_state_dict_layout_and_roundtrip():
    from cvpr23_lfdm_amd import FlowDiffusion
    m = FlowDiffusion(img_size=8, num_frames=4, sampling_timesteps=5, is_train=True, config_pth=synth.CONFIG)
    dsd = m.diffusion.state_dict()
    assert len(dsd) == 324
    assert sum(p.numel() for p in m.unet.parameters()) == 42731203
    assert len(m.generator.state_dict()) == 196
    assert len(m.region_predictor.state_dict()) == 73 and len(m.bg_predictor.state_dict()) == 37
    assert "denoise_fn.downs.2.3.fn.fn.fn.rotary_emb.freqs" in dsd and "betas" in dsd
    m.diffusion.load_state_dict(dsd)
    assert isinstance(m.optimizer_diff, torch.optim.Optimizer) and m.optimizer_diff.param_groups[0]["betas"] == (0.9, 0.99)
    torch.optim.lr_scheduler.MultiStepLR(m.optimizer_diff, milestones=[3], gamma=0.1)      # train script :210-211


def test_bench_timing_protocol_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, time, json\n"
        "sys.path.insert(0, %r)\n"
        "import torch, torch.distributed as dist\n"
        "import bench\n"
        "rank, world, local = bench.dist_setup(2)\n"
        "def step():\n"
        "    time.sleep(0.01 * (rank + 1))\n"
        "el = bench.timed_region(step, 3, 1, world, lambda: None)\n"
        "if rank == 0: print(json.dumps({'elapsed': el, 'world': world}))\n"
        "dist.destroy_process_group()\n" % REPO)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29571", str(script)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["world"] == 2 and out["elapsed"] >= 0.055     # max over ranks: rank 1 sleeps 3 x 20 ms


@pytest.mark.parametrize("n", [2, 8])
def test_bench_self_launch(n):
    """`python bench.py --gpus N` with no launcher in the environment re-execs itself under torch.distributed.run: N ranks
    rendezvous (gloo here), run the barrier / max-over-ranks protocol and rank 0 prints ONE line with n_gpus = N.  (N = 8: the node
    the driver's scaling run uses - launcher, rendezvous and protocol exercised at that world size before any 8-GPU box is.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LFDM_BENCH_DRYRUN="1", LFDM_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout       # stdout = the JSON line and nothing else (no backend chatter)
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["elapsed"] >= 0.009 * n      # the slowest rank sleeps 2 x 5 n ms


@pytest.mark.gpu
def test_bench_gpus2_over_rccl():
    """`python bench.py --gpus 2` on a box with two GPUs: the ranks must sit on DIFFERENT devices and rendezvous over RCCL (backend
    nccl) - bench.py asserts it and reports `rccl_ranks`.  Skips on the one-GPU box (there `profiles/*_bench_n2_one_gpu.json` is the
    gloo flavour with both ranks on cuda:0)."""
    import json
    import torch
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n < 2:
        pytest.skip("needs two GPUs for RCCL: %d visible" % n)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LFDM_DIST_BACKEND")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--blocks", "1",
                        "--no-cpu-baseline", "--no-roofline", "--train-steps", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["rccl_ranks"]["backend"] == "nccl" and len(set(out["rccl_ranks"]["devices"])) == 2, out.get("rccl_ranks")
    assert out["train"]["allreduce"]["buckets"] >= 1 and "error" not in out["train"]


def test_asm_load_pipelines_are_safe():
    """Every kernel file that issues loads by inline asm (lfdm_gload_f4) is compiled to gfx950 assembly and walked by
    tools/check_asm_pipeline.py: no instruction may touch the destination of a load that is still in flight (hipcc spills / re-uses
    such registers under pressure - a wild-pointer fault on the GPU), nothing pending at a branch or at the end."""
    import glob
    files = [f for f in glob.glob(os.path.join(REPO, "cvpr23_lfdm_amd", "csrc", "*.hip")) if "lfdm_gload_f4" in open(f).read()]
    assert files
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "check_asm_pipeline.py")] + files, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "0 problem(s)" in r.stdout


def test_pmc_video_report_on_a_synthetic_rocpd_database(tmp_path):
    """tools/pmc_video_report.py (whole-step counters behind bench.py's roofline.step_traffic): the last sampler step is cut out between
    the last two sampler_update_kernel dispatches, split-K reduce launches go to the family of the convolution before them, FETCH_SIZE is
    doubled and both sizes are KB."""
    import json
    import sqlite3
    import subprocess
    import sys
    seq = ["void (anonymous namespace)::step_cond_kernel(int)", "void conv_wino_kernel<false, 1, false>(lfdm_conv_params)",
           "void conv_splitk_reduce_kernel(lfdm_conv_params)", "void conv_pw_kernel<1, 4, true>(lfdm_conv_params, int, int)",
           "void gn_apply_kernel<512>(float const*)", "void sampler_update_kernel(p)"]
    dbs = []
    for tag, ctrs in (("f", {"FETCH_SIZE": 100.0}), ("w", {"WRITE_SIZE": 10.0}),
                      ("s", {"SQ_VALU_MFMA_BUSY_CYCLES": 512.0, "SQ_BUSY_CU_CYCLES": 1.0, "SQ_WAVE_CYCLES": 4.0, "SQ_WAIT_ANY": 1.0}),
                      ("t", {"TCC_HIT_sum": 3.0, "TCC_REQ_sum": 4.0, "GRBM_GUI_ACTIVE": 1.0})):
        path = str(tmp_path / ("p_%s.db" % tag))
        db = sqlite3.connect(path)
        db.execute("create table pmc_events(dispatch_id int, name text, counter_name text, counter_value real, duration int)")
        did = 0
        for _ in range(3):
            for k in seq:
                did += 1
                for cn, v in ctrs.items():
                    for _inst in range(2):                      # two counter instances per dispatch
                        db.execute("insert into pmc_events values(?,?,?,?,?)", (did, k, cn, v, 5000))
        for k in ("void warp_cl_kernel<4>(p)", "void warp_planar_pixel_kernel(p)"):
            did += 1
            for cn, v in ctrs.items():
                db.execute("insert into pmc_events values(?,?,?,?,?)", (did, k, cn, v, 30000))
        db.commit()
        db.close()
        dbs.append(path)
    out = str(tmp_path / "step.json")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_video_report.py")] + dbs + [out], check=True, stdout=subprocess.DEVNULL)
    r = json.load(open(out))
    assert r["launches_per_step"] == len(seq)
    assert r["step_fetch_bytes"] == len(seq) * 2 * 100 * 2 * 1024          # 2 instances x 100 KB, doubled
    assert r["step_write_bytes"] == len(seq) * 2 * 10 * 1024
    assert r["families"]["winograd"]["launches"] == 2                       # the convolution and its reduce pass
    assert r["families"]["direct"]["launches"] == 1 and r["families"]["norm"]["launches"] == 1
    assert r["warp_launches"] == 2 and [w["kernel"][:7] for w in r["warp_launch_list"]] == ["warp_cl", "warp_pl"]
    assert abs(r["step_mfma_util"] - (len(seq) * 2 * 512.0) / (1024.0 * len(seq) * 1.0)) < 1e-3   # GUI_ACTIVE: mean over instances
