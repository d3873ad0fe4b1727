"""Python face of the C ABI (include/lfdm_hip.h): one thin function per entry point.

Tensors are torch fp32 tensors used only as device memory (data_ptr + current HIP stream);
all arithmetic happens in the HIP kernels of liblfdm_hip.so.  "CL" activations are 2-D tensors
(rows, C): rows = N*H*W pixels in (n, y, x) order, UNet frames n = b*T + t.
"""
import ctypes as C

import torch

from . import _native
from ._native import ConvParams, WarpParams

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SILU, ACT_GELU = 0, 1, 2, 3, 4


def _lib():
    return _native.library()


def _stream(lib):
    if lib.kind == "hip":
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return None


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(lib, *tensors):
    for t in tensors:
        if t is None:
            continue
        if t.dtype not in (torch.float32, torch.int32):
            raise TypeError("lfdm ops take float32/int32 tensors, got %s" % t.dtype)
        if lib.kind == "hip" and not t.is_cuda:
            raise RuntimeError("lfdm ops need tensors on the GPU (no CPU fallback exists)")
        if lib.kind == "emu" and t.is_cuda:
            raise RuntimeError("emulation library needs CPU tensors")


def empty(shape, like=None, device=None, dtype=torch.float32):
    dev = device if device is not None else like.device
    return torch.empty(shape, dtype=dtype, device=dev)


# ---------------------------------------------------------------------------------------------
# weight packing (one-time, load-time plumbing)
# ---------------------------------------------------------------------------------------------

def round_up(v, m):
    return (v + m - 1) // m * m


def _pack_k_major(wk, cout):
    """(K, cout) rows in k order -> [ceil(K/32)][coutp][32] (k contiguous per output channel)."""
    k = wk.shape[0]
    kp, coutp = round_up(k, 32), round_up(cout, 32)
    full = torch.zeros(kp, coutp, dtype=torch.float32, device=wk.device)
    full[:k, :cout] = wk
    return full.view(kp // 32, 32, coutp).permute(0, 2, 1).contiguous()


def pack_conv_weight(w):
    """(Cout, Cin, kh, kw) [or (Cout, Cin, 1, kh, kw) / (Cout, Cin)] -> lfdm_conv2d_cl_f32 layout:
    [ceil(K/32)][coutp][32], K = kh*kw*Cin flattened tap-major then channel."""
    if w.dim() == 5:
        w = w[:, :, 0]
    if w.dim() == 2:
        w = w[:, :, None, None]
    cout, cin, kh, kw = w.shape
    return _pack_k_major(w.permute(2, 3, 1, 0).reshape(kh * kw * cin, cout), cout)


def pack_deconv_weight(w):
    """ConvTranspose3d weight (Cin, Cout, 1, 4, 4), stride 2, padding 1 -> four parity packs
    [(py, px, packed)]: output pixel (2q+py, 2q'+px) = 2x2 conv with pad (1-py, 1-px);
    tap ky' uses kernel row ky = 3 - 2ky' (py = 0) or 2 - 2ky' (py = 1)."""
    if w.dim() == 5:
        w = w[:, :, 0]
    cin, cout, kh, kw = w.shape
    assert kh == 4 and kw == 4
    packs = []
    for py in (0, 1):
        for px in (0, 1):
            kys = [3, 1] if py == 0 else [2, 0]
            kxs = [3, 1] if px == 0 else [2, 0]
            taps = [w[:, :, ky, kx] for ky in kys for kx in kxs]          # each (Cin, Cout)
            packs.append((py, px, _pack_k_major(torch.cat(taps, dim=0), cout)))
    return packs


def pack_deconv4_weight(w):
    """The four parity packs of pack_deconv_weight stacked in parity order q = 2*py + px: (4, chunks, coutp, 32), the
    `deconv4=` operand of conv_params (the whole ConvTranspose k4 s2 p1 as ONE launch)."""
    return torch.stack([pk for _, _, pk in pack_deconv_weight(w)]).contiguous()


def pack_conv_weight_dev(w, mode=0):
    """The packs of pack_conv_weight (mode 0), of the data-gradient filter w.transpose(0, 1).flip(-2, -1) (mode 1) and of
    pack_deconv4_weight (mode 2) built by ONE library launch (lfdm_pack_conv_weight_f32) - the training path re-packs every
    step.  w: (O, I, kh, kw), any strides on the two channel axes (a slice is not copied), the taps contiguous."""
    lib = _lib()
    _chk(lib, w)
    if w.dim() == 5:
        w = w[:, :, 0]
    if w.dim() == 2:
        w = w[:, :, None, None]
    n_o, n_i, kh, kw = w.shape
    taps = kh * kw
    if taps > 1 and (w.stride(3) != 1 or w.stride(2) != kw):
        w = w.contiguous()
    k = taps * n_i if mode == 0 else taps * n_o if mode == 1 else 4 * n_o
    n = n_o if mode == 0 else n_i
    shape = (round_up(k, 32) // 32, round_up(n, 32), 32)
    out = torch.empty(((4,) + shape) if mode == 2 else shape, dtype=torch.float32, device=w.device)
    lib.check(lib.lfdm_pack_conv_weight_f32(_p(w), n_o, n_i, taps, w.stride(0), w.stride(1), mode, _p(out), _stream(lib)),
              "lfdm_pack_conv_weight_f32")
    return out


def svd2x2_sym(a, b, c):
    """lfdm_svd2x2_sym_f32: (U (n,2,2), S (n,2)) of [[a, b], [b, c]] with LAPACK's sign convention."""
    lib = _lib()
    abc = torch.stack((a, b, c), dim=-1).float().contiguous()
    _chk(lib, abc)
    n = abc.shape[0]
    u = torch.empty(n, 2, 2, dtype=torch.float32, device=abc.device)
    s = torch.empty(n, 2, dtype=torch.float32, device=abc.device)
    lib.check(lib.lfdm_svd2x2_sym_f32(_p(abc), n, _p(u), _p(s), _stream(lib)), "lfdm_svd2x2_sym_f32")
    return u, s


def lfae_region_stats(logits, n, k, h, w, temperature):
    """lfdm_lfae_region_stats_f32 on the `regions` head's channels-last rows -> the RegionPredictor's output dict."""
    lib = _lib()
    _chk(lib, logits)
    dev = logits.device
    e = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
    heat, shift, covar, affine, u, d = e(n, k, h, w), e(n, k, 2), e(n, k, 2, 2), e(n, k, 2, 2), e(n * k, 2, 2), e(n * k, 2, 2)
    lib.check(lib.lfdm_lfae_region_stats_f32(_p(logits), logits.stride(0), n, k, h, w, float(temperature), _p(heat), _p(shift),
                                             _p(covar), _p(affine), _p(u), _p(d), _stream(lib)), "lfdm_lfae_region_stats_f32")
    return {"shift": shift, "heatmap": heat, "covar": covar, "affine": affine, "u": u, "d": d}


def lfae_motion_inputs(src_img, driving, source, bg, frames, *, region_var, revert_axis_swap, use_covar, pad_to=32):
    """lfdm_lfae_motion_inputs_f32: -> (rows (N*h*w, ld) = the pixelwise-flow hourglass' channels-last input, sparse (N, K+1, h, w, 2)).
    src_img (B, 3, h, w); driving: dict of (N = B*frames, K, ...) tensors, source: dict of (B, K, ...) tensors."""
    lib = _lib()
    b, c, h, w = src_img.shape
    n = b * frames
    k = driving["shift"].shape[1]
    f = lambda t: None if t is None else t.detach().float().contiguous()
    img = f(src_img)
    dsh, ssh = f(driving["shift"]), f(source["shift"])
    dcv, scv = (f(driving["covar"]), f(source["covar"])) if use_covar else (None, None)
    daf, saf = (f(driving["affine"]), f(source["affine"])) if "affine" in driving else (None, None)
    bgm = f(bg)
    _chk(lib, img, dsh, ssh, dcv, scv, daf, saf, bgm)
    assert c == 3 and dsh.shape[0] == n and ssh.shape[0] == b
    ld = round_up(4 * (k + 1), pad_to)
    rows = torch.empty(n * h * w, ld, dtype=torch.float32, device=img.device)
    sparse = torch.empty(n, k + 1, h, w, 2, dtype=torch.float32, device=img.device)
    lib.check(lib.lfdm_lfae_motion_inputs_f32(_p(img), _p(dsh), _p(dcv), _p(daf), _p(ssh), _p(scv), _p(saf), _p(bgm),
                                              float(region_var), int(bool(revert_axis_swap)), b, frames, k, h, w, _p(rows), ld,
                                              _p(sparse), _stream(lib)), "lfdm_lfae_motion_inputs_f32")
    return rows, sparse


def lfae_motion_combine(heads, sparse, has_occ):
    """lfdm_lfae_motion_combine_f32: heads = channels-last rows (N*h*w, >= K+1 [+1]) of the mask (+ occlusion) convolutions ->
    (optical_flow (N, h, w, 2), occlusion_map (N, 1, h, w) or None)."""
    lib = _lib()
    _chk(lib, heads, sparse)
    n, k1, h, w, _ = sparse.shape
    flow = torch.empty(n, h, w, 2, dtype=torch.float32, device=sparse.device)
    occ = torch.empty(n, 1, h, w, dtype=torch.float32, device=sparse.device) if has_occ else None
    lib.check(lib.lfdm_lfae_motion_combine_f32(_p(heads), heads.stride(0), _p(sparse), n, k1 - 1, h * w, _p(flow), _p(occ),
                                               _stream(lib)), "lfdm_lfae_motion_combine_f32")
    return flow, occ


def pack_wino_weight_grouped(ws):
    """[(Cout_g, Cin_g, 3, 3)] * G -> (G, 16, Cin_g/16, Cout_g, 16): the filters of a grouped 3x3 convolution
    (lfdm_conv_params.groups), each group's Winograd pack back to back."""
    packs = [pack_wino_weight(w.contiguous(), coutp=w.shape[0]) for w in ws]
    return torch.stack(packs).contiguous()


def pack_ln_conv_weight(w, gamma):
    """1x1 conv / linear weight (Cout, Cin[,1,1]) preceded by a channel LayerNorm with scale gamma (Cin,):
    returns (packed W' = W*gamma, ln_wsum (coutp,) = sum_c W'[o][c]) for lfdm_conv_params.ln_wsum."""
    w2 = w.reshape(w.shape[0], -1).float() * gamma.reshape(1, -1).float()
    packed = pack_conv_weight(w2)
    wsum = torch.zeros(packed.shape[1], dtype=torch.float32, device=w.device)
    wsum[: w2.shape[0]] = w2.double().sum(dim=1).float()
    return packed, wsum


def pack_wino_weight(w, coutp=None, dgrad=False, out=None):
    """(Cout, Cin, 3, 3) -> Winograd F(2x2,3x3) filters U = G g G^T as [16][K/16][coutp][16] (position, reduction-channel
    chunk, output channel, channel in chunk) - the operand order conv_wino.hip loads (lfdm_conv_params.weight_wino).
    dgrad=True: the filters of the data-gradient convolution dY -> dX (channel roles exchanged, taps flipped);
    `w` may be a slice w[:, lo:hi] of the input-channel axis (no copy).  out: an existing pack of this geometry to refill IN PLACE
    (its address may be held by a captured graph).  lfdm_pack_wino_weight_f32."""
    lib = _lib()
    if w.dim() == 5:
        w = w[:, :, 0]
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3 and w.dtype == torch.float32
    assert w.stride(3) == 1 and w.stride(2) == 3 and w.stride(1) == 9, "input-channel slices of a contiguous filter only"
    _chk(lib, w)
    k, n = (cout, cin) if dgrad else (cin, cout)
    assert k % 16 == 0
    coutp = coutp or (out.shape[2] if out is not None else (n + 31) // 32 * 32)
    if out is None:
        out = torch.empty(16, k // 16, coutp, 16, dtype=torch.float32, device=w.device)
    assert tuple(out.shape) == (16, k // 16, coutp, 16) and out.is_contiguous() and out.device == w.device
    lib.check(lib.lfdm_pack_wino_weight_f32(_p(w), w.stride(0), cout, cin, coutp, int(dgrad), _p(out), _stream(lib)),
              "lfdm_pack_wino_weight_f32")
    return out


_PACK_JOB_TABLES = {}


def pack_wino_weights_multi(jobs):
    """lfdm_pack_wino_weights_multi_f32: jobs = [(w view (Cout, Cin, 3, 3) or an input-channel slice of one, packed output tensor, dgrad)]
    - every filter re-packed into its EXISTING output tensor by one launch.  The device job table is cached per job list (pointers)."""
    import numpy as np
    lib = _lib()
    key, recs, block0 = [], [], 0
    for w, out, dgrad in jobs:
        cout, cin = w.shape[0], w.shape[1]
        k, n = (cout, cin) if dgrad else (cin, cout)
        coutp = out.shape[2]
        assert out.shape == (16, k // 16, coutp, 16) and out.is_contiguous() and w.stride(3) == 1 and w.stride(2) == 3 and w.stride(1) == 9
        _chk(lib, w, out)
        recs.append((w.data_ptr(), out.data_ptr(), w.stride(0), cout, cin, coutp, int(bool(dgrad)), block0))
        key.append(recs[-1][:7])
        block0 += (k // 4 * coutp + 255) // 256          # one thread per four reduction channels (conv_wino.hip pack_wino_item)
    key = (tuple(key), str(jobs[0][1].device))
    table = _PACK_JOB_TABLES.get(key)
    if table is None:
        if len(_PACK_JOB_TABLES) > 16:
            _PACK_JOB_TABLES.clear()
        dt = np.dtype([("w", "<u8"), ("out", "<u8"), ("ld_o", "<i4"), ("cout", "<i4"), ("cin", "<i4"), ("coutp", "<i4"), ("dgrad", "<i4"),
                       ("block0", "<i4")])
        arr = np.array(recs, dtype=dt)
        assert arr.dtype.itemsize == 40
        table = _PACK_JOB_TABLES[key] = torch.from_numpy(arr.view(np.uint8).copy()).to(jobs[0][1].device)
    lib.check(lib.lfdm_pack_wino_weights_multi_f32(C.c_void_p(table.data_ptr()), len(recs), block0, _stream(lib)), "lfdm_pack_wino_weights_multi_f32")


def pack_wino4_weight(w, coutp=None):
    """(Cout, Cin, 3, 3) -> Winograd F(4x4,3x3) filters U = G g G^T as [36][Cin/8][coutp][8] (lfdm_conv_params.weight_wino4, the
    batched-shape schedule of conv_wino4.hip).  lfdm_pack_wino4_weight_f32."""
    lib = _lib()
    if w.dim() == 5:
        w = w[:, :, 0]
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3 and w.dtype == torch.float32 and cin % 8 == 0
    assert w.stride(3) == 1 and w.stride(2) == 3 and w.stride(1) == 9, "input-channel slices of a contiguous filter only"
    _chk(lib, w)
    coutp = coutp or (cout + 31) // 32 * 32
    out = torch.empty(36, cin // 8, coutp, 8, dtype=torch.float32, device=w.device)
    lib.check(lib.lfdm_pack_wino4_weight_f32(_p(w), w.stride(0), cout, cin, coutp, _p(out), _stream(lib)), "lfdm_pack_wino4_weight_f32")
    return out


def pack_planar_in_weight(w):
    """(Cout, Cin, kh, kw) -> [kh*kw*Cin][Cout] (tap-major, then channel) for conv_planar_in_cl."""
    if w.dim() == 5:
        w = w[:, :, 0]
    cout, cin, kh, kw = w.shape
    return w.permute(2, 3, 1, 0).reshape(kh * kw * cin, cout).contiguous()


# ---------------------------------------------------------------------------------------------
# ops
# ---------------------------------------------------------------------------------------------

def conv_params(src0, weight, cout, kh, kw, n_img, hi, wi, *, src1=None, bias=None, pad=None, stride=1,
                upsample=False, reflect=False, residual=None, act=ACT_NONE, out=None, hq=None, wq=None,
                ho=None, wo=None, out_scale=1, out_off=(0, 0), ksplit=0, ln_wsum=None, ln_eps=1e-5,
                tile_counters=None, weight_wino=None, deconv4=None, groups=1, pool2=False, weight_wino4=None, weight_pw=None, res_gn=None):
    """Fills an lfdm_conv_params struct (allocating `out` if needed); returns (params, out).
    ksplit=0 lets the library choose (conv_plan reports the choice)."""
    lib = _lib()
    _chk(lib, src0, src1, weight, bias, residual, out, ln_wsum)
    cin = src0.shape[1] + (src1.shape[1] if src1 is not None else 0)
    if groups > 1:      # grouped convolution: Winograd form only, `weight` is not read (lfdm_conv_params.groups)
        assert weight_wino is not None and src1 is None and weight is weight_wino
        coutp = cout
    elif weight is None:     # Winograd-only call (training re-packs filters every step: no direct-form pack); conv2d_cl verifies
        assert weight_wino is not None and deconv4 is None          # with lfdm_conv2d_schedule that the library agrees
        weight, coutp = weight_wino, weight_wino.shape[2]
    else:
        assert weight.shape[0] == (kh * kw * cin + 31) // 32 and weight.shape[2] == 32, (weight.shape, kh, kw, cin)
        coutp = weight.shape[1]
    pad_y, pad_x = (kh // 2, kw // 2) if pad is None else pad
    h_in = hi * 2 if upsample else hi
    w_in = wi * 2 if upsample else wi
    if hq is None:
        hq = (h_in + 2 * pad_y - kh) // stride + 1
        wq = (w_in + 2 * pad_x - kw) // stride + 1
    if ho is None:
        ho, wo = (hq // 2, wq // 2) if pool2 else (hq * out_scale, wq * out_scale)
    if out is None:
        out = torch.empty(n_img * ho * wo, cout, dtype=torch.float32, device=src0.device)
    p = ConvParams()
    p.src0, p.src1 = _p(src0), _p(src1)
    p.c0, p.c1 = src0.shape[1], (src1.shape[1] if src1 is not None else 0)
    p.ld0, p.ld1 = src0.stride(0), (src1.stride(0) if src1 is not None else 0)
    p.n_img, p.hi, p.wi, p.hq, p.wq = n_img, hi, wi, hq, wq
    p.stride, p.upsample, p.pad_mode = stride, int(upsample), int(reflect)
    p.kh, p.kw, p.pad_y, p.pad_x = kh, kw, pad_y, pad_x
    p.weight, p.cout, p.coutp, p.bias = _p(weight), cout, coutp, _p(bias)
    p.out, p.ldo, p.ho, p.wo = _p(out), out.stride(0), ho, wo
    p.out_scale, p.out_off_y, p.out_off_x = out_scale, out_off[0], out_off[1]
    p.residual, p.ldr = _p(residual), (residual.stride(0) if residual is not None else 0)
    p.act, p.ksplit, p.partial = act, ksplit, None
    p.gn_partial, p.gn_groups, p.gn_pixels = None, 0, 0
    p.ln_wsum, p.ln_eps = _p(ln_wsum), ln_eps
    p.tile_counters, p.tile_counters_len = None, 0
    if tile_counters is not None:       # zero-initialised int32 words: split-K slabs are reduced inside the launch
        assert tile_counters.dtype == torch.int32 and tile_counters.is_contiguous()
        _chk(lib, tile_counters)
        p.tile_counters, p.tile_counters_len = tile_counters.data_ptr(), tile_counters.numel()
    p.weight_wino = None
    p.groups = groups if groups > 1 else 0
    if weight_wino is not None:
        _chk(lib, weight_wino)
        want = (16, cin // 16, coutp, 16) if groups <= 1 else (groups, 16, cin // groups // 16, coutp // groups, 16)
        assert kh == 3 and kw == 3 and weight_wino.shape == want and weight_wino.is_contiguous(), (weight_wino.shape, want)
        p.weight_wino = weight_wino.data_ptr()
    p.deconv4 = 0
    if deconv4 is not None:     # the four parity packs of pack_deconv_weight, stacked: ONE launch (lfdm_conv_params.deconv4)
        _chk(lib, deconv4)
        assert kh == 2 and kw == 2 and out_scale == 2 and deconv4.is_contiguous() and deconv4.shape == (4,) + tuple(weight.shape) \
            and deconv4.data_ptr() == weight.data_ptr(), "weight must be deconv4[0]"
        p.deconv4 = 1
    p.weight_wino4 = None
    if weight_wino4 is not None:        # F(4x4,3x3) form for batched shapes (the library decides: lfdm_conv2d_schedule == 4)
        _chk(lib, weight_wino4)
        assert weight_wino is not None and weight_wino4.shape == (36, cin // 8, coutp, 8) and weight_wino4.is_contiguous()
        p.weight_wino4 = weight_wino4.data_ptr()
    p.weight_pw = None
    if weight_pw is not None:          # the same 1x1 filter in operand order for the pointwise schedule (pack_pw_weight)
        _chk(lib, weight_pw)
        assert weight_pw.numel() == weight.numel() and weight_pw.is_contiguous()      # (1x1, or the gather form of the 4x4 / stride-2 Downsample)
        p.weight_pw = weight_pw.data_ptr()
    keep_gn = (weight_pw,)
    p.gn_in_partial, p.defer_reduce = None, 0      # (reserved since ABI 12: must stay NULL / 0)
    # res_gn = dict(partial, nchunk, pixels, gamma, beta, groups=8, eps=1e-5): `residual` is a RAW convolution output whose GroupNorm + SiLU is
    # applied in this (pointwise) convolution's epilogue (lfdm_conv_params.res_gn_*)
    p.res_gn_partial = None
    if res_gn is not None:
        gp, gg, gb = res_gn["partial"], res_gn["gamma"], res_gn["beta"]
        _chk(lib, gp, gg, gb)
        assert residual is not None and gg.numel() == cout == gb.numel() and gp.is_contiguous()
        p.res_gn_partial, p.res_gn_nchunk, p.res_gn_groups, p.res_gn_pixels = _p(gp), int(res_gn["nchunk"]), int(res_gn.get("groups", 8)), int(res_gn["pixels"])
        p.res_gn_gamma, p.res_gn_beta, p.res_gn_eps = _p(gg), _p(gb), float(res_gn.get("eps", 1e-5))
        keep_gn = keep_gn + (gp, gg, gb)
    p.pool2 = int(bool(pool2))          # Winograd schedule only (the library refuses it elsewhere): the 2x2 average pool behind conv -> act
    p._keep = (src0, src1, weight, bias, residual, out, ln_wsum, tile_counters, weight_wino, deconv4, weight_wino4) + keep_gn   # keep the tensors alive with the struct
    return p, out


def conv_partial_floats(p):
    """Split-K scratch the library wants for these params (slabs [+ LayerNorm row statistics]), in floats."""
    return _lib().lfdm_conv2d_partial_bytes(C.byref(p)) // 4


def conv_plan_slabs(p):
    """Slabs per output tile the launch will sum (lfdm_conv2d_plan_slabs): ksplit, or more for a balanced Winograd launch."""
    return _lib().lfdm_conv2d_plan_slabs(C.byref(p))


class WinogradUnavailable(RuntimeError):
    pass


def conv_plan(p):
    """(tile_rows, ksplit) the library will use for these params (lfdm_conv2d_plan)."""
    lib = _lib()
    rows, ks = C.c_int(0), C.c_int(0)
    lib.check(lib.lfdm_conv2d_plan(C.byref(p), C.byref(rows), C.byref(ks)), "lfdm_conv2d_plan")
    return rows.value, ks.value


def conv_schedule(p):
    """lfdm_conv2d_schedule: 0 = implicit GEMM, 1 = K-split across waves, 2 = Winograd F(2x2), 3 = pointwise, 4 = Winograd F(4x4)."""
    return _lib().lfdm_conv2d_schedule(C.byref(p))


def conv_launch(p):
    lib = _lib()
    lib.check(lib.lfdm_conv2d_cl_f32(C.byref(p), _stream(lib)), "lfdm_conv2d_cl_f32")


def conv2d_cl(src0, weight, cout, kh, kw, n_img, hi, wi, *, partial=None, gn_partial=None, gn_groups=8,
              gn_pixels=0, **kw_):
    """One convolution (see conv_params for the keywords).  Split-K scratch is allocated on demand."""
    lib = _lib()
    _chk(lib, partial, gn_partial)
    p, out = conv_params(src0, weight, cout, kh, kw, n_img, hi, wi, **kw_)
    if (weight is None or kw_.get("pool2")) and lib.lfdm_conv2d_schedule(C.byref(p)) not in ((2,) if kw_.get("pool2") else (2, 4)):
        raise WinogradUnavailable("the library would not run the Winograd schedule for this geometry: pass the direct-form pack / "
                                  "run the pooling as a launch of its own")
    if gn_partial is not None:      # before the plan is asked for: the pointwise schedule has no fused statistics
        p.gn_partial, p.gn_groups, p.gn_pixels = _p(gn_partial), gn_groups, gn_pixels
    need = conv_partial_floats(p)          # (also non-zero for a ksplit = 1 plan that balances its Winograd launch: lfdm_hip.h, tile_counters)
    if need > 0:
        if partial is None or partial.numel() < need:
            partial = torch.empty(need, dtype=torch.float32, device=src0.device)
        p.partial = _p(partial)
    conv_launch(p)
    return out


def pack_smalln_weight(w, bias=None):
    """(Cout <= 4, Cin, k, k) -> ([k*k][Cin][4] filters innermost, zero padded; bias padded to 4)."""
    cout, cin, kh, kw = w.shape
    assert cout <= 4 and kh == kw
    wp = torch.zeros(kh * kw, cin, 4, dtype=torch.float32, device=w.device)
    wp[:, :, :cout] = w.permute(2, 3, 1, 0).reshape(kh * kw, cin, cout)
    bp = torch.zeros(4, dtype=torch.float32, device=w.device)
    if bias is not None:
        bp[:cout] = bias
    return wp.contiguous(), bp


def conv2d_smalln_cl(x, wpacked, bias4, cout, k, n_img, h, w, *, act=ACT_NONE, out=None):
    """Convolution with <= 4 output channels (4x4x1 MFMA blocks); out rows have stride out.stride(0) >= cout."""
    lib = _lib()
    _chk(lib, x, wpacked, bias4, out)
    if out is None:
        out = torch.empty(n_img * h * w, 4, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_conv2d_smalln_cl_f32(_p(x), x.stride(0), x.shape[1], n_img, h, w, _p(wpacked), _p(bias4), _p(out),
                                            out.stride(0), cout, k, act, _stream(lib)), "lfdm_conv2d_smalln_cl_f32")
    return out


def deconv4x4s2_cl(src, packs, cout, n_img, hi, wi, *, bias=None, out=None):
    """ConvTranspose (1,4,4) stride (1,2,2) pad (0,1,1) as four parity 2x2 convolutions: ONE launch when `packs` is the
    stacked tensor of pack_deconv4_weight, four launches for the list of pack_deconv_weight."""
    if out is None:
        out = torch.empty(n_img * 4 * hi * wi, cout, dtype=torch.float32, device=src.device)
    if torch.is_tensor(packs):
        conv2d_cl(src, packs[0], cout, 2, 2, n_img, hi, wi, bias=bias, pad=(1, 1), out=out, hq=hi, wq=wi, ho=2 * hi, wo=2 * wi,
                  out_scale=2, deconv4=packs)
        return out
    for py, px, w in packs:
        conv2d_cl(src, w, cout, 2, 2, n_img, hi, wi, bias=bias, pad=(1 - py, 1 - px), out=out,
                  hq=hi, wq=wi, ho=2 * hi, wo=2 * wi, out_scale=2, out_off=(py, px))
    return out


def groupnorm_silu_cl(x, batch, gamma, beta, *, groups=8, scale_shift=None, residual=None, eps=1e-5,
                      silu=True, out=None, ws=None):
    lib = _lib()
    rows, ch = x.shape
    pixels = rows // batch
    _chk(lib, x, gamma, beta, scale_shift, residual, out, ws)
    if out is None:
        out = torch.empty_like(x)
    need = lib.lfdm_groupnorm_ws_bytes(batch, pixels, ch)
    if ws is None:
        ws = torch.empty(need // 4, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_groupnorm_silu_cl_f32(_p(x), _p(out), batch, pixels, ch, groups, _p(gamma),
                                             _p(beta), _p(scale_shift),
                                             scale_shift.stride(0) if scale_shift is not None else 0,
                                             _p(residual), eps, int(silu), _p(ws),
                                             ws.numel() * 4, _stream(lib)), "lfdm_groupnorm_silu_cl_f32")
    return out


def groupnorm_apply_cl(x, batch, gamma, beta, partial, nchunk, *, groups=8, scale_shift=None, residual=None,
                       eps=1e-5, silu=True, out=None, ws=None):
    """GroupNorm whose statistics came from the producing convolution (gn_partial)."""
    lib = _lib()
    rows, ch = x.shape
    _chk(lib, x, gamma, beta, partial, scale_shift, residual, out, ws)
    if out is None:
        out = torch.empty_like(x)
    if ws is None:
        ws = torch.empty(batch * 2 * ch, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_groupnorm_apply_cl_f32(_p(x), _p(out), batch, rows // batch, ch, groups, _p(gamma), _p(beta),
                                              _p(scale_shift),
                                              scale_shift.stride(0) if scale_shift is not None else 0,
                                              _p(residual), eps, int(silu), _p(partial), nchunk, _p(ws),
                                              ws.numel() * 4, _stream(lib)), "lfdm_groupnorm_apply_cl_f32")
    return out


def layernorm_cl(x, gamma, eps=1e-5, out=None):
    lib = _lib()
    _chk(lib, x, gamma, out)
    if out is None:
        out = torch.empty_like(x)
    lib.check(lib.lfdm_layernorm_cl_f32(_p(x), _p(out), x.shape[0], x.shape[1], _p(gamma), eps,
                                        _stream(lib)), "lfdm_layernorm_cl_f32")
    return out


def attention_cl(qkv, batch, frames, hw, mode, *, bias=None, rot_cos=None, rot_sin=None, out=None):
    lib = _lib()
    _chk(lib, qkv, bias, rot_cos, rot_sin, out)
    assert qkv.shape[1] == 768 and qkv.is_contiguous()
    if out is None:
        out = torch.empty(qkv.shape[0], 256, dtype=torch.float32, device=qkv.device)
    lib.check(lib.lfdm_attention_cl_f32(_p(qkv), _p(out), batch, frames, hw, mode, _p(bias),
                                        _p(rot_cos), _p(rot_sin), _stream(lib)), "lfdm_attention_cl_f32")
    return out


def temporal_attention_fused_cl(x, wqkv, batch, frames, hw, *, bias=None, rot_cos=None, rot_sin=None, eps=1e-5, out=None):
    """LayerNorm + to_qkv + temporal attention in one launch (C in {64, 128}); wqkv (768, C) with gamma folded."""
    lib = _lib()
    _chk(lib, x, wqkv, bias, rot_cos, rot_sin, out)
    c = x.shape[1]
    assert wqkv.shape == (768, c) and wqkv.is_contiguous()
    if out is None:
        out = torch.empty(x.shape[0], 256, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_temporal_attention_fused_cl_f32(_p(x), x.stride(0), c, _p(wqkv), _p(out), batch, frames, hw, _p(bias),
                                                       _p(rot_cos), _p(rot_sin), eps, _stream(lib)),
              "lfdm_temporal_attention_fused_cl_f32")
    return out


def pack_pw_weight(packed):
    """Direct-form pack of a 1x1 filter ([K/32][coutp][32], pack_conv_weight) -> the MFMA-operand order of the pointwise schedule
    (lfdm_conv_params.weight_pw): [K/32][coutp/32][4 u][64 lanes = 32*kh + column][4 e], k % 32 = 8u + 4kh + e."""
    g, coutp, k32 = packed.shape
    assert k32 == 32 and coutp % 32 == 0
    # (g, ct, column, u, kh, e) -> (g, ct, u, kh, column, e)
    return packed.reshape(g, coutp // 32, 32, 4, 2, 4).permute(0, 1, 3, 4, 2, 5).contiguous().view(g, coutp // 32, 4, 64, 4)


def pack_tattn_weights(wqkv_folded, wout):
    """(768, 64) LayerNorm-folded to_qkv weight and (64, 256) to_out weight -> the MFMA-operand order of
    lfdm_temporal_attention_fused_out_cl_f32 (every fragment load = one contiguous 1 KB): ([3][8][2][4][64][4], [4][16][64][4])."""
    assert wqkv_folded.shape == (768, 64) and wout.shape == (64, 256)
    # rows: which(3) head(8) half(2) l15(16); columns: lq(4) quad(4) e(4)  ->  which head half quad (lq l15) e
    wq = wqkv_folded.float().reshape(3, 8, 2, 16, 4, 4, 4).permute(0, 1, 2, 5, 4, 3, 6).contiguous().view(3, 8, 2, 4, 64, 4)
    # rows: ct(4) l15(16); columns: S(16) lq(4) e(4)  ->  ct S (lq l15) e
    wo = wout.float().reshape(4, 16, 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous().view(4, 16, 64, 4)
    return wq, wo


def temporal_attention_fused_out_cl(x, wqkv, wout, batch, frames, hw, *, bias=None, rot_cos=None, rot_sin=None, eps=1e-5, out=None):
    """The whole temporal-attention block in one launch (C == 64): out = x + to_out(attention(LayerNorm(x))); wqkv, wout =
    pack_tattn_weights(W_qkv * gamma, W_out)."""
    lib = _lib()
    _chk(lib, x, wqkv, wout, bias, rot_cos, rot_sin, out)
    c = x.shape[1]
    assert c == 64 and wqkv.shape == (3, 8, 2, 4, 64, 4) and wqkv.is_contiguous() and wout.shape == (4, 16, 64, 4) and wout.is_contiguous()
    if out is None:
        out = torch.empty(x.shape[0], c, dtype=torch.float32, device=x.device)
    assert out.data_ptr() != x.data_ptr()
    lib.check(lib.lfdm_temporal_attention_fused_out_cl_f32(_p(x), x.stride(0), c, _p(wqkv), _p(wout), _p(out), out.stride(0), batch, frames, hw,
                                                           _p(bias), _p(rot_cos), _p(rot_sin), eps, _stream(lib)),
              "lfdm_temporal_attention_fused_out_cl_f32")
    return out


def pack_linattn_weights(wqkv_folded):
    """(768, 64) LayerNorm-folded to_qkv weight of SpatialLinearAttention -> the MFMA-operand order of lfdm_linear_attention_fused_cl_f32:
    [3 = q|k|v][8 heads][8 quads][64 lanes = 32*kh + l31][4]  <-  W[which*256 + head*32 + l31][32*kh + 4*quad + e]."""
    assert wqkv_folded.shape == (768, 64)
    # rows: which(3) head(8) l31(32); columns: kh(2) quad(8) e(4)  ->  which head quad (kh l31) e
    return wqkv_folded.float().reshape(3, 8, 32, 2, 8, 4).permute(0, 1, 4, 3, 2, 5).contiguous().view(3, 8, 8, 64, 4)


def linear_attention_fused_ws_floats(n_frames, hw):
    return (int(_lib().lfdm_linear_attention_fused_ws_bytes(n_frames, hw)) + 3) // 4


def linear_attention_fused_cl(x, wqkv, n_frames, hw, *, eps=1e-5, out=None, ws=None):
    """LayerNorm + to_qkv + linear attention core without materialising qkv (C == 64); wqkv = pack_linattn_weights(W_qkv * gamma)."""
    lib = _lib()
    _chk(lib, x, wqkv, out, ws)
    assert x.shape[1] == 64 and wqkv.shape == (3, 8, 8, 64, 4) and wqkv.is_contiguous()
    if out is None:
        out = torch.empty(x.shape[0], 256, dtype=torch.float32, device=x.device)
    need = lib.lfdm_linear_attention_fused_ws_bytes(n_frames, hw)
    if ws is not None:
        ws = ws.reshape(-1)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_linear_attention_fused_cl_f32(_p(x), x.stride(0), x.shape[1], _p(wqkv), _p(out), n_frames, hw, eps,
                                                     _p(ws), ws.numel() * 4, _stream(lib)),
              "lfdm_linear_attention_fused_cl_f32")
    return out


def pack_linattn_out_weight(wout):
    """(64, 256) to_out weight of SpatialLinearAttention -> the MFMA-operand order of lfdm_linear_attention_fused_out_cl_f32:
    [8 heads][2 row blocks][4 quads][64 lanes = 32*kh + c_local][4]  <-  Wout[32*cb + c_local][32*h + 8*quad + 4*kh + e]."""
    w = wout.float().reshape(64, 256)
    # rows: cb(2) c_local(32); columns: h(8) quad(4) kh(2) e(4)  ->  h cb quad (kh c_local) e
    return w.reshape(2, 32, 8, 4, 2, 4).permute(2, 0, 3, 4, 1, 5).contiguous().view(8, 2, 4, 64, 4)


def linear_attention_fused_out_cl(x, wqkv, wout, bias_out, n_frames, hw, *, eps=1e-5, out=None, ws=None):
    """The whole Residual(PreNorm(SpatialLinearAttention)) block at C == 64: out = x + to_out(linear_attention(LayerNorm(x))) + bias;
    wqkv = pack_linattn_weights(W_qkv * gamma), wout = pack_linattn_out_weight(W_out)."""
    lib = _lib()
    _chk(lib, x, wqkv, wout, bias_out, out, ws)
    assert x.shape[1] == 64 and wqkv.shape == (3, 8, 8, 64, 4) and wqkv.is_contiguous() and wout.shape == (8, 2, 4, 64, 4) and wout.is_contiguous()
    if out is None:
        out = torch.empty(x.shape[0], 64, dtype=torch.float32, device=x.device)
    assert out.data_ptr() != x.data_ptr()
    need = lib.lfdm_linear_attention_fused_ws_bytes(n_frames, hw)
    if ws is not None:
        ws = ws.reshape(-1)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_linear_attention_fused_out_cl_f32(_p(x), x.stride(0), x.shape[1], _p(wqkv), _p(wout), _p(bias_out), _p(out), out.stride(0),
                                                         n_frames, hw, eps, _p(ws), ws.numel() * 4, _stream(lib)),
              "lfdm_linear_attention_fused_out_cl_f32")
    return out


def linear_attention_lowres_cl(x, wqkv, wsum, n_frames, hw, *, eps=1e-5, out=None):
    """LayerNorm + to_qkv + linear attention core in ONE launch (low-resolution levels: hw <= 64 or 192 < hw <= 256 pixels per frame,
    C % 64 == 0); wqkv (768, C) with gamma folded, wsum (768,) = its row sums (pack_ln_conv_weight)."""
    lib = _lib()
    _chk(lib, x, wqkv, wsum, out)
    assert wqkv.shape == (768, x.shape[1]) and wqkv.is_contiguous() and wsum.numel() >= 768
    if out is None:
        out = torch.empty(x.shape[0], 256, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_linear_attention_lowres_cl_f32(_p(x), x.stride(0), x.shape[1], _p(wqkv), _p(wsum), _p(out), n_frames, hw, eps,
                                                      _stream(lib)), "lfdm_linear_attention_lowres_cl_f32")
    return out


def linear_attention_lowres_ok(hw, channels):
    return channels % 64 == 0 and (hw <= 64 or 192 < hw <= 256)


def attention_lowres_cl(x, wqkv, wsum, batch, frames, hw, mode, *, bias=None, rot_cos=None, rot_sin=None, eps=1e-5, out=None):
    """LayerNorm + to_qkv + softmax attention core in ONE launch (<= 64 tokens per sequence, C % 64 == 0): mode 0 over the frames of
    a pixel (rotary tables + relative-position bias), mode 1 over the pixels of a frame."""
    lib = _lib()
    _chk(lib, x, wqkv, wsum, bias, rot_cos, rot_sin, out)
    assert wqkv.shape == (768, x.shape[1]) and wqkv.is_contiguous() and wsum.numel() >= 768
    if out is None:
        out = torch.empty(x.shape[0], 256, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_attention_lowres_cl_f32(_p(x), x.stride(0), x.shape[1], _p(wqkv), _p(wsum), _p(out), batch, frames, hw, mode,
                                               _p(bias), _p(rot_cos), _p(rot_sin), eps, _stream(lib)), "lfdm_attention_lowres_cl_f32")
    return out


def linear_attention_cl(qkv, n_frames, hw, *, out=None, ws=None):
    lib = _lib()
    _chk(lib, qkv, out, ws)
    assert qkv.shape[1] == 768 and qkv.is_contiguous()
    if out is None:
        out = torch.empty(qkv.shape[0], 256, dtype=torch.float32, device=qkv.device)
    need = lib.lfdm_linear_attention_ws_bytes(n_frames)
    if ws is None:
        ws = torch.empty(need // 4, dtype=torch.float32, device=qkv.device)
    lib.check(lib.lfdm_linear_attention_cl_f32(_p(qkv), _p(out), n_frames, hw, _p(ws), ws.numel() * 4,
                                               _stream(lib)), "lfdm_linear_attention_cl_f32")
    return out


def linear_small(x, w, bias=None, *, act_in=ACT_NONE, act_out=ACT_NONE, out=None):
    lib = _lib()
    _chk(lib, x, w, bias, out)
    batch, k = x.shape
    n = w.shape[0]
    assert w.shape[1] == k and w.stride(1) == 1
    if out is None:
        out = torch.empty(batch, n, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_linear_small_f32(_p(x), _p(w), _p(bias), _p(out), batch, k, n, x.stride(0),
                                        w.stride(0), out.stride(0), act_in, act_out, _stream(lib)),
              "lfdm_linear_small_f32")
    return out


def step_cond(step_table, batch_base, step_dev, out):
    lib = _lib()
    _chk(lib, step_table, batch_base, step_dev, out)
    batch, n = batch_base.shape
    lib.check(lib.lfdm_step_cond_f32(_p(step_table), _p(batch_base), _p(step_dev), _p(out), batch, n,
                                     _stream(lib)), "lfdm_step_cond_f32")
    return out


def calib_mfma(device, blocks=256, iters=4000, reps=3):
    """Box calibration (lfdm_calib_mfma_f32; never on the product path): the fp32 matrix rate and the shader clock this GPU holds
    under a pure v_mfma_f32_32x32x2_f32 load RIGHT NOW.  Returns {"tflops", "mhz", "mhz_min", "mhz_max", "us"} (best of `reps`)."""
    lib = _lib()
    out = torch.zeros(2 * blocks + 256, dtype=torch.float32, device=device)
    _chk(lib, out)
    launch = lambda n: lib.check(lib.lfdm_calib_mfma_f32(_p(out), blocks, n, _stream(lib)), "lfdm_calib_mfma_f32")
    launch(64)
    # an idle chip sits at ~500 MHz and needs some milliseconds of load to ramp: 30 ms of the same kernel first (round 4, call A: the
    # un-warmed figure read 134 TFLOP/s at 2115 MHz on a box that holds 2390 MHz)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60):
        launch(iters)
    e1.record()
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch(iters)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        v = out[:2 * blocks].view(blocks, 2).double().cpu()
        mhz = 100.0 * v[:, 0] / v[:, 1].clamp(min=1.0)
        row = {"tflops": round(blocks * 4 * iters * 4 * 4096.0 / us / 1e6, 1), "mhz": round(float(mhz.median()), 0),
               "mhz_min": round(float(mhz.min()), 0), "mhz_max": round(float(mhz.max()), 0), "us": round(us, 1)}
        if best is None or row["tflops"] > best["tflops"]:
            best = row
    return best


def sinusoidal_freqs(dim, device):
    """fp32 frequency table, computed with the reference's expression (video_flow_diffusion.py:148-150)."""
    import math
    half = dim // 2
    return torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1))).to(device)


def sinusoidal(t_dev, freqs, batch, dim, *, t_stride=1, out=None):
    lib = _lib()
    _chk(lib, t_dev, freqs, out)
    assert t_dev.dtype == torch.int32
    if out is None:
        out = torch.empty(batch, dim, dtype=torch.float32, device=t_dev.device)
    lib.check(lib.lfdm_sinusoidal_f32(_p(t_dev), t_stride, _p(freqs), _p(out), batch, dim, out.stride(0),
                                      _stream(lib)), "lfdm_sinusoidal_f32")
    return out


def conv_planar_in_cl(x, batch, cin, cin_total, frames, h, w, wgt, kh, kw, cout, *, bias=None,
                      add_term=None, act=ACT_NONE, out=None):
    lib = _lib()
    _chk(lib, x, wgt, bias, add_term, out)
    if out is None:
        out = torch.empty(batch * frames * h * w, cout, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_conv_planar_in_cl_f32(_p(x), batch, cin, cin_total, frames, h, w, _p(wgt), kh, kw,
                                             cout, _p(bias), _p(add_term), _p(out), out.stride(0), act,
                                             _stream(lib)), "lfdm_conv_planar_in_cl_f32")
    return out


def heads_cl_to_planar(y_flow, y_occ, w_flow, b_flow, w_occ, b_occ, batch, frames, hw, *, out=None):
    """y_flow / y_occ: (rows, C) each, equal row stride (column slices of one (rows, 2C) tensor are fine)."""
    lib = _lib()
    _chk(lib, y_flow, y_occ, w_flow, b_flow, w_occ, b_occ, out)
    ch = y_flow.shape[1]
    assert y_flow.stride(0) == y_occ.stride(0) and y_flow.stride(1) == 1 and y_occ.stride(1) == 1
    if out is None:
        out = torch.empty(batch, 3, frames, hw, dtype=torch.float32, device=y_flow.device)
    lib.check(lib.lfdm_heads_cl_to_planar_f32(_p(y_flow), _p(y_occ), ch, y_flow.stride(0), _p(w_flow), _p(b_flow),
                                              _p(w_occ), _p(b_occ), _p(out), batch, frames, hw,
                                              _stream(lib)), "lfdm_heads_cl_to_planar_f32")
    return out


def heads_res_cl_to_planar(y_flow, y_occ, w_flow, b_flow, w_occ, b_occ, x0, x1, w_extra, batch, frames, hw, *, out=None):
    """heads_cl_to_planar with the ResnetBlocks' res_conv(cat(x0, x1)) folded in: w_extra (3, c0 + c1) = [W_flow Wres_flow ; W_occ Wres_occ],
    b_flow / b_occ include W1 bres (lfdm_heads_res_cl_to_planar_f32)."""
    lib = _lib()
    _chk(lib, y_flow, y_occ, w_flow, b_flow, w_occ, b_occ, x0, x1, w_extra, out)
    ch = y_flow.shape[1]
    c0, c1 = x0.shape[1], (x1.shape[1] if x1 is not None else 0)
    assert y_flow.stride(0) == y_occ.stride(0) and y_flow.stride(1) == 1 and y_occ.stride(1) == 1
    assert w_extra.shape == (3, c0 + c1) and w_extra.is_contiguous() and x0.stride(1) == 1
    if out is None:
        out = torch.empty(batch, 3, frames, hw, dtype=torch.float32, device=y_flow.device)
    lib.check(lib.lfdm_heads_res_cl_to_planar_f32(_p(y_flow), _p(y_occ), ch, y_flow.stride(0), _p(w_flow), _p(b_flow), _p(w_occ), _p(b_occ),
                                                  _p(x0), x0.stride(0), c0, _p(x1), x1.stride(0) if x1 is not None else 0, c1, _p(w_extra),
                                                  _p(out), batch, frames, hw, _stream(lib)), "lfdm_heads_res_cl_to_planar_f32")
    return out


def heads_gn_res_cl_to_planar(y, partial, nchunk, gamma, beta, w_flow, b_flow, w_occ, b_occ, x0, x1, w_extra, batch, frames, hw, *, groups=16,
                              eps=1e-5, out=None):
    """heads_res_cl_to_planar with the preceding GroupNorm + SiLU folded in: y (rows, 2C) = the RAW second convolution of the merged heads block,
    partial / nchunk = its fused statistics, gamma / beta (2C,) (lfdm_heads_gn_res_cl_to_planar_f32)."""
    lib = _lib()
    _chk(lib, y, partial, gamma, beta, w_flow, b_flow, w_occ, b_occ, x0, x1, w_extra, out)
    ch = y.shape[1] // 2
    c0, c1 = x0.shape[1], (x1.shape[1] if x1 is not None else 0)
    assert y.stride(1) == 1 and y.shape[1] == 2 * ch and gamma.numel() == 2 * ch == beta.numel() and partial.is_contiguous()
    assert w_extra.shape == (3, c0 + c1) and w_extra.is_contiguous() and x0.stride(1) == 1 and w_flow.numel() == 2 * ch and w_occ.numel() == ch
    if out is None:
        out = torch.empty(batch, 3, frames, hw, dtype=torch.float32, device=y.device)
    lib.check(lib.lfdm_heads_gn_res_cl_to_planar_f32(_p(y), y.stride(0), ch, _p(partial), nchunk, groups, _p(gamma), _p(beta), eps, _p(w_flow), _p(b_flow),
                                                     _p(w_occ), _p(b_occ), _p(x0), x0.stride(0), c0, _p(x1), x1.stride(0) if x1 is not None else 0, c1,
                                                     _p(w_extra), _p(out), batch, frames, hw, _stream(lib)), "lfdm_heads_gn_res_cl_to_planar_f32")
    return out


def sampler_ws(batch, n, device):
    """Workspace of sampler_step / abs_quantile, initialised (lfdm_sampler_ws_init: histograms + end-of-step ticket cleared)."""
    lib = _lib()
    ws = torch.empty((lib.lfdm_sampler_ws_bytes(batch, n) + 3) // 4, dtype=torch.float32, device=device)
    _chk(lib, ws)
    lib.check(lib.lfdm_sampler_ws_init(_p(ws), ws.numel() * 4, batch, n, _stream(lib)), "lfdm_sampler_ws_init")
    return ws


def sampler_step(x, eps, noise, coef, step_dev, *, quantile=0.9, advance=True, x0_out=None, ws=None):
    lib = _lib()
    _chk(lib, x, eps, noise, coef, step_dev, x0_out, ws)
    batch = x.shape[0]
    n = x.numel() // batch
    if ws is None:
        ws = sampler_ws(batch, n, x.device)
    lib.check(lib.lfdm_sampler_step_f32(_p(x), _p(eps), _p(noise), _p(x0_out), batch, n, _p(coef),
                                        _p(step_dev), quantile, int(advance), _p(ws), ws.numel() * 4,
                                        _stream(lib)), "lfdm_sampler_step_f32")
    return x


def cfg_combine(cond_eps, null_eps, scale, out):
    lib = _lib()
    _chk(lib, cond_eps, null_eps, out)
    lib.check(lib.lfdm_cfg_combine_f32(_p(cond_eps), _p(null_eps), float(scale), _p(out), cond_eps.numel(),
                                       _stream(lib)), "lfdm_cfg_combine_f32")
    return out


def abs_quantile(x, quantile=0.9, ws=None):
    lib = _lib()
    batch = x.shape[0]
    n = x.numel() // batch
    _chk(lib, x, ws)
    if ws is None:
        ws = sampler_ws(batch, n, x.device)
    out = torch.empty(batch, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_abs_quantile_f32(_p(x), batch, n, quantile, _p(out), _p(ws), ws.numel() * 4,
                                        _stream(lib)), "lfdm_abs_quantile_f32")
    return out


def _warp_params(src, out, batch, frames, h, w, c, flow_x, flow_y, occ, fh, fw, fsb, fst, prev,
                 occ_scale, occ_bias, ld_src, ld_prev, ld_out, prev_is_cl):
    p = WarpParams()
    p.src, p.prev, p.out = _p(src), _p(prev), _p(out)
    p.batch, p.frames, p.h, p.w, p.c = batch, frames, h, w, c
    p.ld_src, p.ld_prev, p.ld_out = ld_src, ld_prev, ld_out
    p.flow_x, p.flow_y, p.occ = _p(flow_x), _p(flow_y), _p(occ)
    p.fh, p.fw, p.fsb, p.fst = fh, fw, fsb, fst
    p.occ_scale, p.occ_bias, p.prev_is_cl = occ_scale, occ_bias, int(prev_is_cl)
    return p


def warp_cl(src, batch, frames, h, w, flow_x, flow_y, occ, fh, fw, fsb, fst, *, prev=None,
            occ_scale=1.0, occ_bias=0.0, out=None):
    """src: CL (batch*h*w, C); out/prev: CL (batch*frames*h*w, C). flow_x/flow_y/occ: tensors whose
    data_ptr is the map base (element (b,t,y,x) at b*fsb + t*fst + y*fw + x)."""
    lib = _lib()
    _chk(lib, src, flow_x, flow_y, occ, prev, out)
    c = src.shape[1]
    if out is None:
        out = torch.empty(batch * frames * h * w, c, dtype=torch.float32, device=src.device)
    p = _warp_params(src, out, batch, frames, h, w, c, flow_x, flow_y, occ, fh, fw, fsb, fst, prev,
                     occ_scale, occ_bias, src.stride(0), prev.stride(0) if prev is not None else 0,
                     out.stride(0), True)
    lib.check(lib.lfdm_warp_cl_f32(C.byref(p), _stream(lib)), "lfdm_warp_cl_f32")
    return out


def warp_planar(src, frames, flow_x, flow_y, occ, fh, fw, fsb, fst, *, prev=None, prev_is_cl=False,
                occ_scale=1.0, occ_bias=0.0, out=None):
    """src: planar (B, C, H, W); out: planar (B, C, frames, H, W)."""
    lib = _lib()
    _chk(lib, src, flow_x, flow_y, occ, prev, out)
    b, c, h, w = src.shape
    assert src.is_contiguous()
    if out is None:
        out = torch.empty(b, c, frames, h, w, dtype=torch.float32, device=src.device)
    ld_prev = prev.stride(0) if (prev is not None and prev_is_cl) else 0
    p = _warp_params(src, out, b, frames, h, w, c, flow_x, flow_y, occ, fh, fw, fsb, fst, prev,
                     occ_scale, occ_bias, 0, ld_prev, 0, prev_is_cl)
    lib.check(lib.lfdm_warp_planar_f32(C.byref(p), _stream(lib)), "lfdm_warp_planar_f32")
    return out


def affine_act_cl(x, a, b, act=ACT_RELU, out=None):
    lib = _lib()
    _chk(lib, x, a, b, out)
    if out is None:
        out = torch.empty(x.shape[0], x.shape[1], dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_affine_act_cl_f32(_p(x), _p(out), x.shape[0], x.shape[1], x.stride(0),
                                         out.stride(0), _p(a), _p(b), act, _stream(lib)), "lfdm_affine_act_cl_f32")
    return out


def avgpool2_cl(x, n_img, h, w, out=None):
    lib = _lib()
    _chk(lib, x, out)
    assert x.is_contiguous()
    c = x.shape[1]
    if out is None:
        out = torch.empty(n_img * (h // 2) * (w // 2), c, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_avgpool2_cl_f32(_p(x), _p(out), n_img, h, w, c, _stream(lib)), "lfdm_avgpool2_cl_f32")
    return out


def planar_to_cl(x, n_img, channels, hw, out=None):
    lib = _lib()
    _chk(lib, x, out)
    if out is None:
        out = torch.empty(n_img * hw, channels, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_planar_to_cl_f32(_p(x), _p(out), n_img, channels, hw, out.stride(0), _stream(lib)),
              "lfdm_planar_to_cl_f32")
    return out


def cl_to_planar(x, n_img, channels, hw, out=None):
    lib = _lib()
    _chk(lib, x, out)
    if out is None:
        out = torch.empty(n_img, channels, hw, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_cl_to_planar_f32(_p(x), _p(out), n_img, channels, hw, x.stride(0), _stream(lib)),
              "lfdm_cl_to_planar_f32")
    return out


def depthwise_down_planar(x, weight, stride, pad_lo, pad_hi):
    """AntiAliasInterpolation2d core: x (N,C,H,W) planar, weight (C,1,k,k) -> (N,C,Ho,Wo)."""
    lib = _lib()
    x = x.contiguous()
    wt = weight.reshape(weight.shape[0], weight.shape[-2], weight.shape[-1]).contiguous()
    _chk(lib, x, wt)
    n, c, h, w = x.shape
    k = wt.shape[-1]
    ho = (h + pad_lo + pad_hi - k + 1 + stride - 1) // stride
    wo = (w + pad_lo + pad_hi - k + 1 + stride - 1) // stride
    out = torch.empty(n, c, ho, wo, dtype=torch.float32, device=x.device)
    lib.check(lib.lfdm_depthwise_down_planar_f32(_p(x), _p(wt), _p(out), n, c, h, w, k, pad_lo, pad_hi, stride, _stream(lib)),
              "lfdm_depthwise_down_planar_f32")
    return out
