"""CPU ORACLE - TEST INFRASTRUCTURE ONLY.

A functional, fp32, torch-CPU restatement of the reference algorithm for the LFDM hot path
(SURVEY.md section 8a).  It exists only to *check* the HIP path: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product package
(cvpr23_lfdm_amd/) never does.

Parity status: PINNED against the real reference - oracle/make_golden.py imports the
unmodified reference from /root/reference (CPU, import shims in oracle/ref_shims/) and
writes the fixtures in tests/golden/; tests/test_oracle_golden.py checks every function
here against those fixtures, and tests/test_oracle_vs_reference.py checks them against the
live reference when /root/reference is present.  Third-party arithmetic that is not vendored
in the reference (rotary_embedding_torch==0.1.5, torch ATen ops) is restated from its
published algorithm and is "parity unpinned" by the reference's own tests (it has none).

All weights are addressed by the reference's state-dict keys so the
same synthetic checkpoint drives the reference, this oracle and the HIP path.

Layouts: UNet tensors (B, C, T, H, W); LFAE tensors (B, C, H, W); flow grids (B, h, w, 2).
"""
import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# elementary ops restated at formula level (these are what the HIP kernels implement)
# --------------------------------------------------------------------------------------


def bilinear_resize(x, out_h, out_w):
    """F.interpolate(mode='bilinear', align_corners=False) restated
    (reference call sites: LFAE/modules/generator.py:65,80).
    src = (dst + 0.5) * in/out - 0.5, clamped below at 0; upper tap clamped to in-1."""
    b, c, h, w = x.shape
    if (h, w) == (out_h, out_w):
        return x

    def taps(n_in, n_out):
        dst = torch.arange(n_out, dtype=torch.float32)
        src = ((dst + 0.5) * (float(n_in) / float(n_out)) - 0.5).clamp_min(0.0)
        i0 = src.floor().long().clamp_max(n_in - 1)
        i1 = (i0 + 1).clamp_max(n_in - 1)
        frac = src - i0.float()
        return i0, i1, frac

    y0, y1, fy = taps(h, out_h)
    x0, x1, fx = taps(w, out_w)
    rows0, rows1 = x[:, :, y0, :], x[:, :, y1, :]
    fy = fy.view(1, 1, -1, 1)
    rows = rows0 * (1.0 - fy) + rows1 * fy
    fx = fx.view(1, 1, 1, -1)
    return rows[:, :, :, x0] * (1.0 - fx) + rows[:, :, :, x1] * fx


def grid_sample_bilinear_zeros(inp, grid):
    """F.grid_sample(inp, grid) with the defaults the reference relies on
    (bilinear, padding_mode='zeros', align_corners=False; LFAE/modules/generator.py:67).
    ix = ((x+1)*W-1)/2, iy = ((y+1)*H-1)/2; taps outside the image contribute 0."""
    b, c, h, w = inp.shape
    gx, gy = grid[..., 0], grid[..., 1]
    ix = ((gx + 1.0) * w - 1.0) * 0.5
    iy = ((gy + 1.0) * h - 1.0) * 0.5
    x0 = ix.floor()
    y0 = iy.floor()
    wx1 = ix - x0
    wy1 = iy - y0
    wx0 = 1.0 - wx1
    wy0 = 1.0 - wy1
    flat = inp.reshape(b, c, h * w)
    out = torch.zeros(b, c, grid.shape[1], grid.shape[2], dtype=inp.dtype)
    for dy, wy in ((0, wy0), (1, wy1)):
        for dx, wx in ((0, wx0), (1, wx1)):
            xi = (x0 + dx).long()
            yi = (y0 + dy).long()
            ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
            idx = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)).view(b, 1, -1).expand(b, c, -1)
            vals = flat.gather(2, idx).view(b, c, grid.shape[1], grid.shape[2])
            out = out + vals * (wy * wx * ok.float()).unsqueeze(1)
    return out


def deform_input(inp, flow):
    """Generator.deform_input (LFAE/modules/generator.py:59-67): flow (B,h,w,2) is an
    absolute sampling grid; it is bilinearly resized to the input resolution first."""
    _, h_old, w_old, _ = flow.shape
    _, _, h, w = inp.shape
    if (h_old, w_old) != (h, w):
        flow = bilinear_resize(flow.permute(0, 3, 1, 2), h, w).permute(0, 2, 3, 1)
    return grid_sample_bilinear_zeros(inp, flow)


def apply_optical(prev, skip, flow, occ):
    """Generator.apply_optical with motion_params present (generator.py:69-88)."""
    warped = deform_input(skip, flow)
    if occ is not None:
        if occ.shape[2:] != warped.shape[2:]:
            occ = bilinear_resize(occ, warped.shape[2], warped.shape[3])
        warped = warped * occ + prev * (1.0 - occ) if prev is not None else warped * occ
    return warped


def abs_quantile(x_flat, q):
    """torch.quantile(|x|, q, dim=-1) with linear interpolation
    (video_flow_diffusion.py:722-726): pos = q*(n-1); lerp(sorted[floor], sorted[ceil])."""
    srt = x_flat.abs().sort(dim=-1).values
    n = srt.shape[-1]
    # aten evaluates the rank in the input dtype: fp32(q) * (n - 1), then floor / lerp
    pos = torch.tensor(q, dtype=x_flat.dtype) * (n - 1)
    lo = pos.floor()
    frac = pos - lo
    lo_i = int(lo)
    hi_i = min(lo_i + 1, n - 1)
    return torch.lerp(srt[:, lo_i], srt[:, hi_i], frac)


def dynamic_threshold(x0, q=0.9, dynamic=True):
    """video_flow_diffusion.py:719-732 / :805-818: s = max(1, quantile_q |x0|) per sample with use_dynamic_thres=True (what
    FlowDiffusion passes), s = 1 otherwise (GaussianDiffusion's own default); then x0.clamp(-s, s) / s."""
    if not dynamic:
        return x0.clamp(-1.0, 1.0)
    s = abs_quantile(x0.reshape(x0.shape[0], -1), q).clamp_min(1.0)
    s = s.view(-1, *([1] * (x0.dim() - 1)))
    return torch.maximum(torch.minimum(x0, s), -s) / s


# --------------------------------------------------------------------------------------
# schedule (video_flow_diffusion.py:598-608, :635-680)
# --------------------------------------------------------------------------------------

SCHEDULE_KEYS = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
    "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
    "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
)


def make_schedule(timesteps=1000, s=0.008):
    """fp64 cosine schedule -> the 12 fp32 buffers the reference registers."""
    steps = timesteps + 1
    x = torch.linspace(0, timesteps, steps, dtype=torch.float64)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.9999)
    alphas = 1.0 - betas
    acp = torch.cumprod(alphas, dim=0)
    acp_prev = F.pad(acp[:-1], (1, 0), value=1.0)
    post_var = betas * (1.0 - acp_prev) / (1.0 - acp)
    vals = {
        "betas": betas,
        "alphas_cumprod": acp,
        "alphas_cumprod_prev": acp_prev,
        "sqrt_alphas_cumprod": acp.sqrt(),
        "sqrt_one_minus_alphas_cumprod": (1.0 - acp).sqrt(),
        "log_one_minus_alphas_cumprod": (1.0 - acp).log(),
        "sqrt_recip_alphas_cumprod": (1.0 / acp).sqrt(),
        "sqrt_recipm1_alphas_cumprod": (1.0 / acp - 1).sqrt(),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": post_var.clamp(min=1e-20).log(),
        "posterior_mean_coef1": betas * acp_prev.sqrt() / (1.0 - acp),
        "posterior_mean_coef2": (1.0 - acp_prev) * alphas.sqrt() / (1.0 - acp),
    }
    return {k: v.to(torch.float32) for k, v in vals.items()}


def ddim_time_pairs(total_timesteps, sampling_timesteps):
    """video_flow_diffusion.py:784-786."""
    times = torch.linspace(0.0, total_timesteps, steps=sampling_timesteps + 2)[:-1]
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


# --------------------------------------------------------------------------------------
# UNet pieces (DM/modules/video_flow_diffusion.py)
# --------------------------------------------------------------------------------------

HEADS = 8
DIM_HEAD = 32
HIDDEN = HEADS * DIM_HEAD


def rel_pos_bucket(rel, num_buckets=32, max_distance=32):
    """RelativePositionBias._relative_position_bucket (:85-102); rel = k_pos - q_pos."""
    n = -rel
    half = num_buckets // 2
    ret = (n < 0).long() * half
    n = n.abs()
    max_exact = half // 2
    large = max_exact + (
        torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (half - max_exact)
    ).long()
    large = torch.minimum(large, torch.full_like(large, half - 1))
    return ret + torch.where(n < max_exact, n, large)


def rel_pos_bias(emb_weight, n):
    """RelativePositionBias.forward (:104-111) -> (heads, n, n)."""
    pos = torch.arange(n)
    bucket = rel_pos_bucket(pos[None, :] - pos[:, None])
    return emb_weight[bucket].permute(2, 0, 1).contiguous()


def rotary_tables(freqs, n):
    """cos/sin tables (n, 32) of rotary_embedding_torch (pairs share an angle)."""
    ang = torch.arange(n).float()[:, None] * freqs[None, :]
    ang = ang.repeat_interleave(2, dim=-1)
    return ang.cos(), ang.sin()


def apply_rotary(t, cos, sin):
    even, odd = t[..., 0::2], t[..., 1::2]
    swapped = torch.stack((-odd, even), dim=-1).flatten(-2)
    return t * cos + swapped * sin


def channel_layernorm(x, gamma, eps=1e-5):
    """LayerNorm over dim=1, biased variance, gamma only (:170-179)."""
    var = x.var(dim=1, unbiased=False, keepdim=True)
    mean = x.mean(dim=1, keepdim=True)
    return (x - mean) / (var + eps).sqrt() * gamma


def softmax_attention(tokens, w_qkv, w_out, pos_bias=None, rotary=None, focus=None):
    """Attention.forward (:303-363).  tokens (..., n, C) - for the temporal form (b, h*w, f, C); w_qkv (768, C); w_out (C, 256);
    focus: None or the (b,) bool focus_present_mask (:313-317, :342-352: a focused sample attends only to itself)."""
    qkv = F.linear(tokens, w_qkv)                        # (nn.Linear in the reference; `tokens @ w.t()` took torch.matmul's expanded-weight
    q, k, v = qkv.chunk(3, dim=-1)                       #  bmm path on the CPU: 0.5 s of a 1.7 s forward)
    if focus is not None and bool(focus.all()):          # :313-317: the values pass straight through to_out
        return F.linear(v, w_out)

    def heads(z):
        # (contiguous, as einops.rearrange leaves it in the reference: torch.bmm on a batch whose strides do not fold walks the 8 192
        #  (pixel, head) pairs one by one - 0.6 s of a 1.7 s forward on the CPU)
        return z.reshape(*z.shape[:-1], HEADS, DIM_HEAD).transpose(-2, -3).contiguous()  # (..., h, n, d)

    q, k, v = heads(q), heads(k), heads(v)
    q = q * (DIM_HEAD ** -0.5)
    if rotary is not None:
        cos, sin = rotary
        q, k = apply_rotary(q, cos, sin), apply_rotary(k, cos, sin)
    sim = q @ k.transpose(-1, -2)
    if pos_bias is not None:
        sim = sim + pos_bias
    if focus is not None and not bool((~focus).all()):  # :342-352
        n = sim.shape[-1]
        mask = torch.where(focus.view(-1, 1, 1, 1, 1), torch.eye(n, dtype=torch.bool).view(1, 1, 1, n, n),
                           torch.ones(n, n, dtype=torch.bool).view(1, 1, 1, n, n))
        sim = sim.masked_fill(~mask, -torch.finfo(sim.dtype).max)
    sim = sim - sim.amax(dim=-1, keepdim=True)
    attn = sim.softmax(dim=-1)
    out = attn @ v                                   # (..., h, n, d)
    out = out.transpose(-2, -3).reshape(*tokens.shape[:-1], HIDDEN)
    return F.linear(out, w_out)


def temporal_attention(x, sd, prefix, pos_bias, rotary, focus=None):
    """Residual(PreNorm(EinopsToAndFrom('b c f h w','b (h w) f c', Attention))) (:397-399,413)."""
    b, c, f, h, w = x.shape
    normed = channel_layernorm(x, sd[prefix + "fn.norm.gamma"])
    tokens = normed.permute(0, 3, 4, 2, 1).reshape(b, h * w, f, c).contiguous()      # (a strided view otherwise: see softmax_attention)
    out = softmax_attention(tokens, sd[prefix + "fn.fn.fn.to_qkv.weight"],
                            sd[prefix + "fn.fn.fn.to_out.weight"], pos_bias, rotary, focus)
    out = out.reshape(b, h, w, f, c).permute(0, 4, 3, 1, 2)
    return out + x


def mid_spatial_attention(x, sd, prefix):
    """mid_spatial_attn: 'b c f h w' -> 'b f (h w) c', no rotary, no bias (:473-475)."""
    b, c, f, h, w = x.shape
    normed = channel_layernorm(x, sd[prefix + "fn.norm.gamma"])
    tokens = normed.permute(0, 2, 3, 4, 1).reshape(b, f, h * w, c).contiguous()
    out = softmax_attention(tokens, sd[prefix + "fn.fn.fn.to_qkv.weight"],
                            sd[prefix + "fn.fn.fn.to_out.weight"])
    out = out.reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3)
    return out + x


def spatial_linear_attention(x, sd, prefix):
    """Residual(PreNorm(SpatialLinearAttention)) (:240-265): per frame, q softmax over the
    32-dim axis, k softmax over tokens, ctx = k v^T, out = ctx^T q, 1x1 out conv with bias."""
    b, c, f, h, w = x.shape
    normed = channel_layernorm(x, sd[prefix + "fn.norm.gamma"])
    frames = normed.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    # (to_qkv / to_out are nn.Conv2d 1x1 in the reference (:246-247): the same ATen convolution here - as a broadcast einsum this was a
    #  40-batch bmm with an expanded weight, 1.7x the reference's wall time for the whole forward, profiles/port_vs_reference_cpu.json)
    qkv = F.conv2d(frames, sd[prefix + "fn.fn.to_qkv.weight"].reshape(3 * HIDDEN, c, 1, 1))
    q, k, v = [z.reshape(b * f, HEADS, DIM_HEAD, h * w) for z in qkv.chunk(3, dim=1)]
    q = q.softmax(dim=-2) * (DIM_HEAD ** -0.5)
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b * f, HIDDEN, h, w)
    out = F.conv2d(out, sd[prefix + "fn.fn.to_out.weight"].reshape(c, HIDDEN, 1, 1), sd[prefix + "fn.fn.to_out.bias"])
    out = out.reshape(b, f, c, h, w).permute(0, 2, 1, 3, 4)
    return out + x


def block(x, sd, prefix, scale_shift=None):
    """Block (:196-211): conv(1,3,3) -> GroupNorm(8) over (C/8,T,H,W) -> x*(scale+1)+shift -> SiLU."""
    x = F.conv3d(x, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"], padding=(0, 1, 1))
    x = F.group_norm(x, 8, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], eps=1e-5)
    if scale_shift is not None:
        scale, shift = scale_shift
        x = x * (scale + 1) + shift
    return F.silu(x)


def resnet_block(x, sd, prefix, t_emb=None):
    """ResnetBlock (:214-237)."""
    scale_shift = None
    if (prefix + "mlp.1.weight") in sd:
        e = F.linear(F.silu(t_emb), sd[prefix + "mlp.1.weight"], sd[prefix + "mlp.1.bias"])
        e = e.view(e.shape[0], -1, 1, 1, 1)
        scale_shift = e.chunk(2, dim=1)
    h = block(x, sd, prefix + "block1.", scale_shift)
    h = block(h, sd, prefix + "block2.")
    if (prefix + "res_conv.weight") in sd:
        x = F.conv3d(x, sd[prefix + "res_conv.weight"], sd[prefix + "res_conv.bias"])
    return h + x


def time_embedding(time, sd, prefix, dim=64):
    """SinusoidalPosEmb (:141-153) + time_mlp (:423-428)."""
    half = dim // 2
    freq = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    e = time[:, None].float() * freq[None, :]
    e = torch.cat((e.sin(), e.cos()), dim=-1)
    e = F.linear(e, sd[prefix + "time_mlp.1.weight"], sd[prefix + "time_mlp.1.bias"])
    e = F.gelu(e)
    return F.linear(e, sd[prefix + "time_mlp.3.weight"], sd[prefix + "time_mlp.3.bias"])


def unet_forward(sd, x, time, cond, null_mask=None, prefix="denoise_fn.", focus_mask=None):
    """Unet3D.forward (:528-588); focus_mask = the (B,) bool focus_present_mask (None = all False; it reaches every temporal attention
    of the down / mid / up path, not init_temporal_attn: :547, :570, :575, :584).
    x (B,259,T,S,S) = [3 noisy channels | 256 fea channels]; time (B,) long; cond (B,768);
    null_mask (B,) bool (True -> use the null condition embedding).
    The variant (deconv/zeros vs upconv/reflect, learned null cond) is inferred from the keys."""
    p = prefix
    b, _, nf, _, _ = x.shape
    bias = rel_pos_bias(sd[p + "time_rel_pos_bias.relative_attention_bias.weight"], nf)
    rotary = rotary_tables(sd[p + "init_temporal_attn.fn.fn.fn.rotary_emb.freqs"], nf)

    x = F.conv3d(x, sd[p + "init_conv.weight"], sd[p + "init_conv.bias"], padding=(0, 3, 3))
    r = x
    x = temporal_attention(x, sd, p + "init_temporal_attn.", bias, rotary)

    t = time_embedding(time, sd, p)
    null_emb = sd.get(p + "null_cond_emb", torch.zeros(1, cond.shape[1]))
    if null_mask is None:
        null_mask = torch.zeros(b, dtype=torch.bool)
    cond = torch.where(null_mask.view(b, 1), null_emb, cond)
    t = torch.cat((t, cond), dim=-1)

    skips = []
    n_levels = 4
    for lvl in range(n_levels):
        q = "%sdowns.%d." % (p, lvl)
        x = resnet_block(x, sd, q + "0.", t)
        x = resnet_block(x, sd, q + "1.", t)
        x = spatial_linear_attention(x, sd, q + "2.")
        x = temporal_attention(x, sd, q + "3.", bias, rotary, focus_mask)
        skips.append(x)
        if (q + "4.weight") in sd:
            x = F.conv3d(x, sd[q + "4.weight"], sd[q + "4.bias"], stride=(1, 2, 2), padding=(0, 1, 1))

    x = resnet_block(x, sd, p + "mid_block1.", t)
    x = mid_spatial_attention(x, sd, p + "mid_spatial_attn.")
    x = temporal_attention(x, sd, p + "mid_temporal_attn.", bias, rotary, focus_mask)
    x = resnet_block(x, sd, p + "mid_block2.", t)

    for lvl in range(n_levels):
        q = "%sups.%d." % (p, lvl)
        x = torch.cat((x, skips.pop()), dim=1)
        x = resnet_block(x, sd, q + "0.", t)
        x = resnet_block(x, sd, q + "1.", t)
        x = spatial_linear_attention(x, sd, q + "2.")
        x = temporal_attention(x, sd, q + "3.", bias, rotary, focus_mask)
        if (q + "4.weight") in sd:                      # ConvTranspose3d (:158)
            x = F.conv_transpose3d(x, sd[q + "4.weight"], sd[q + "4.bias"],
                                   stride=(1, 2, 2), padding=(0, 1, 1))
        elif (q + "4.1.weight") in sd:                  # nearest x2 + reflect conv (:160-163)
            x = F.interpolate(x, scale_factor=(1, 2, 2), mode="nearest")
            x = F.pad(x, (1, 1, 1, 1, 0, 0), mode="reflect")
            x = F.conv3d(x, sd[q + "4.1.weight"], sd[q + "4.1.bias"])

    x = torch.cat((x, r), dim=1)
    outs = []
    for head in ("final_conv.", "occlusion_map."):
        y = resnet_block(x, sd, p + head + "0.")
        outs.append(F.conv3d(y, sd[p + head + "1.weight"], sd[p + head + "1.bias"]))
    return torch.cat(outs, dim=1)


def unet_forward_with_cond_scale(sd, x, time, cond, cond_scale=1.0, prefix="denoise_fn."):
    """Unet3D.forward_with_cond_scale (:511-526)."""
    b = x.shape[0]
    ones = torch.ones(b, dtype=torch.bool)
    if cond_scale == 0:
        return unet_forward(sd, x, time, cond, ones, prefix)
    logits = unet_forward(sd, x, time, cond, ~ones, prefix)
    if cond_scale == 1:
        return logits
    null_logits = unet_forward(sd, x, time, cond, ones, prefix)
    return null_logits + (logits - null_logits) * cond_scale


# --------------------------------------------------------------------------------------
# samplers (GaussianDiffusion, video_flow_diffusion.py:697-830)
# --------------------------------------------------------------------------------------


def _bc(v, b):
    return v.reshape(b, 1, 1, 1, 1)


def predict_start_from_noise(sd, x_t, t, eps):
    b = x_t.shape[0]
    return (_bc(sd["sqrt_recip_alphas_cumprod"][t], b) * x_t
            - _bc(sd["sqrt_recipm1_alphas_cumprod"][t], b) * eps)


def ddim_step(sd, img, fea_rep, cond, time, time_next, noise, eta=1.0, cond_scale=1.0, dynamic=True):
    """One iteration of ddim_sample (:791-827). `noise` is the tensor the reference would draw
    with randn_like (ignored when time_next == 0). Returns (img_next, pred_noise, x_start)."""
    b = img.shape[0]
    alpha = sd["alphas_cumprod_prev"][time]
    alpha_next = sd["alphas_cumprod_prev"][time_next]
    t = torch.full((b,), time, dtype=torch.long)
    eps = unet_forward_with_cond_scale(sd, torch.cat([img, fea_rep], dim=1), t, cond, cond_scale)
    x0 = dynamic_threshold(predict_start_from_noise(sd, img, t, eps), dynamic=dynamic)
    sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
    c = ((1 - alpha_next) - sigma ** 2).sqrt()
    nz = noise if time_next > 0 else 0.0
    return x0 * alpha_next.sqrt() + c * eps + sigma * nz, eps, x0


def ddpm_step(sd, img, fea_rep, cond, time, noise, cond_scale=1.0, dynamic=True):
    """p_sample (:737-746) with p_mean_variance (:712-735)."""
    b = img.shape[0]
    t = torch.full((b,), time, dtype=torch.long)
    eps = unet_forward_with_cond_scale(sd, torch.cat([img, fea_rep], dim=1), t, cond, cond_scale)
    x0 = dynamic_threshold(predict_start_from_noise(sd, img, t, eps), dynamic=dynamic)
    mean = _bc(sd["posterior_mean_coef1"][t], b) * x0 + _bc(sd["posterior_mean_coef2"][t], b) * img
    logvar = _bc(sd["posterior_log_variance_clipped"][t], b)
    nonzero = 0.0 if time == 0 else 1.0
    return mean + nonzero * (0.5 * logvar).exp() * noise, eps, x0


def sample(sd, fea, cond, shape, sampling_timesteps, timesteps=1000, eta=1.0, cond_scale=1.0,
           noise_fn=None, record=None, dynamic=True):
    """GaussianDiffusion.sample (:762-775): DDIM when sampling_timesteps < timesteps else DDPM.
    Noise is drawn through `noise_fn(shape)` (default torch.randn on the default generator) in
    the reference's order: one draw for x_T, then one per step (DDIM skips
    the draw of the last step; DDPM draws on every step).  `record`, if a list, receives every
    drawn tensor so a GPU run can replay them."""
    noise_fn = noise_fn or (lambda s: torch.randn(s))

    def draw():
        z = noise_fn(tuple(shape))
        if record is not None:
            record.append(z)
        return z

    img = draw()
    fea_rep = fea.unsqueeze(2).repeat(1, 1, shape[2], 1, 1)
    if sampling_timesteps < timesteps:
        for time, time_next in ddim_time_pairs(timesteps, sampling_timesteps):
            z = draw() if time_next > 0 else None
            img, _, _ = ddim_step(sd, img, fea_rep, cond, time, time_next, z, eta, cond_scale, dynamic)
    else:
        for time in reversed(range(timesteps)):
            img, _, _ = ddpm_step(sd, img, fea_rep, cond, time, draw(), cond_scale, dynamic)
    return img


# --------------------------------------------------------------------------------------
# LFAE generator decode (LFAE/modules/generator.py, LFAE/modules/util.py)
# --------------------------------------------------------------------------------------


def _bn_eval(x, sd, prefix):
    """SynchronizedBatchNorm2d eval path = F.batch_norm with running stats
    (sync_batchnorm/batchnorm.py:50-53), eps 1e-5."""
    return F.batch_norm(x, sd[prefix + "running_mean"], sd[prefix + "running_var"],
                        sd[prefix + "weight"], sd[prefix + "bias"], False, 0.0, 1e-5)


def _conv(x, sd, prefix, pad):
    return F.conv2d(x, sd[prefix + "weight"], sd[prefix + "bias"], padding=pad)


def generator_encode(gsd, img, n_down=2):
    """first (SameBlock2d 7x7) + down blocks (util.py:115-150); returns all skips."""
    out = F.relu(_bn_eval(_conv(img, gsd, "first.conv.", 3), gsd, "first.norm."))
    skips = [out]
    for i in range(n_down):
        q = "down_blocks.%d." % i
        out = F.relu(_bn_eval(_conv(out, gsd, q + "conv.", 1), gsd, q + "norm."))
        out = F.avg_pool2d(out, 2)
        skips.append(out)
    return skips


def generator_compute_fea(gsd, img, n_down=2):
    """Generator.compute_fea (generator.py:130-134)."""
    return generator_encode(gsd, img, n_down)[-1]


def generator_forward_with_flow(gsd, img, flow, occ, n_down=2, n_bottleneck=6, use_skips=True):
    """Generator.forward_with_flow (generator.py:136-166); use_skips = the constructor's `skips` (:153, :156, :161).
    img (B,3,H,W); flow (B,h,w,2); occ (B,1,h,w) -> dict(prediction, deformed)."""
    skips = generator_encode(gsd, img, n_down)
    deformed = deform_input(img, flow)
    out = apply_optical(None, skips[-1], flow, occ)
    for i in range(n_bottleneck):                      # ResBlock2d (util.py:84-92)
        q = "bottleneck.r%d." % i
        y = _conv(F.relu(_bn_eval(out, gsd, q + "norm1.")), gsd, q + "conv1.", 1)
        y = _conv(F.relu(_bn_eval(y, gsd, q + "norm2.")), gsd, q + "conv2.", 1)
        out = y + out
    for i in range(n_down):                            # UpBlock2d (util.py:107-112)
        if use_skips:
            out = apply_optical(out, skips[-(i + 1)], flow, occ)
        q = "up_blocks.%d." % i
        out = F.interpolate(out, scale_factor=2)
        out = F.relu(_bn_eval(_conv(out, gsd, q + "conv.", 1), gsd, q + "norm."))
    if use_skips:
        out = apply_optical(out, skips[0], flow, occ)
    out = torch.sigmoid(F.conv2d(out, gsd["final.weight"], gsd["final.bias"], padding=3))
    if use_skips:
        out = apply_optical(out, img, flow, occ)
    return {"prediction": out, "deformed": deformed}


def identity_grid(b, nf, h, w):
    """FlowDiffusion.get_grid(normalize=True) (video_flow_diffusion_model.py:232-240): linspace(-1,1)
    meshgrid in (x, y) order, (B, 2, nf, H, W)."""
    ys, xs = torch.linspace(-1, 1, h), torch.linspace(-1, 1, w)
    grid = torch.stack((xs.view(1, w).expand(h, w), ys.view(h, 1).expand(h, w)), dim=0)
    return grid.view(1, 2, 1, h, w).repeat(b, 1, nf, 1, 1)


def sample_one_video(dsd, gsd, img, cond, num_frames, latent_size, sampling_timesteps,
                     timesteps=1000, eta=1.0, cond_scale=1.0, noise_fn=None, record=None,
                     use_residual_flow=False, dynamic=True):
    """FlowDiffusion.sample_one_video (video_flow_diffusion_model.py:190-216)."""
    fea = generator_compute_fea(gsd, img)
    b = cond.shape[0]
    pred = sample(dsd, fea, cond, (b, 3, num_frames, latent_size, latent_size),
                  sampling_timesteps, timesteps, eta, cond_scale, noise_fn, record, dynamic)
    grid = pred[:, :2]
    if use_residual_flow:
        grid = grid + identity_grid(b, num_frames, latent_size, latent_size)
    conf = (pred[:, 2:3] + 1) * 0.5
    outs, warps = [], []
    for f in range(num_frames):
        g = generator_forward_with_flow(gsd, img, grid[:, :, f].permute(0, 2, 3, 1), conf[:, :, f])
        outs.append(g["prediction"])
        warps.append(g["deformed"])
    return {
        "sample_vid_grid": grid, "sample_vid_conf": conf,
        "sample_out_vid": torch.stack(outs, dim=2), "sample_warped_vid": torch.stack(warps, dim=2),
        "sample_img_fea": fea,
    }


# --------------------------------------------------------------------------------------
# Training rows: frozen LFAE motion predictors + Generator.forward + the DM training step
# (LFAE/modules/{region_predictor,bg_motion_predictor,pixelwise_flow_predictor,util}.py,
#  DM/modules/video_flow_diffusion.py:848-903, DM/modules/video_flow_diffusion_model.py:116-188)
# Restated per frame, exactly in the reference's order of operations (incl. its broadcast matmuls and
# torch.inverse / torch.svd calls), so it is the slow-but-literal counterpart of cvpr23_lfdm_amd/lfae_predictors.py.
# --------------------------------------------------------------------------------------


def make_coordinate_grid(h, w):
    """util.py:51-67."""
    x = 2 * (torch.arange(w).float() / (w - 1)) - 1
    y = 2 * (torch.arange(h).float() / (h - 1)) - 1
    return torch.cat((x.view(1, -1).repeat(h, 1).unsqueeze(2), y.view(-1, 1).repeat(1, w).unsqueeze(2)), 2)


def anti_alias_down(x, weight, scale=0.25):
    """AntiAliasInterpolation2d.forward (util.py:254-264)."""
    ks = weight.shape[-1]
    ka = ks // 2
    kb = ka - 1 if ks % 2 == 0 else ka
    out = F.conv2d(F.pad(x, (ka, kb, ka, kb)), weight=weight, groups=x.shape[1])
    s = int(1 / scale)
    return out[:, :, ::s, ::s]


def hourglass(x, sd, prefix, num_blocks=5, decoder=True):
    """Encoder / Decoder (util.py:153-214): DownBlock2d = conv3x3 -> BN -> ReLU -> AvgPool2 (util.py:115-133),
    UpBlock2d = interpolate x2 -> conv3x3 -> BN -> ReLU (util.py:95-112); decoder concatenates the skips."""
    outs = [x]
    for i in range(num_blocks):
        q = "%sencoder.down_blocks.%d." % (prefix, i)
        y = F.relu(_bn_eval(_conv(outs[-1], sd, q + "conv.", 1), sd, q + "norm."))
        outs.append(F.avg_pool2d(y, 2))
    if not decoder:
        return outs
    out = outs.pop()
    for j in range(num_blocks):
        q = "%sdecoder.up_blocks.%d." % (prefix, j)
        out = F.interpolate(out, scale_factor=2)
        out = F.relu(_bn_eval(_conv(out, sd, q + "conv.", 1), sd, q + "norm."))
        out = torch.cat([out, outs.pop()], dim=1)
    return out


def region_predictor(rsd, x, temperature=0.1, scale=0.25, pad=3, pca_based=True, num_blocks=5):
    """RegionPredictor.forward (region_predictor.py:52-117): pca_based (:109-116, what every LFDM yaml selects), the FOMM-like regression head
    when the state dict has `jacobian.*` (:98-108), or centres + heat-maps only."""
    x = anti_alias_down(x, rsd["down.weight"], scale) if scale != 1 else x
    fmap = hourglass(x, rsd, "predictor.", num_blocks=num_blocks)
    pred = F.conv2d(fmap, rsd["regions.weight"], rsd["regions.bias"], padding=pad)
    shp = pred.shape
    region = F.softmax(pred.view(shp[0], shp[1], -1) / temperature, dim=2).view(*shp)
    grid = make_coordinate_grid(shp[2], shp[3]).unsqueeze(0).unsqueeze(0)
    r = region.unsqueeze(-1)
    mean = (r * grid).sum(dim=(2, 3))
    if not pca_based:
        out = {"shift": mean, "heatmap": region}
        if "jacobian.weight" in rsd:
            jmap = F.conv2d(fmap, rsd["jacobian.weight"], rsd["jacobian.bias"], padding=pad).reshape(shp[0], 1, 4, shp[2], shp[3])
            jac = (region.unsqueeze(2) * jmap).view(shp[0], shp[1], 4, -1).sum(dim=-1).view(shp[0], shp[1], 2, 2)
            out["affine"] = jac
            out["covar"] = torch.matmul(jac, jac.permute(0, 1, 3, 2))
        return out
    mean_sub = grid - mean.unsqueeze(-2).unsqueeze(-2)
    covar = torch.matmul(mean_sub.unsqueeze(-1), mean_sub.unsqueeze(-2)) * r.unsqueeze(-1)
    covar = covar.sum(dim=(2, 3))
    u, s, _ = torch.svd(covar.view(-1, 2, 2))
    affine = torch.matmul(u, torch.diag_embed(s ** 0.5)).view(*covar.shape)
    return {"shift": mean, "covar": covar, "affine": affine, "heatmap": region}


def bg_predictor(bsd, source, driving, bg_type="affine", num_blocks=5):
    """BGMotionPredictor.forward (bg_motion_predictor.py:42-57) for every bg_type of :19: 'zero' (identity, no network), 'shift'
    (fc -> translation column), 'affine' (fc -> upper 2x3), 'perspective' (fc -> upper 2x3 and the first two entries of the last row)."""
    bs = source.shape[0]
    out = torch.eye(3).unsqueeze(0).repeat(bs, 1, 1)
    if bg_type == "zero":
        return out
    feats = hourglass(torch.cat([source, driving], dim=1), bsd, "", num_blocks=num_blocks, decoder=False)
    pred = F.linear(feats[-1].mean(dim=(2, 3)), bsd["fc.weight"], bsd["fc.bias"])
    if bg_type == "shift":
        out[:, :2, 2] = pred
    elif bg_type == "affine":
        out[:, :2, :] = pred.view(-1, 2, 3)
    elif bg_type == "perspective":
        out[:, :2, :] = pred[:, :6].view(bs, 2, 3)
        out[:, 2, :2] = pred[:, 6:].view(bs, 2)
    else:
        raise ValueError(bg_type)
    return out


def region2gaussian(center, covar, h, w):
    """util.py:22-48 (matrix covariance)."""
    grid = make_coordinate_grid(h, w).view(1, 1, h, w, 2)
    d = grid - center.view(*center.shape[:2], 1, 1, 2)
    inv = torch.inverse(covar).view(*covar.shape[:2], 1, 1, 2, 2)
    under = torch.matmul(torch.matmul(d.unsqueeze(-2), inv), d.unsqueeze(-1))
    return torch.exp(-0.5 * under.sum(dim=(-1, -2)))


def pixelwise_flow_predictor(gsd, source_image, driving, source, bg_params, num_regions=10, scale=0.25, use_deformed_source=True,
                             num_blocks=5):
    """PixelwiseFlowPredictor.forward with use_covar_heatmap, use_deformed_source, revert_axis_swap, occlusion
    (pixelwise_flow_predictor.py:48-137)."""
    p = "pixelwise_flow_predictor."
    img = anti_alias_down(source_image, gsd[p + "down.weight"], scale)
    bs, c, h, w = img.shape
    k = num_regions
    heat = region2gaussian(driving["shift"], driving["covar"], h, w) - region2gaussian(source["shift"], source["covar"], h, w)
    heat = torch.cat([torch.zeros(bs, 1, h, w), heat], dim=1).unsqueeze(2)
    ident = make_coordinate_grid(h, w).view(1, 1, h, w, 2)
    cg = ident - driving["shift"].view(bs, k, 1, 1, 2)
    affine = torch.matmul(source["affine"], torch.inverse(driving["affine"]))
    affine = affine * torch.sign(affine[:, :, 0:1, 0:1])
    affine = affine.unsqueeze(-3).unsqueeze(-3).repeat(1, 1, h, w, 1, 1)
    cg = torch.matmul(affine, cg.unsqueeze(-1)).squeeze(-1)
    d2s = cg + source["shift"].view(bs, k, 1, 1, 2)
    bg = ident.repeat(bs, 1, 1, 1, 1)
    bg = torch.cat([bg, torch.ones_like(bg[..., :1])], dim=-1)
    bg = torch.matmul(bg_params.view(bs, 1, 1, 1, 3, 3), bg.unsqueeze(-1)).squeeze(-1)
    bg = bg[..., :2] / bg[..., 2:]
    sparse = torch.cat([bg, d2s], dim=1)
    rep = img.unsqueeze(1).unsqueeze(1).repeat(1, k + 1, 1, 1, 1, 1).view(bs * (k + 1), -1, h, w)
    deformed = F.grid_sample(rep, sparse.view(bs * (k + 1), h, w, -1), align_corners=False).view(bs, k + 1, -1, h, w)
    inp = (torch.cat([heat, deformed], dim=2) if use_deformed_source else heat).reshape(bs, -1, h, w)      # :116-119
    pred = hourglass(inp, gsd, p + "hourglass.", num_blocks)
    mask = F.softmax(F.conv2d(pred, gsd[p + "mask.weight"], gsd[p + "mask.bias"], padding=3), dim=1).unsqueeze(2)
    deformation = (sparse.permute(0, 1, 4, 2, 3) * mask).sum(dim=1).permute(0, 2, 3, 1)
    occ = torch.sigmoid(F.conv2d(pred, gsd[p + "occlusion.weight"], gsd[p + "occlusion.bias"], padding=3))
    return {"optical_flow": deformation, "occlusion_map": occ}


def generator_forward(gsd, img, driving, source, bg_params):
    """Generator.forward (generator.py:90-128)."""
    motion = pixelwise_flow_predictor(gsd, img, driving, source, bg_params)
    out = generator_forward_with_flow(gsd, img, motion["optical_flow"], motion["occlusion_map"])
    out.update(motion)
    out["bottle_neck_feat"] = generator_compute_fea(gsd, img)
    return out


def pseudo_ground_truth(gsd, rsd, bsd, ref_img, real_vid):
    """The no-grad loop of FlowDiffusion.forward (video_flow_diffusion_model.py:124-141), one frame at a time."""
    src = region_predictor(rsd, ref_img)
    grids, confs, outs, warps = [], [], [], []
    for idx in range(real_vid.shape[2]):
        frame = real_vid[:, :, idx]
        drv = region_predictor(rsd, frame)
        bg = bg_predictor(bsd, ref_img, frame)
        g = generator_forward(gsd, ref_img, drv, src, bg)
        grids.append(g["optical_flow"].permute(0, 3, 1, 2))
        confs.append(g["occlusion_map"])
        outs.append(g["prediction"])
        warps.append(g["deformed"])
    return {"real_vid_grid": torch.stack(grids, dim=2), "real_vid_conf": torch.stack(confs, dim=2),
            "real_out_vid": torch.stack(outs, dim=2), "real_warped_vid": torch.stack(warps, dim=2),
            "ref_img_fea": g["bottle_neck_feat"]}


def p_losses(dsd, x_start, t, fea, cond, noise, null_mask=None):
    """GaussianDiffusion.p_losses (video_flow_diffusion.py:856-895), loss_type 'l2', dynamic threshold.
    -> (loss, pred_x0); differentiable w.r.t. every floating entry of dsd that requires grad."""
    b = x_start.shape[0]
    x_noisy = _bc(dsd["sqrt_alphas_cumprod"][t], b) * x_start + _bc(dsd["sqrt_one_minus_alphas_cumprod"][t], b) * noise
    fea_rep = fea.unsqueeze(2).repeat(1, 1, x_start.shape[2], 1, 1)
    pred_noise = unet_forward(dsd, torch.cat([x_noisy, fea_rep], dim=1), t, cond, null_mask)
    loss = F.mse_loss(noise, pred_noise)
    pred_x0 = dynamic_threshold(predict_start_from_noise(dsd, x_noisy, t, pred_noise.detach()))
    return loss, pred_x0
