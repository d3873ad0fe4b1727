// PreNorm LayerNorm + to_qkv (1x1 conv) + SpatialLinearAttention core in three launches that never materialise qkv
// (include/lfdm_hip.h: lfdm_linear_attention_fused_cl_f32), for C = 64 (the finest UNet level):
// DM/modules/video_flow_diffusion.py:170-189 (LayerNorm / PreNorm) + :249-263 (to_qkv, softmaxes, context, out).
//
// Separate kernels spent 79 us writing the 126 MB qkv tensor and 81 us reading it twice.  Here every kernel starts from
// the 10.5 MB input x and recomputes the projection it needs on the MFMA units, using accumulator layouts AS operand
// layouts (v_mfma_f32_32x32x2: D has lane = column, registers = rows (r&3) + 8*(r>>2) + 4*half):
//  A. context partials, one wavefront per (frame, head, 128-token split): K = xhat Wk^T and V = xhat Wv^T are normal
//     GEMMs (rows = tokens, cols = features); their accumulators are exactly the A / B operands of
//     ctx[d][e] += sum_n exp(K[n][d] - m[d]) V[n][e] for the token order t(half, s) - no layout change.  The k-softmax is
//     over ALL tokens of a frame, so each split keeps its own max m_p and sum s_p (merged in B).
//  B. merge of the splits (a few KB per frame/head): ctx = sum_p e^{m_p - M} ctx_p / sum_p e^{m_p - M} s_p.
//  C. output, one wavefront per (frame, 32-token tile), all heads: Q^T = Wq xhat^T is computed TRANSPOSED so that its
//     accumulator (lane = token, registers = 16 features) is the A operand of out = q~ ctx; softmax over the 32
//     features = register reduction + one shuffle.
// LayerNorm: a lane holds half of its token's channels; mean / variance = local sums + one shuffle; gamma is folded
// into the weights by the host.
#include <stdlib.h>

#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

constexpr int HEADS = 8;
constexpr int DH = 32;
constexpr int OUT_LD = HEADS * DH;      // 256
constexpr int C = 64;                   // input channels (finest level)
constexpr int CH = C / 2;               // channels per k-slot
constexpr int SPLIT_TOK = 32;           // smallest context split (one token tile): sizes the workspace
constexpr int PART = DH * DH + 2 * DH;  // floats per partial: ctx[32][32] | m[32] | s[32]
constexpr float LA_SCALE = 0.17677669529663687f;

// xhat fragment of one token: lane (token l31 of the tile, half kh) holds channels CH*kh .. CH*kh + CH-1, normalised.  `xrow` is ALWAYS a valid
// row (the callers clamp the token index) and is loaded unconditionally; !ok only zeroes the result.  (Round 6: behind `if (ok)` every one of the
// eight 16-byte loads got a branch of its own and, in the context pass's tile loop, an s_waitcnt vmcnt(0) right behind it - eight memory round
// trips in a row per 32-token tile where one was meant.)
__device__ __forceinline__ void load_xraw(const float* __restrict__ xrow, int kh, float4 (&raw)[CH / 4]) {
#pragma unroll
  for (int q = 0; q < CH / 4; ++q) raw[q] = *reinterpret_cast<const float4*>(xrow + CH * kh + 4 * q);
}
__device__ __forceinline__ void xhat_stats(const float4 (&raw)[CH / 4], float eps, float& mean, float& rstd) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int q = 0; q < CH / 4; ++q) {
    const float4 v = raw[q];
    s1 += (v.x + v.y) + (v.z + v.w);
    s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  s1 += __shfl_xor(s1, 32);
  s2 += __shfl_xor(s2, 32);
  mean = s1 * (1.0f / (float)C);
  float var = s2 * (1.0f / (float)C) - mean * mean;
  if (var < 0.f) var = 0.f;
  rstd = 1.0f / sqrtf(var + eps);
}
__device__ __forceinline__ void xhat_apply(const float4 (&raw)[CH / 4], bool ok, float mean, float rstd, float (&xf)[CH]) {
#pragma unroll
  for (int q = 0; q < CH / 4; ++q) {
    const float4 v = raw[q];
    xf[4 * q] = ok ? (v.x - mean) * rstd : 0.f;
    xf[4 * q + 1] = ok ? (v.y - mean) * rstd : 0.f;
    xf[4 * q + 2] = ok ? (v.z - mean) * rstd : 0.f;
    xf[4 * q + 3] = ok ? (v.w - mean) * rstd : 0.f;
  }
}
__device__ __forceinline__ void load_xhat(const float* __restrict__ xrow, bool ok, int kh, float eps, float (&xf)[CH]) {
  float4 raw[CH / 4];
  float mean, rstd;
  load_xraw(xrow, kh, raw);
  xhat_stats(raw, eps, mean, rstd);
  xhat_apply(raw, ok, mean, rstd, xf);
}

// Weight fragment of one (q | k | v, head): lane (feature l31, k-slot kh) holds W[which*256 + head*32 + l31][CH*kh .. CH*kh + CH-1].
// The weights arrive PACKED in that order (round 4, ops.pack_linattn_weights): [3][8 heads][8 quads][64 lanes][4], so each of the eight load
// instructions reads one contiguous 1 KB - in the row-major layout every instruction touched a 16-byte piece of 64 different 128-byte lines.
__device__ __forceinline__ void load_wfrag(const float* __restrict__ wp, int which, int head, int lane, float (&wf)[CH]) {
  const float* src = wp + ((int64_t)(which * HEADS + head) * (CH / 4) * 64 + lane) * 4;
#pragma unroll
  for (int q = 0; q < CH / 4; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(src + 256 * q);
    wf[4 * q] = v.x; wf[4 * q + 1] = v.y; wf[4 * q + 2] = v.z; wf[4 * q + 3] = v.w;
  }
}

// ---- A: grid (n_frames*8, nsplit), 64 threads ----
// Round 4: a wave walks its split's 32-token tiles ONE AT A TIME with a running k-softmax (max m, sum s, context rescaled by e^{m_old - m_new}
// per feature row), so the split length no longer costs registers: the launcher sizes the splits for ONE round of waves on the chip (the
// fixed 64-token splits gave 5120 waves for 3072 slots at 32x32 - a second, two-thirds-empty round - and 16 partials per (frame, head)
// for the merge pass).  The rescale factor lives per LANE (feature d = l31) while the context rows live per REGISTER: 16 shuffles per tile.
// Round 6, measured and NOT kept: the units (frame, head, tile) as ONE linear sequence cut into equal pieces per wave - exactly ten units per
// SIMD at 32x32 x 40 frames instead of 2.5 waves per SIMD, pieces crossing (frame, head) boundaries: +1.6 ms per video at five units per wave,
// equal at four (profiles/r06_m_linattn_ctx_balanced_ab.txt) - the pass does not wait for its busiest SIMD.
__global__ __launch_bounds__(64, 3) void linattn_fused_ctx_kernel(const float* __restrict__ x, int ldx,
                                                                  const float* __restrict__ wqkv, int hw, float eps,
                                                                  float* __restrict__ part, int split_tok) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
  const int f = blockIdx.x >> 3, h = blockIdx.x & 7;
  const int split = blockIdx.y, nsplit = gridDim.y;
  const int n0 = split * split_tok;
  const int n1 = (n0 + split_tok < hw) ? n0 + split_tok : hw;

  float wk[CH], wv[CH];
  load_wfrag(wqkv, 1, h, lane, wk);                        // B operands: lane = feature l31, k-slot = channel half
  load_wfrag(wqkv, 2, h, lane, wv);

  float m_run = -3.0e38f, s_run = 0.f;                     // per lane = feature d = l31 (both halves hold the same value)
  f32x16 ctx;
#pragma unroll
  for (int r = 0; r < 16; ++r) ctx[r] = 0.f;
  // (Round 6, tried: the NEXT tile's rows requested before this tile's products.  hipcc sinks such loads to their first use behind the back-edge; pinned
  //  there by a use behind the products, the 32 extra live registers push the kernel over its 168 and the loads end up behind the last MFMAs anyway.
  //  What is kept: the tile's eight loads are unconditional and in flight TOGETHER - one round trip per tile instead of six to eight.)
  for (int t0 = n0; t0 < n1; t0 += 32) {
    const int n = t0 + l31;
    float xf[CH];
    load_xhat(x + ((int64_t)f * hw + (n < n1 ? n : n1 - 1)) * ldx, n < n1, kh, eps, xf);
    f32x16 ka, va;
#pragma unroll
    for (int r = 0; r < 16; ++r) ka[r] = va[r] = 0.f;
#pragma unroll
    for (int s = 0; s < CH; ++s) {
      ka = mfma_32x32x2(xf[s], wk[s], ka);                 // rows = tokens, cols = features d
      va = mfma_32x32x2(xf[s], wv[s], va);                 // rows = tokens, cols = features e
    }
    // lane: feature l31; register r = token t0 + (r&3) + 8*(r>>2) + 4*kh
    float m = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tok = t0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (tok < n1) m = fmaxf(m, ka[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    const float m_new = fmaxf(m_run, m);
    const float alpha = expf(m_run - m_new);               // 0 for the first tile (m_run = -3e38)
    float ssum = 0.f;
    float e[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tok = t0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      e[r] = tok < n1 ? expf(ka[r] - m_new) : 0.f;
      ssum += e[r];
    }
    ssum += __shfl_xor(ssum, 32);
    s_run = s_run * alpha + ssum;
    m_run = m_new;
    // context rows d = (r&3) + 8*(r>>2) + 4*kh are scaled by the factor of feature d, which lane d holds
#pragma unroll
    for (int r = 0; r < 16; ++r) ctx[r] *= __shfl(alpha, (r & 3) + 8 * (r >> 2) + 4 * kh);
#pragma unroll
    for (int r = 0; r < 16; ++r) ctx = mfma_32x32x2(e[r], va[r], ctx);      // A: lane = d, k-slot half = token; B: lane = e, same token
  }
  float* dst = part + ((int64_t)blockIdx.x * nsplit + split) * PART;
#pragma unroll
  for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2) + 4 * kh) * DH + l31] = ctx[r];   // ctx[d][e], lane = e
  if (kh == 0) {
    dst[DH * DH + l31] = m_run;
    dst[DH * DH + DH + l31] = s_run;
  }
}

// ---- B: grid (n_frames*8), 256 threads: ctx[d][e] merged over the splits ----
__global__ __launch_bounds__(256) void linattn_fused_merge_kernel(const float* __restrict__ part, int nsplit,
                                                                  float* __restrict__ ctx_out) {
  constexpr int MAXS = 64;
  __shared__ float s_w[MAXS][DH];                  // weight of split p for feature d: e^{m_p - M} / denominator
  const int tid = threadIdx.x;
  const float* base = part + (int64_t)blockIdx.x * nsplit * PART;
  // (round 4: every loop over the splits is unrolled by four - a load per trip paid one L2 round trip per split, 16 of them in a row at
  //  32x32: 12.6 us for a kernel that moves 22 MB)
  if (nsplit <= 8) {
    // (round 6) the sampler's shape - eight splits or fewer: everything the thread will read is requested at once - its four context elements of every
    // split and (threads < 32) the splits' maxima and sums - instead of in three dependent phases (maxima -> weights | barrier | weighted sum)
    constexpr int PS = 8;
    float cv[4][PS], pm[PS], psum[PS];
#pragma unroll
    for (int p = 0; p < PS; ++p) {
      const int pc = p < nsplit ? p : nsplit - 1;
#pragma unroll
      for (int e = 0; e < 4; ++e) cv[e][p] = base[(int64_t)pc * PART + tid + 256 * e];
      pm[p] = base[(int64_t)pc * PART + DH * DH + (tid & (DH - 1))];
      psum[p] = base[(int64_t)pc * PART + DH * DH + DH + (tid & (DH - 1))];
    }
    if (tid < DH) {
      float mm = -3.0e38f;
#pragma unroll
      for (int p = 0; p < PS; ++p)
        if (p < nsplit) mm = fmaxf(mm, pm[p]);
      float den = 0.f, w[PS];
#pragma unroll
      for (int p = 0; p < PS; ++p) {
        w[p] = p < nsplit ? expf(pm[p] - mm) : 0.f;
        den += w[p] * psum[p];
      }
      const float inv = 1.0f / den;
#pragma unroll
      for (int p = 0; p < PS; ++p)
        if (p < nsplit) s_w[p][tid] = w[p] * inv;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = tid + 256 * e, d = i >> 5;
      float acc = 0.f;
#pragma unroll
      for (int p = 0; p < PS; ++p)
        if (p < nsplit) acc += s_w[p][d] * cv[e][p];
      ctx_out[(int64_t)blockIdx.x * DH * DH + i] = acc;
    }
    return;
  }
  if (tid < DH) {
    float mm = -3.0e38f;
#pragma unroll 4
    for (int p = 0; p < nsplit; ++p) mm = fmaxf(mm, base[(int64_t)p * PART + DH * DH + tid]);
    float den = 0.f;
#pragma unroll 4
    for (int p = 0; p < nsplit; ++p) {
      const float w = expf(base[(int64_t)p * PART + DH * DH + tid] - mm);
      s_w[p][tid] = w;
      den += w * base[(int64_t)p * PART + DH * DH + DH + tid];
    }
    const float inv = 1.0f / den;
    for (int p = 0; p < nsplit; ++p) s_w[p][tid] *= inv;
  }
  __syncthreads();
  for (int i = tid; i < DH * DH; i += 256) {
    const int d = i >> 5;
    float acc = 0.f;
#pragma unroll 4
    for (int p = 0; p < nsplit; ++p) acc += s_w[p][d] * base[(int64_t)p * PART + i];
    ctx_out[(int64_t)blockIdx.x * DH * DH + i] = acc;
  }
}

// ---- C: grid (ceil(hw/32), n_frames, head groups), 64 threads ----
__global__ __launch_bounds__(64, 2) void linattn_fused_out_kernel(const float* __restrict__ x, int ldx,
                                                                  const float* __restrict__ wqkv,
                                                                  const float* __restrict__ ctx, int hw, float eps,
                                                                  float* __restrict__ out) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
  const int f = blockIdx.y;
  const int n = blockIdx.x * 32 + l31;
  float xf[CH];                                            // B operand of Q^T = Wq xhat^T: lane = token
  load_xhat(x + ((int64_t)f * hw + (n < hw ? n : 0)) * ldx, n < hw, kh, eps, xf);
  // grid.z splits the heads (the token tile's 8 KB of x are simply read again): more wavefronts per SIMD to hide the
  // weight / context fragment latency of the serial per-head chain
  const int hpb = HEADS / gridDim.z;
#pragma unroll 1
  for (int h = blockIdx.z * hpb; h < (blockIdx.z + 1) * hpb; ++h) {
    float wq[CH];
    load_wfrag(wqkv, 0, h, lane, wq);                      // A operand: lane = feature d
    f32x16 q;
#pragma unroll
    for (int r = 0; r < 16; ++r) q[r] = 0.f;
#pragma unroll
    for (int s = 0; s < CH; ++s) q = mfma_32x32x2(wq[s], xf[s], q);
    // lane = token; q[r] = feature d(kh, r) = (r&3) + 8*(r>>2) + 4*kh : softmax over the 32 features
    float m = q[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, q[r]);
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      q[r] = expf(q[r] - m);
      sum += q[r];
    }
    sum += __shfl_xor(sum, 32);
    const float inv = LA_SCALE / sum;
    // out[token][e] = sum_d q~[token][d] ctx[d][e]: A = q (lane = token, slot = d half), B = ctx[d(kh, r)][e = l31]
    const float* cb = ctx + ((int64_t)f * HEADS + h) * DH * DH + l31;
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) o = mfma_32x32x2(q[r] * inv, cb[((r & 3) + 8 * (r >> 2) + 4 * kh) * DH], o);
    // lane = feature e; o[r] = token (r&3) + 8*(r>>2) + 4*kh of the tile
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tok = blockIdx.x * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (tok < hw) out[((int64_t)f * hw + tok) * OUT_LD + h * DH + l31] = o[r];
    }
  }
}


// ---- C': output + to_out + bias + residual in one pass (round 6; lfdm_linear_attention_fused_out_cl_f32) ----
// The separate to_out projection of a C = 64 block was a 25 us launch that re-read the 42 MB attention output pass C had just written in
// 128-byte pieces.  Here a workgroup of four waves owns one 32-token tile of a frame; wave w runs heads 2w, 2w + 1:
//   Q^T = Wq xhat^T as in pass C (lane = token, registers = 16 features), softmax over the features,
//   O^T[e][tok] = sum_d ctx[d][e] q~[tok][d]  - pass C's product with the operands exchanged: the accumulator then has lane = token, registers =
//     features e(half, r), which IS the B operand (column = token, k = e) of
//   Y^T[c][tok] += sum_e Wout[c][32 h + e] O^T[e][tok]   (A = Wout rows in operand order, two 32-channel row blocks).
// The four waves' partial Y^T (their two heads each) meet in LDS ([wave][token][64 + 4] floats: a lane's registers (r & 3) are four consecutive
// channels = one 16-byte store), and 256 threads finish out[tok][c] = x[tok][c] + bias[c] + sum of the four partials as float4 rows.
// wout arrives packed (ops.pack_linattn_out_weight): [8 heads][2 row blocks][4 quads][64 lanes = 32 kh + c_local][4] <- Wout[32 cb + c_local][32 h + 8 quad + 4 kh + e].
constexpr int LDY = C + 4;
__global__ __launch_bounds__(256, 2) void linattn_fused_out2_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ wqkv,
                                                                    const float* __restrict__ wout, const float* __restrict__ bias_out,
                                                                    const float* __restrict__ ctx, int hw, float eps, float* __restrict__ out,
                                                                    int ldo) {
  __shared__ __attribute__((aligned(16))) float s_y[4 * 32 * LDY];
  const int tid = threadIdx.x;
  const int wave = lfdm_uniform(tid >> 6);
  const int lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
  const int f = blockIdx.y;
  const int n0 = blockIdx.x * 32;
  const int n = n0 + l31;
  float xf[CH];                                            // B operand of Q^T = Wq xhat^T: lane = token
  load_xhat(x + ((int64_t)f * hw + (n < hw ? n : 0)) * ldx, n < hw, kh, eps, xf);
  f32x16 yT[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) yT[cb][r] = 0.f;
#pragma unroll 1
  for (int h = 2 * wave; h < 2 * wave + 2; ++h) {
    float wq[CH];
    load_wfrag(wqkv, 0, h, lane, wq);                      // A operand: lane = feature d
    f32x16 q;
#pragma unroll
    for (int r = 0; r < 16; ++r) q[r] = 0.f;
#pragma unroll
    for (int s = 0; s < CH; ++s) q = mfma_32x32x2(wq[s], xf[s], q);
    // lane = token; q[r] = feature d(kh, r) = (r&3) + 8*(r>>2) + 4*kh : softmax over the 32 features
    float m = q[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, q[r]);
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      q[r] = expf(q[r] - m);
      sum += q[r];
    }
    sum += __shfl_xor(sum, 32);
    const float inv = LA_SCALE / sum;
    // O^T[e][tok]: A = ctx[d(kh, r)][e = l31] (lane = row e), B = q~ (lane = column tok, k = d(kh, r))
    const float* cb_ = ctx + ((int64_t)f * HEADS + h) * DH * DH + l31;
    f32x16 oT;
#pragma unroll
    for (int r = 0; r < 16; ++r) oT[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) oT = mfma_32x32x2(cb_[((r & 3) + 8 * (r >> 2) + 4 * kh) * DH], q[r] * inv, oT);
    // lane = token; oT[r] = O^T[e(kh, r)][tok].  Y^T[c][tok] += Wout[c][32 h + e] O^T[e][tok]: A = Wout fragment (lane = row c), B = oT
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const float* wsrc = wout + ((int64_t)((h * 2 + cb) * 4) * 64 + lane) * 4;
      float4 wo[4];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) wo[qd] = *reinterpret_cast<const float4*>(wsrc + 256 * qd);
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        yT[cb] = mfma_32x32x2(wo[qd].x, oT[4 * qd + 0], yT[cb]);
        yT[cb] = mfma_32x32x2(wo[qd].y, oT[4 * qd + 1], yT[cb]);
        yT[cb] = mfma_32x32x2(wo[qd].z, oT[4 * qd + 2], yT[cb]);
        yT[cb] = mfma_32x32x2(wo[qd].w, oT[4 * qd + 3], yT[cb]);
      }
    }
  }
  // lane (tok = l31, half kh): yT[cb][r] = Y^T[c = 32 cb + (r&3) + 8 (r>>2) + 4 kh][tok] over this wave's two heads
  float* ys = s_y + (wave * 32 + l31) * LDY;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(ys + 32 * cb + 8 * g + 4 * kh) = make_float4(yT[cb][4 * g], yT[cb][4 * g + 1], yT[cb][4 * g + 2], yT[cb][4 * g + 3]);
  __syncthreads();
  const int tok = tid >> 3, c8 = (tid & 7) * 8;
  if (n0 + tok < hw) {
    const int64_t row = (int64_t)f * hw + n0 + tok;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = c8 + 4 * j;
      float4 v = *reinterpret_cast<const float4*>(x + row * ldx + c);
      const float4 b4 = bias_out ? *reinterpret_cast<const float4*>(bias_out + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        const float4 u = *reinterpret_cast<const float4*>(s_y + (w4 * 32 + tok) * LDY + c);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      *reinterpret_cast<float4*>(out + row * ldo + c) = v;
    }
  }
}

}  // namespace

extern "C" size_t lfdm_linear_attention_fused_ws_bytes(int n_frames, int hw) {
  const int nsplit = (hw + SPLIT_TOK - 1) / SPLIT_TOK;
  return ((size_t)n_frames * HEADS * nsplit * PART + (size_t)n_frames * HEADS * DH * DH) * sizeof(float);
}

namespace {
// passes A + B: context partials and their merge; *ctx_out = the merged contexts inside the workspace
void launch_ctx_and_merge(const float* x, int ldx, const float* wqkv, int n_frames, int hw, float ln_eps, void* ws, float** ctx_out,
                          hipStream_t stream) {
  // splits of whole 32-token tiles, as many as fit ONE round of waves (3 per SIMD: 3072), at most 64 (the merge kernel's table)
  const int tiles = (hw + 31) / 32;
  // (batched shapes - more (frame, head) pairs than half a round - get at least four rounds of waves instead of one long wave per pair:
  //  a single split left 5120 waves of 32 serial tiles for 3072 slots at B = 16, profiles/r04_t_other_configs.json)
  const int64_t pairs = (int64_t)n_frames * HEADS;
  int want = (int)((pairs <= 1536 ? 3072 : 4 * 3072 + pairs - 1) / pairs);
  if (const char* e = lfdm_knob("LFDM_LINATTN_SPLITS")) want = atoi(e);      // experiment knob
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  if (want > tiles) want = tiles;
  const int split_tok = ((tiles + want - 1) / want) * 32;
  const int nsplit = (hw + split_tok - 1) / split_tok;
  float* part = (float*)ws;
  float* ctx = part + (size_t)n_frames * HEADS * ((hw + SPLIT_TOK - 1) / SPLIT_TOK) * PART;
  LFDM_LAUNCH(linattn_fused_ctx_kernel, dim3(n_frames * HEADS, nsplit), dim3(64), 0, stream, x, ldx, wqkv, hw, ln_eps, part, split_tok);
  LFDM_LAUNCH(linattn_fused_merge_kernel, dim3(n_frames * HEADS), dim3(256), 0, stream, (const float*)part, nsplit, ctx);
  *ctx_out = ctx;
}
}  // namespace

extern "C" int lfdm_linear_attention_fused_cl_f32(const float* x, int ldx, int channels, const float* wqkv, float* out,
                                                  int n_frames, int hw, float ln_eps, void* ws, size_t ws_bytes,
                                                  lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !wqkv || !out || n_frames <= 0 || hw <= 0 || hw > 64 * 64 || channels != C || ldx < C || ldx % 4 != 0 ||
      (((uintptr_t)x | (uintptr_t)wqkv) & 15) || (int64_t)n_frames * HEADS > 0x7fffffff) {
    lfdm_set_error("linear_attention_fused: needs C == 64 and 16-byte aligned rows");
    return LFDM_EINVAL;
  }
  if (!ws || ws_bytes < lfdm_linear_attention_fused_ws_bytes(n_frames, hw)) {
    lfdm_set_error("linear_attention_fused: workspace too small");
    return LFDM_EWORKSPACE;
  }
  float* ctx = nullptr;
  launch_ctx_and_merge(x, ldx, wqkv, n_frames, hw, ln_eps, ws, &ctx, stream);
  const int64_t otiles = (int64_t)((hw + 31) / 32) * n_frames;
  const int hgroups = otiles >= 4096 ? 1 : (otiles >= 2048 ? 2 : 4);
  LFDM_LAUNCH(linattn_fused_out_kernel, dim3((hw + 31) / 32, n_frames, hgroups), dim3(64), 0, stream, x, ldx, wqkv,
              (const float*)ctx, hw, ln_eps, out);
  return lfdm_check_launch("linear_attention_fused");
}

extern "C" int lfdm_linear_attention_fused_out_cl_f32(const float* x, int ldx, int channels, const float* wqkv, const float* wout,
                                                      const float* bias_out, float* out, int ldo, int n_frames, int hw, float ln_eps,
                                                      void* ws, size_t ws_bytes, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !wqkv || !wout || !out || out == x || n_frames <= 0 || hw <= 0 || hw > 64 * 64 || channels != C || ldx < C || ldx % 4 != 0 || ldo < C ||
      ldo % 4 != 0 || (((uintptr_t)x | (uintptr_t)wqkv | (uintptr_t)wout | (uintptr_t)out | (uintptr_t)bias_out) & 15) ||
      (int64_t)n_frames * HEADS > 0x7fffffff || n_frames > 65535) {
    lfdm_set_error("linear_attention_fused_out: needs C == 64, 16-byte aligned rows, out != x, <= 65535 frames");
    return LFDM_EINVAL;
  }
  if (!ws || ws_bytes < lfdm_linear_attention_fused_ws_bytes(n_frames, hw)) {
    lfdm_set_error("linear_attention_fused_out: workspace too small");
    return LFDM_EWORKSPACE;
  }
  float* ctx = nullptr;
  launch_ctx_and_merge(x, ldx, wqkv, n_frames, hw, ln_eps, ws, &ctx, stream);
  LFDM_LAUNCH(linattn_fused_out2_kernel, dim3((hw + 31) / 32, n_frames), dim3(256), 0, stream, x, ldx, wqkv, wout, bias_out, (const float*)ctx, hw,
              ln_eps, out, ldo);
  return lfdm_check_launch("linear_attention_fused_out");
}
