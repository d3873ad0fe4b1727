// PreNorm LayerNorm + to_qkv + temporal attention in ONE kernel (include/lfdm_hip.h:
// lfdm_temporal_attention_fused_cl_f32) for the two finest UNet levels (C = 64, 128):
// Residual(PreNorm(EinopsToAndFrom(Attention))) without its to_out projection,
// DM/modules/video_flow_diffusion.py:170-189,270-283,303-363.
//
// Why: at C = 64 the separate kernels are bound by the 126 MB qkv tensor - the projection takes 66 us to WRITE
// it (the GEMM itself is 26 us of MFMA) and the attention kernel 70 us to read it back.  Here qkv never exists.
// One wavefront owns one pixel's sequence (T <= 64 tokens) and walks over the 8 heads:
//  * the sequence's rows of x (T x C) are loaded ONCE, straight into MFMA operand registers: lane (token l&15,
//    k-slot l>>4) holds the C/4 consecutive channels (C/4)*slot .. of its token = float4 loads; the channel
//    LayerNorm statistics are two shuffles (the four slots of a token) and the rows are normalised in registers
//    (gamma is folded into the weights by the host);
//  * Q^T and K^T are computed TRANSPOSED (features x tokens = W_h . x^T): the accumulator layout of
//    v_mfma_f32_16x16x4 then has lane = token, registers = features 16*f + 4*slot + r, which IS a legal
//    operand layout for S = Q K^T (any feature order works if Q and K agree) - no layout change, scale and
//    rotary are applied on the accumulators (a rotation pair stays inside a lane);
//  * V = x W_v^T is computed untransposed: lane = feature, registers = tokens 16*t + 4*slot + r, exactly the
//    B operand of P V with the token order t(slot, s) = 16*(s>>2) + 4*slot + (s&3);
//  * the scores are computed TRANSPOSED too (S^T = K Q^T): lane = query token, registers = key tokens, so the softmax
//    over the keys is a register reduction + two shuffles and P^T is already the A operand of P V: no LDS at all.
// Weight fragments (rows of W, contiguous over channels) come from L2 as float4.
#include <stdlib.h>

#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

constexpr int HEADS = 8;
constexpr int DH = 32;
constexpr int OUT_LD = HEADS * DH;      // 256

#if defined(LFDM_EMU_BUILD)
static inline float fast_exp(float x) { return expf(x); }
#else
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }      // v_exp_f32(x * log2 e)
#endif

// Attention of one (sequence, head) from operand-layout fragments (see the header comment): scores transposed,
// softmax in registers, P V, store.  qf/kf[ti][4*fi + r] = feature 16*fi + 4*lq + r of token 16*ti + l15 (q scaled and
// rotated, k rotated); vf[half][4*ti + r] = v[token 16*ti + 4*lq + r][16*half + l15].
// o_lds != nullptr: the attention output goes to the workgroup's LDS tile [token][256] (16-byte quads XOR-swizzled by the token, see
// temporal_attn_fused_out_kernel) instead of global memory.
template <int NT>
__device__ __forceinline__ void attend_store(const float (&qf)[NT][8], const float (&kf)[NT][8], const float (&vf)[2][4 * NT],
                                             int head, int L, int l15, int lq, const float* __restrict__ bias, bool bias_vec,
                                             float* __restrict__ out, int64_t row0, int hw, float* o_lds = nullptr) {
  // ---- S^T = K Q^T: lane = query token 16*ti + l15, registers = key tokens 16*tj + 4*lq + r.  In this orientation the
  // softmax over the keys of a query is a reduction over the lane's registers plus two shuffles (the four k-slots),
  // and the result is ALREADY the A operand of P V for the token order t(lq, s) = 16*(s>>2) + 4*lq + (s&3): no LDS. ----
  f32x4 st[NT][NT];                                   // [ti (query tile)][tj (key tile)]
#pragma unroll
  for (int ti = 0; ti < NT; ++ti)
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) acc = mfma_16x16x4(kf[tj][s], qf[ti][s], acc);
      st[ti][tj] = acc;
    }
  // VALU diet (round 4; the head loop spends as many issue cycles outside the matrix pipe as inside it: 39 IEEE divisions, 36 expf
  // expansions per head): exp is the hardware exponential (v_exp_f32 on x * log2 e: the arguments are <= 0, the
  // result within 2 ulp of expf), and the normalisation multiplies by ONE reciprocal per query instead of dividing every probability.
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    const int qt = ti * 16 + l15;                     // this lane's query token
    float m = -3.0e38f;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      const int key0 = tj * 16 + lq * 4;
      if (bias && qt < L && key0 < L) {                 // (guarded: unconditional loads of all nine fragments spill 84 registers)
        const float* bp = bias + ((int64_t)head * L + qt) * L + key0;
        if (bias_vec) {                                 // four consecutive keys of one query row: one 16-byte load
          const float4 b4 = *reinterpret_cast<const float4*>(bp);
          bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[r] = (key0 + r < L) ? bp[r] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = (key0 + r >= L) ? -3.0e38f : st[ti][tj][r] + bv[r];
        st[ti][tj][r] = v;
        m = fmaxf(m, v);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = tj * 16 + lq * 4 + r;
        const float e = key < L ? fast_exp(st[ti][tj][r] - m) : 0.f;
        st[ti][tj][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[ti][tj][r] = st[ti][tj][r] * inv;
  }

  // ---- O = P V ----
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    f32x4 o[2];
    o[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    o[1] = o[0];
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[0] = mfma_16x16x4(st[ti][tj][r], vf[0][4 * tj + r], o[0]);
        o[1] = mfma_16x16x4(st[ti][tj][r], vf[1][4 * tj + r], o[1]);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = ti * 16 + lq * 4 + r;
      if (t < L) {
        if (o_lds) {      // column c = head*32 + 16*half + l15 -> quad c >> 2, swizzled by the row
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int c = head * DH + 16 * half + l15;
            o_lds[t * OUT_LD + ((((c >> 2) ^ (t & 15)) << 2) | (c & 3))] = o[half][r];
          }
        } else {
          float* dst = out + (row0 + (int64_t)t * hw) * OUT_LD + head * DH;
          dst[l15] = o[0][r];
          dst[16 + l15] = o[1][r];
        }
      }
    }
  }
}

// The body of one wavefront: heads [head_begin, head_begin + heads_per_block) of sequence `seq`; results to global `out` or to `o_lds`.
// ROT_RELOAD: the rotary factors are re-read from their (L1-resident, 2.5 KB) tables per head instead of living in 24 registers across the
// head loop - the one-launch block needs the registers (27 spills otherwise).
// WPACK: wqkv is in MFMA-operand order [q|k|v][head][feature half][quad q][lane][4] (lane (l15, lq) <- W[row 16*half + l15][C/4*lq + 4q ..]):
// every fragment load instruction then reads ONE contiguous 1 KB instead of a 16-byte piece of each of its 64 lanes' 64-byte segments
// (four times the L1 line look-ups per byte in the row-major layout).
template <int LP, int C, bool ROT_RELOAD = false, bool WPACK = false>
__device__ __forceinline__ void tattn_heads(const float* __restrict__ x, int ldx, int head_begin, int heads_per_block,
                                            const float* __restrict__ wqkv, float* __restrict__ out, int frames, int hw, int64_t seq,
                                            const float* __restrict__ bias, const float* __restrict__ rot_cos,
                                            const float* __restrict__ rot_sin, float eps, float* o_lds) {
  constexpr int NT = LP / 16;
  constexpr int CQ = C / 4;                        // channels per k-slot

  const int lane = threadIdx.x & 63;
  const int l15 = lane & 15, lq = lane >> 4;
  const int L = frames;
  const int64_t b = seq / hw, pix = seq - b * hw;
  const int64_t row0 = b * frames * hw + pix;
  const float scale = 0.17677669529663687f;  // 32^-0.5

  // ---- the sequence's rows, normalised, as MFMA fragments: xf[ti][s] = xhat[token 16*ti + l15][CQ*lq + s] ----
  float xf[NT][CQ];
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    const int t = ti * 16 + l15;
    float s1 = 0.f, s2 = 0.f;
    if (t < L) {
      const float* src = x + (row0 + (int64_t)t * hw) * ldx + CQ * lq;
#pragma unroll
      for (int q = 0; q < CQ / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(src + 4 * q);
        xf[ti][4 * q] = v.x; xf[ti][4 * q + 1] = v.y; xf[ti][4 * q + 2] = v.z; xf[ti][4 * q + 3] = v.w;
        s1 += (v.x + v.y) + (v.z + v.w);
        s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
    } else {
#pragma unroll
      for (int s = 0; s < CQ; ++s) xf[ti][s] = 0.f;
    }
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    const float mean = s1 * (1.0f / (float)C);
    float var = s2 * (1.0f / (float)C) - mean * mean;
    if (var < 0.f) var = 0.f;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (t < L) {
#pragma unroll
      for (int s = 0; s < CQ; ++s) xf[ti][s] = (xf[ti][s] - mean) * rstd;
    }
  }

  // rotary factors of this lane's tokens / feature pairs (head independent)
  float rc[NT][4], rs[NT][4];
  if (rot_cos && !ROT_RELOAD) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
      const int t = ti * 16 + l15;
      const int tt = t < L ? t : 0;
#pragma unroll
      for (int fi = 0; fi < 2; ++fi)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          rc[ti][2 * fi + pr] = rot_cos[tt * 16 + 8 * fi + 2 * lq + pr];
          rs[ti][2 * fi + pr] = rot_sin[tt * 16 + 8 * fi + 2 * lq + pr];
        }
    }
  }
  const bool bias_vec = bias && (L % 4 == 0) && ((((uintptr_t)bias) & 15) == 0);

#pragma unroll 1
  for (int head = head_begin; head < head_begin + heads_per_block; ++head) {
    // ---- Q^T, K^T: rows = features (2 tiles of 16), cols = tokens (NT tiles) ----
    float qf[NT][8], kf[NT][8];
    {
      // four weight-row fragments (q/k x two feature tiles), all loads issued before the first MFMA; the NT token
      // tiles of a fragment are independent accumulator chains and are interleaved step by step.
      // (Round 6, measured and NOT kept: the head's 24 fragment loads as a hand-counted asm pipeline - 16 up front, the v fragments under
      //  the k projections, rotary moved behind the v projection: +1.9 ms per video, profiles/r06_n_tattn_asm_pipeline_ab.txt.  The ISA
      //  hipcc emits here already runs the loads one to two quads (12-24 MFMAs) ahead of their use; the weight stream is not what
      //  the launch waits for.)
      float wa[4][CQ];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float* wsrc = WPACK ? wqkv + ((int64_t)(((g >> 1) * HEADS + head) * 2 + (g & 1)) * (CQ / 4) * 64 + lane) * 4
                                  : wqkv + ((int64_t)((g >> 1) * OUT_LD + head * DH + 16 * (g & 1) + l15)) * C + CQ * lq;
#pragma unroll
        for (int q = 0; q < CQ / 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(wsrc + (WPACK ? 256 * q : 4 * q));
          wa[g][4 * q] = v.x; wa[g][4 * q + 1] = v.y; wa[g][4 * q + 2] = v.z; wa[g][4 * q + 3] = v.w;
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 acc[NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) acc[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < CQ; ++s)
#pragma unroll
          for (int ti = 0; ti < NT; ++ti) acc[ti] = mfma_16x16x4(wa[g][s], xf[ti][s], acc[ti]);
        // lane: token 16*ti + l15; acc[ti][r] = feature 16*fi + 4*lq + r
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if ((g >> 1) == 0) qf[ti][4 * (g & 1) + r] = acc[ti][r] * scale;
            else kf[ti][4 * (g & 1) + r] = acc[ti][r];
          }
      }
    }
    if (rot_cos) {
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) {
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            float c, sn;                                                    // pair of features 16*fi + 4*lq + 2*pr (+1)
            if (ROT_RELOAD) {
              const int t = ti * 16 + l15, tt = t < L ? t : 0;
              c = rot_cos[tt * 16 + 8 * fi + 2 * lq + pr];
              sn = rot_sin[tt * 16 + 8 * fi + 2 * lq + pr];
            } else {
              c = rc[ti][2 * fi + pr];
              sn = rs[ti][2 * fi + pr];
            }
            const float qx = qf[ti][4 * fi + 2 * pr], qy = qf[ti][4 * fi + 2 * pr + 1];
            const float kx = kf[ti][4 * fi + 2 * pr], ky = kf[ti][4 * fi + 2 * pr + 1];
            qf[ti][4 * fi + 2 * pr] = qx * c - qy * sn;
            qf[ti][4 * fi + 2 * pr + 1] = qy * c + qx * sn;
            kf[ti][4 * fi + 2 * pr] = kx * c - ky * sn;
            kf[ti][4 * fi + 2 * pr + 1] = ky * c + kx * sn;
          }
      }
    }
    // ---- V: rows = tokens, cols = features; vf[half][4*ti + r] = v[token 16*ti + 4*lq + r][16*half + l15] ----
    float vf[2][4 * NT];
    {
      float wb[2][CQ];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const float* wsrc = WPACK ? wqkv + ((int64_t)((2 * HEADS + head) * 2 + half) * (CQ / 4) * 64 + lane) * 4
                                  : wqkv + ((int64_t)(2 * OUT_LD + head * DH + 16 * half + l15)) * C + CQ * lq;
#pragma unroll
        for (int q = 0; q < CQ / 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(wsrc + (WPACK ? 256 * q : 4 * q));
          wb[half][4 * q] = v.x; wb[half][4 * q + 1] = v.y; wb[half][4 * q + 2] = v.z; wb[half][4 * q + 3] = v.w;
        }
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f32x4 acc[NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) acc[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < CQ; ++s)
#pragma unroll
          for (int ti = 0; ti < NT; ++ti) acc[ti] = mfma_16x16x4(xf[ti][s], wb[half][s], acc[ti]);
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
          for (int r = 0; r < 4; ++r) vf[half][4 * ti + r] = acc[ti][r];
      }
    }

    attend_store<NT>(qf, kf, vf, head, L, l15, lq, bias, bias_vec, out, row0, hw, o_lds);
  }
}

template <int LP, int C>
__global__ __launch_bounds__(64, 2) void temporal_attn_fused_kernel(const float* __restrict__ x, int ldx, int heads_per_block,
                                                                 const float* __restrict__ wqkv,   // (768, C), gamma folded
                                                                 float* __restrict__ out, int batch, int frames, int hw,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ rot_cos,
                                                                 const float* __restrict__ rot_sin, float eps) {
  // grid.y splits the 8 heads over workgroups when there are few sequences (x is re-read per workgroup: 10-20 KB)
  tattn_heads<LP, C>(x, ldx, blockIdx.y * heads_per_block, heads_per_block, wqkv, out, frames, hw, blockIdx.x, bias, rot_cos, rot_sin, eps, nullptr);
}

// ... + to_out + residual in the same launch (round 4; lfdm_temporal_attention_fused_out_cl_f32): Residual(PreNorm(Attention)) complete,
// video_flow_diffusion.py:170-189, 286-363.  The separate to_out projection was a 25 us launch at 32x32 (0.67 GFLOP, 8 K chunks: all fixed
// cost) that re-read the 42 MB attention output the kernel above had just written in 64-byte pieces.  Here a workgroup of NW waves owns one
// pixel sequence: wave w runs heads [w * 8/NW, ...) exactly as above but parks its head outputs in the workgroup's LDS tile
// O[token][256] (row = 1 KB, the 16-byte quads XOR-swizzled by the token so that the unpadded tile reads conflict-free as MFMA A
// fragments), and after one barrier the waves share the projection out[t][c] = x[t][c] + sum_k O[t][k] Wout[c][k] on
// v_mfma_f32_16x16x4_f32: output tiles (16 tokens x 16 channels) dealt round-robin to the waves, A = O from LDS (one ds_read_b128 feeds
// four k-steps: k-slot lq of step (S, j) takes k = 16 S + 4 lq + j for BOTH operands), B = Wout straight from L2 in the same order.
// Both weight tensors arrive PACKED in operand order (ops.pack_tattn_weights): wqkv [3][8][2][4][64 lanes][4], wout [4 ct][16 S][64 lanes][4].
// LROWS = rows of the LDS tile (>= frames): 40 for the 40-frame videos (40 KB: four workgroups per CU), LP otherwise.
template <int LP, int LROWS, int NW>
__global__ __launch_bounds__(64 * NW, 2) void temporal_attn_fused_out_kernel(const float* __restrict__ x, int ldx,
                                                                          const float* __restrict__ wqkv, const float* __restrict__ wout,
                                                                          float* __restrict__ out, int ldo, int frames, int hw,
                                                                          const float* __restrict__ bias, const float* __restrict__ rot_cos,
                                                                          const float* __restrict__ rot_sin, float eps) {
  constexpr int C = 64, NT = LP / 16, CT = C / 16;
  __shared__ __attribute__((aligned(16))) float o_lds[LROWS * OUT_LD];
  const int wave = lfdm_uniform((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
  const int L = frames;
  const int64_t seq = blockIdx.x;
  tattn_heads<LP, C, (NW < 8), true>(x, ldx, wave * (HEADS / NW), HEADS / NW, wqkv, nullptr, frames, hw, seq, bias, rot_cos, rot_sin, eps, o_lds);
  __syncthreads();
  const int64_t b = seq / hw, pix = seq - b * hw;
  const int64_t row0 = b * frames * hw + pix;
  // output tiles (ti, ct), ti-major, dealt to the waves
  for (int tile = wave; tile < NT * CT; tile += NW) {
    const int ti = tile / CT, ct = tile - ti * CT;
    const int t = ti * 16 + l15;                        // A rows: this lane's token
    const int tr = t < L ? t : L - 1;                   // (padded tokens read a valid row; their results are never stored)
    const float* arow = o_lds + tr * OUT_LD;
    const float* brow = wout + ((int64_t)ct * (OUT_LD / 16) * 64 + lane) * 4;      // packed [ct][S][lane][4]: one contiguous 1 KB per load
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = acc;      // two chains: the 40-cycle dependent latency of 16x16x4 exceeds its 32-cycle issue
    // the tile's residual values are requested in front of its 128 MFMAs (round 6: loaded behind `if (tok < L)` at the store, they were a
    // dependent round trip at the end of every tile)
    float xr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tok = ti * 16 + 4 * lq + r;
      xr[r] = x[(row0 + (int64_t)(tok < L ? tok : L - 1) * hw) * ldx + ct * 16 + l15];
    }
#pragma unroll 4
    for (int S = 0; S < OUT_LD / 16; S += 2) {
      const float4 a = *reinterpret_cast<const float4*>(arow + (((4 * S + lq) ^ (tr & 15)) << 2));
      const float4 a1 = *reinterpret_cast<const float4*>(arow + (((4 * S + 4 + lq) ^ (tr & 15)) << 2));
      const float4 w = *reinterpret_cast<const float4*>(brow + 256 * S);
      const float4 w1 = *reinterpret_cast<const float4*>(brow + 256 * S + 256);
      acc = mfma_16x16x4(a.x, w.x, acc);
      acc1 = mfma_16x16x4(a1.x, w1.x, acc1);
      acc = mfma_16x16x4(a.y, w.y, acc);
      acc1 = mfma_16x16x4(a1.y, w1.y, acc1);
      acc = mfma_16x16x4(a.z, w.z, acc);
      acc1 = mfma_16x16x4(a1.z, w1.z, acc1);
      acc = mfma_16x16x4(a.w, w.w, acc);
      acc1 = mfma_16x16x4(a1.w, w1.w, acc1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += acc1[r];
    // D: lane = channel ct*16 + l15, register r = token ti*16 + 4*lq + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tok = ti * 16 + 4 * lq + r;
      if (tok < L) {
        const int64_t row = row0 + (int64_t)tok * hw;
        out[row * ldo + ct * 16 + l15] = acc[r] + xr[r];
      }
    }
  }
}

// Wide variant (C = 128, 256, ...): the channels are streamed in blocks of 64 with persistent Q^T / K^T / V
// accumulators, so the register budget does not grow with C; the sequence's rows are re-read from L1/L2 for every head
// and channel block (10-40 KB) and normalised on the fly with the token's mean / rstd computed once up front.
template <int LP>
__global__ __launch_bounds__(64, 1) void temporal_attn_fused_wide_kernel(const float* __restrict__ x, int ldx, int channels,
                                                                      int heads_per_block, const float* __restrict__ wqkv,
                                                                      float* __restrict__ out, int batch, int frames, int hw,
                                                                      const float* __restrict__ bias,
                                                                      const float* __restrict__ rot_cos,
                                                                      const float* __restrict__ rot_sin, float eps) {
  constexpr int NT = LP / 16;
  const int lane = threadIdx.x & 63;
  const int l15 = lane & 15, lq = lane >> 4;
  const int L = frames;
  const int64_t seq = blockIdx.x;
  const int64_t b = seq / hw, pix = seq - b * hw;
  const int64_t row0 = b * frames * hw + pix;
  const float scale = 0.17677669529663687f;
  const int ncb = channels / 64;

  float mean[NT], rstd[NT];
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    const int t = ti * 16 + l15;
    float s1 = 0.f, s2 = 0.f;
    if (t < L) {
      const float* src = x + (row0 + (int64_t)t * hw) * ldx + 16 * lq;
      for (int cb = 0; cb < ncb; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(src + 64 * cb + 4 * q);
          s1 += (v.x + v.y) + (v.z + v.w);
          s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    }
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    const float m = s1 / (float)channels;
    float var = s2 / (float)channels - m * m;
    if (var < 0.f) var = 0.f;
    mean[ti] = m;
    rstd[ti] = 1.0f / sqrtf(var + eps);
  }
  const bool bias_vec = bias && (L % 4 == 0) && ((((uintptr_t)bias) & 15) == 0);

  const int head_begin = blockIdx.y * heads_per_block;
#pragma unroll 1
  for (int head = head_begin; head < head_begin + heads_per_block; ++head) {
    f32x4 aq[2][NT], ak[2][NT], av[2][NT];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) aq[g][ti] = ak[g][ti] = av[g][ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int cb = 0; cb < ncb; ++cb) {
      // weight-row fragments of this channel block: k-slot lq <-> channels 64*cb + 16*lq + s
      float w[6][16];
#pragma unroll
      for (int g = 0; g < 6; ++g) {
        const float* wsrc = wqkv + ((int64_t)((g >> 1) * OUT_LD + head * DH + 16 * (g & 1) + l15)) * channels + 64 * cb + 16 * lq;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(wsrc + 4 * q);
          w[g][4 * q] = v.x; w[g][4 * q + 1] = v.y; w[g][4 * q + 2] = v.z; w[g][4 * q + 3] = v.w;
        }
      }
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) {
        const int t = ti * 16 + l15;
        float xf[16];
        if (t < L) {
          const float* src = x + (row0 + (int64_t)t * hw) * ldx + 64 * cb + 16 * lq;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(src + 4 * q);
            xf[4 * q] = (v.x - mean[ti]) * rstd[ti]; xf[4 * q + 1] = (v.y - mean[ti]) * rstd[ti];
            xf[4 * q + 2] = (v.z - mean[ti]) * rstd[ti]; xf[4 * q + 3] = (v.w - mean[ti]) * rstd[ti];
          }
        } else {
#pragma unroll
          for (int s2 = 0; s2 < 16; ++s2) xf[s2] = 0.f;
        }
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
          aq[0][ti] = mfma_16x16x4(w[0][s2], xf[s2], aq[0][ti]);      // Q^T, K^T: rows = features, cols = tokens
          aq[1][ti] = mfma_16x16x4(w[1][s2], xf[s2], aq[1][ti]);
          ak[0][ti] = mfma_16x16x4(w[2][s2], xf[s2], ak[0][ti]);
          ak[1][ti] = mfma_16x16x4(w[3][s2], xf[s2], ak[1][ti]);
          av[0][ti] = mfma_16x16x4(xf[s2], w[4][s2], av[0][ti]);      // V: rows = tokens, cols = features
          av[1][ti] = mfma_16x16x4(xf[s2], w[5][s2], av[1][ti]);
        }
      }
    }
    float qf[NT][8], kf[NT][8], vf[2][4 * NT];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int fi = 0; fi < 2; ++fi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          qf[ti][4 * fi + r] = aq[fi][ti][r] * scale;
          kf[ti][4 * fi + r] = ak[fi][ti][r];
          vf[fi][4 * ti + r] = av[fi][ti][r];
        }
    if (rot_cos) {
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) {
        const int t = ti * 16 + l15;
        const int tt = t < L ? t : 0;
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const float c = rot_cos[tt * 16 + 8 * fi + 2 * lq + pr], sn = rot_sin[tt * 16 + 8 * fi + 2 * lq + pr];
            const float qx = qf[ti][4 * fi + 2 * pr], qy = qf[ti][4 * fi + 2 * pr + 1];
            const float kx = kf[ti][4 * fi + 2 * pr], ky = kf[ti][4 * fi + 2 * pr + 1];
            qf[ti][4 * fi + 2 * pr] = qx * c - qy * sn;
            qf[ti][4 * fi + 2 * pr + 1] = qy * c + qx * sn;
            kf[ti][4 * fi + 2 * pr] = kx * c - ky * sn;
            kf[ti][4 * fi + 2 * pr + 1] = ky * c + kx * sn;
          }
      }
    }
    attend_store<NT>(qf, kf, vf, head, L, l15, lq, bias, bias_vec, out, row0, hw);
  }
}

int launch_fused_wide(const float* x, int ldx, int channels, const float* wqkv, float* out, int batch, int frames, int hw,
                      const float* bias, const float* rot_cos, const float* rot_sin, float eps, hipStream_t stream) {
  const int64_t nseq = (int64_t)batch * hw;
  int hpb = 8;
  while (hpb > 1 && nseq * (HEADS / hpb) < 2048) hpb >>= 1;
  const dim3 grid((unsigned)nseq, (unsigned)(HEADS / hpb)), block(64);
  if (frames <= 16) LFDM_LAUNCH((temporal_attn_fused_wide_kernel<16>), grid, block, 0, stream, x, ldx, channels, hpb, wqkv, out, batch, frames, hw, bias, rot_cos, rot_sin, eps);
  else if (frames <= 32) LFDM_LAUNCH((temporal_attn_fused_wide_kernel<32>), grid, block, 0, stream, x, ldx, channels, hpb, wqkv, out, batch, frames, hw, bias, rot_cos, rot_sin, eps);
  else if (frames <= 48) LFDM_LAUNCH((temporal_attn_fused_wide_kernel<48>), grid, block, 0, stream, x, ldx, channels, hpb, wqkv, out, batch, frames, hw, bias, rot_cos, rot_sin, eps);
  else LFDM_LAUNCH((temporal_attn_fused_wide_kernel<64>), grid, block, 0, stream, x, ldx, channels, hpb, wqkv, out, batch, frames, hw, bias, rot_cos, rot_sin, eps);
  return lfdm_check_launch("temporal_attention_fused_wide");
}

template <int C>
int launch_fused(const float* x, int ldx, const float* wqkv, float* out, int batch, int frames, int hw, const float* bias,
                 const float* rot_cos, const float* rot_sin, float eps, hipStream_t stream) {
  const int64_t nseq = (int64_t)batch * hw;
  int hpb = 8;                                       // heads per workgroup: aim at >= 2048 workgroups
  while (hpb > 1 && nseq * (HEADS / hpb) < 2048) hpb >>= 1;
  if (const char* e = lfdm_knob("LFDM_TATTN_HPB")) {    // experiment knob (tools/bench_attn.py)
    const int v = atoi(e);
    if (v == 1 || v == 2 || v == 4 || v == 8) hpb = v;
  }
  const dim3 grid((unsigned)nseq, (unsigned)(HEADS / hpb)), block(64);
  if (frames <= 16) LFDM_LAUNCH((temporal_attn_fused_kernel<16, C>), grid, block, 0, stream, x, ldx, hpb, wqkv, out, batch, frames, hw, bias, rot_cos, rot_sin, eps);
  else if (frames <= 32) LFDM_LAUNCH((temporal_attn_fused_kernel<32, C>), grid, block, 0, stream, x, ldx, hpb, wqkv, out, batch, frames, hw, bias, rot_cos, rot_sin, eps);
  else if (frames <= 48) LFDM_LAUNCH((temporal_attn_fused_kernel<48, C>), grid, block, 0, stream, x, ldx, hpb, wqkv, out, batch, frames, hw, bias, rot_cos, rot_sin, eps);
  else LFDM_LAUNCH((temporal_attn_fused_kernel<64, C>), grid, block, 0, stream, x, ldx, hpb, wqkv, out, batch, frames, hw, bias, rot_cos, rot_sin, eps);
  return lfdm_check_launch("temporal_attention_fused");
}

template <int LP, int LROWS>
int launch_fused_out_lp(const float* x, int ldx, const float* wqkv, const float* wout, float* out, int ldo, int batch, int frames, int hw,
                        const float* bias, const float* rot_cos, const float* rot_sin, float eps, hipStream_t stream) {
  const int64_t nseq = (int64_t)batch * hw;
  // waves per sequence: aim at >= 2048 wavefronts (like the kernel above), but never ONE wave per workgroup: the 40 KB tile admits four
  // workgroups per CU, i.e. one wave per SIMD (batched shapes lost 4 % to that before this floor, profiles/r04_t_other_configs.json)
  int nw = 2;
  while (nw < 8 && nseq * nw < 2048) nw <<= 1;
  if (const char* e = getenv("LFDM_TATTN_OUT_NW")) {      // experiment knob
    const int v = atoi(e);
    if (v == 1 || v == 2 || v == 4 || v == 8) nw = v;
  }
  const dim3 grid((unsigned)nseq);
  if (nw == 1) LFDM_LAUNCH((temporal_attn_fused_out_kernel<LP, LROWS, 1>), grid, dim3(64), 0, stream, x, ldx, wqkv, wout, out, ldo, frames, hw, bias, rot_cos, rot_sin, eps);
  else if (nw == 2) LFDM_LAUNCH((temporal_attn_fused_out_kernel<LP, LROWS, 2>), grid, dim3(128), 0, stream, x, ldx, wqkv, wout, out, ldo, frames, hw, bias, rot_cos, rot_sin, eps);
  else if (nw == 4) LFDM_LAUNCH((temporal_attn_fused_out_kernel<LP, LROWS, 4>), grid, dim3(256), 0, stream, x, ldx, wqkv, wout, out, ldo, frames, hw, bias, rot_cos, rot_sin, eps);
  else LFDM_LAUNCH((temporal_attn_fused_out_kernel<LP, LROWS, 8>), grid, dim3(512), 0, stream, x, ldx, wqkv, wout, out, ldo, frames, hw, bias, rot_cos, rot_sin, eps);
  return lfdm_check_launch("temporal_attention_fused_out");
}

}  // namespace

extern "C" int lfdm_temporal_attention_fused_out_cl_f32(const float* x, int ldx, int channels, const float* wqkv, const float* wout,
                                                        float* out, int ldo, int batch, int frames, int hw, const float* bias,
                                                        const float* rot_cos, const float* rot_sin, float ln_eps, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !wqkv || !wout || !out || batch <= 0 || frames <= 0 || frames > 64 || hw <= 0 || channels != 64 || ldx < 64 || ldx % 4 != 0 ||
      ldo < 64 || (((uintptr_t)x | (uintptr_t)wqkv | (uintptr_t)wout) & 15) || ((rot_cos == nullptr) != (rot_sin == nullptr)) || out == x) {
    lfdm_set_error("temporal_attention_fused_out: needs C == 64, frames <= 64, 16-byte aligned rows, out != x");
    return LFDM_EINVAL;
  }
  if (frames <= 16) return launch_fused_out_lp<16, 16>(x, ldx, wqkv, wout, out, ldo, batch, frames, hw, bias, rot_cos, rot_sin, ln_eps, stream);
  if (frames <= 32) return launch_fused_out_lp<32, 32>(x, ldx, wqkv, wout, out, ldo, batch, frames, hw, bias, rot_cos, rot_sin, ln_eps, stream);
  if (frames <= 40) return launch_fused_out_lp<48, 40>(x, ldx, wqkv, wout, out, ldo, batch, frames, hw, bias, rot_cos, rot_sin, ln_eps, stream);
  if (frames <= 48) return launch_fused_out_lp<48, 48>(x, ldx, wqkv, wout, out, ldo, batch, frames, hw, bias, rot_cos, rot_sin, ln_eps, stream);
  return launch_fused_out_lp<64, 64>(x, ldx, wqkv, wout, out, ldo, batch, frames, hw, bias, rot_cos, rot_sin, ln_eps, stream);
}

extern "C" int lfdm_temporal_attention_fused_cl_f32(const float* x, int ldx, int channels, const float* wqkv, float* out,
                                                    int batch, int frames, int hw, const float* bias,
                                                    const float* rot_cos, const float* rot_sin, float ln_eps,
                                                    lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !wqkv || !out || batch <= 0 || frames <= 0 || frames > 64 || hw <= 0 || ldx < channels || ldx % 4 != 0 ||
      (((uintptr_t)x | (uintptr_t)wqkv) & 15) || ((rot_cos == nullptr) != (rot_sin == nullptr)) ||
      channels < 64 || channels % 64 != 0) {
    lfdm_set_error("temporal_attention_fused: needs C % 64 == 0, frames <= 64, 16-byte aligned rows");
    return LFDM_EINVAL;
  }
  if (channels == 64) return launch_fused<64>(x, ldx, wqkv, out, batch, frames, hw, bias, rot_cos, rot_sin, ln_eps, stream);
  return launch_fused_wide(x, ldx, channels, wqkv, out, batch, frames, hw, bias, rot_cos, rot_sin, ln_eps, stream);
}
