// Backward of the attention kernels (include/lfdm_hip.h, training section).
//
// 1. lfdm_attention_bwd_cl_f32 - gradient of Attention.forward (video_flow_diffusion.py:303-363) w.r.t. the
//    qkv rows and the relative-position bias.  One wavefront owns one (sequence, head) exactly like the
//    forward kernel: Q (scaled, rotated), K (rotated), V and dO are staged in LDS, S = QK^T + bias and the
//    softmax are RECOMPUTED (nothing but qkv was saved), then on v_mfma_f32_16x16x4_f32:
//        dP = dO V^T,  dS = P o (dP - rowsum(P o dP)),  dV = P^T dO,  dQr = dS K,  dKr = dS^T Q,
//    the rotary rotation is undone (R^T) and 32^-0.5 re-applied for dq.  The L x L scores never leave
//    LDS/registers.  The bias gradient is accumulated per wavefront over a grid-stride loop (the stride is
//    a multiple of 8, so a wavefront always serves the same head) and written as per-wave partials that
//    lfdm_sum_leading_f32 adds in a fixed order.
// 2. lfdm_linear_attention_bwd_cl_f32 - gradient of SpatialLinearAttention's core (:254-263):
//        ks = softmax_n(k), qs = softmax_d(q)*scale, ctx = ks v^T, out = ctx^T qs
//        dctx[d][e] = sum_n qs[d,n] dout[e,n];  dqs = ctx dout;  dks = dctx v;  dv = dctx^T ks
//        dk = ks o (dks - sum_e dctx[d,e] ctx[d,e]);  dq = scale * qs/scale o (dqs - sum_d (qs/scale) dqs)
//    A per-(frame, head) reduction kernel (k-softmax statistics, ctx, dctx) and a per-token apply kernel.
#include <stdlib.h>

#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

constexpr int HEADS = 8;
constexpr int DH = 32;
constexpr int QKV_LD = 3 * HEADS * DH;  // 768
constexpr int OUT_LD = HEADS * DH;      // 256
constexpr int SQ = 34;                  // LDS row stride of Q, K
constexpr int SV = 36;                  // LDS row stride of V, dO
constexpr float ATT_SCALE = 0.17677669529663687f;  // 32^-0.5

// LROWS (>= the sequence length, <= LP): rows of the wave's LDS tiles.  The 40-frame videos use 40 of the 48 padded rows: the three GEMMs that
// contract over tokens run 10 instead of 12 k-steps.  (Five waves per CU would fit the 160 KB with 30.4 KB tiles, but the kernel lives on the
// 512 registers a lone wave per SIMD may use: at two waves per SIMD it spills 336 VGPRs and runs 2.1x slower - measured,
// profiles/r05_b_attn_bwd_ab.txt.)  The five tiles of a wave are private to it - the phases are ordered by lfdm_wave_lds_sync, not by
// workgroup barriers: the waves of a workgroup run their units at their own pace.
template <int LP, int WPB, int LROWS = LP>
__global__ __launch_bounds__(64 * WPB) void attention_bwd_kernel(
    const float* __restrict__ qkv, const float* __restrict__ dout, float* __restrict__ dqkv, int batch, int frames,
    int hw, int mode, const float* __restrict__ bias, const float* __restrict__ rot_cos,
    const float* __restrict__ rot_sin, float* __restrict__ dbias_part) {
  constexpr int NT = LP / 16;
  constexpr int SP = LP + 2;
  constexpr int PER_WAVE = LROWS * (2 * SQ + 2 * SV) + LROWS * SP;
  __shared__ __attribute__((aligned(16))) float smem[WPB * PER_WAVE];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* Qs = smem + wave * PER_WAVE;
  float* Ks = Qs + LROWS * SQ;
  float* Vs = Ks + LROWS * SQ;
  float* Gs = Vs + LROWS * SV;     // dO
  float* Ps = Gs + LROWS * SV;     // P, later dS
  // rows >= LROWS do not exist: padded tokens (t >= L) read row LROWS - 1 (finite values whose products are masked / never stored) and are
  // not written
  auto rowc = [](int t) { return LROWS < LP ? (t < LROWS ? t : LROWS - 1) : t; };

  const int L = mode == 0 ? frames : hw;
  const int64_t nseq = mode == 0 ? (int64_t)batch * hw : (int64_t)batch * frames;
  const int64_t units = nseq * HEADS;
  const int l15 = lane & 15, lq = lane >> 4;

  f32x4 db[NT][NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) db[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // One wavefront per SIMD (the LDS tiles of four sequences fill a CU) means nothing hides a global round trip: the next
  // unit's Q / K / V / dO rows are requested into registers as soon as the current ones have been staged, and arrive under
  // the five GEMMs of the current unit.
  constexpr int NR = LP / 8;
  const int rr = lane >> 3, c4 = lane & 7;
  float4 qv[NR], kv[NR], vv[NR], gv[NR];
  auto geometry = [&](int64_t unit, bool& valid, int& head, int64_t& row0, int64_t& tstride) {
    valid = unit < units;
    const int64_t seq = valid ? unit / HEADS : 0;
    head = valid ? (int)(unit - seq * HEADS) : 0;
    if (mode == 0) {
      const int64_t b = seq / hw, pix = seq - b * hw;
      row0 = b * frames * hw + pix;
      tstride = hw;
    } else {
      row0 = seq * hw;
      tstride = 1;
    }
  };
  auto fetch = [&](int64_t unit) {
    bool valid;
    int head;
    int64_t row0, tstride;
    geometry(unit, valid, head, row0, tstride);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int t = rr + 8 * i;
      qv[i] = kv[i] = vv[i] = gv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid && t < L) {
        const int64_t row = row0 + (int64_t)t * tstride;
        const float* src = qkv + row * QKV_LD + head * DH + 4 * c4;
        qv[i] = *reinterpret_cast<const float4*>(src);
        kv[i] = *reinterpret_cast<const float4*>(src + OUT_LD);
        vv[i] = *reinterpret_cast<const float4*>(src + 2 * OUT_LD);
        gv[i] = *reinterpret_cast<const float4*>(dout + row * OUT_LD + head * DH + 4 * c4);
      }
    }
  };
  // BATCH (round 6): the unit's table loads - rotary factors, relative-position bias - are requested in batches in front of the work that hides them instead
  // of one by one at their use; not for the two instantiations that are at the register limit already (48 rows x 48, 64 x 64: the batches spill there)
  constexpr bool BATCH = LROWS < LP || LP <= 32;
  constexpr bool PREFETCH = LP <= 48;          // (64 tokens - the mid spatial attention of the 256x256 configuration, a handful of sequences - is at the register limit already: rows fetched at the top of the unit as before)
  if (PREFETCH) fetch((int64_t)blockIdx.x * WPB + wave);

  for (int64_t base = (int64_t)blockIdx.x * WPB; base < units; base += (int64_t)gridDim.x * WPB) {
    const int64_t unit = base + wave;
    bool valid;
    int head;
    int64_t row0, tstride;
    geometry(unit, valid, head, row0, tstride);
    if (!PREFETCH) fetch(unit);

    // ---- stage Q (scaled + rotary), K (rotary), V, dO ----
    {
      // rotary factors of this lane's rows / feature pairs, requested together (round 6: each of the twelve 8-byte loads sat behind
      // `if (rot_cos && t < L)` with an s_waitcnt vmcnt(0) of its own - twelve L2 round trips in a row per (sequence, head))
      float2 rcos[NR], rsin[NR];
      if (BATCH) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int t = rr + 8 * i, tt = t < L ? t : 0;
          rcos[i] = rot_cos ? *reinterpret_cast<const float2*>(rot_cos + tt * 16 + 2 * c4) : make_float2(1.f, 1.f);
          rsin[i] = rot_cos ? *reinterpret_cast<const float2*>(rot_sin + tt * 16 + 2 * c4) : make_float2(0.f, 0.f);
        }
      }
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int t = rr + 8 * i;
        float4 q = qv[i], k = kv[i];
        q.x *= ATT_SCALE; q.y *= ATT_SCALE; q.z *= ATT_SCALE; q.w *= ATT_SCALE;
        if (rot_cos && t < L) {
          const float c0 = BATCH ? rcos[i].x : rot_cos[t * 16 + 2 * c4], s0 = BATCH ? rsin[i].x : rot_sin[t * 16 + 2 * c4];
          const float c1 = BATCH ? rcos[i].y : rot_cos[t * 16 + 2 * c4 + 1], s1 = BATCH ? rsin[i].y : rot_sin[t * 16 + 2 * c4 + 1];
          float4 qr, kr;
          qr.x = q.x * c0 - q.y * s0; qr.y = q.y * c0 + q.x * s0;
          qr.z = q.z * c1 - q.w * s1; qr.w = q.w * c1 + q.z * s1;
          kr.x = k.x * c0 - k.y * s0; kr.y = k.y * c0 + k.x * s0;
          kr.z = k.z * c1 - k.w * s1; kr.w = k.w * c1 + k.z * s1;
          q = qr; k = kr;
        }
        if (LROWS < LP && t >= LROWS) continue;
        float* dq = Qs + t * SQ + 4 * c4;
        dq[0] = q.x; dq[1] = q.y; dq[2] = q.z; dq[3] = q.w;
        float* dk = Ks + t * SQ + 4 * c4;
        dk[0] = k.x; dk[1] = k.y; dk[2] = k.z; dk[3] = k.w;
        *reinterpret_cast<float4*>(Vs + t * SV + 4 * c4) = vv[i];
        *reinterpret_cast<float4*>(Gs + t * SV + 4 * c4) = gv[i];
      }
    }
    // The relative-position bias is the INITIAL value of the score accumulators (round 6): its 4 NT^2 loads are requested together, in front of the
    // next unit's prefetch (loads retire in order: behind it, the first MFMA's wait for the bias would also wait for the prefetched rows), and the
    // products accumulate on top.  Before, each value was loaded behind `else if (bias && row < L)` inside the softmax and used at once: 36 dependent
    // L1 / L2 round trips per (sequence, head).
    f32x4 p[NT][NT], dp[NT][NT];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) {
        const int col = tj * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = ti * 16 + lq * 4 + r;
          const bool in = BATCH && bias && row < L && col < L;
          const float bv = (BATCH && bias) ? bias[((int64_t)head * L + (row < L ? row : 0)) * L + (col < L ? col : 0)] : 0.f;
          p[ti][tj][r] = in ? bv : 0.f;
        }
      }
    if (PREFETCH && base + (int64_t)gridDim.x * WPB < units) fetch(unit + (int64_t)gridDim.x * WPB);       // in flight under this unit's GEMMs
    lfdm_wave_lds_sync();

    // ---- S = Q K^T (+ bias), dP = dO V^T ----
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) {
        f32x4 acc = p[ti][tj], acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < DH / 4; ++s) {
          const float a = Qs[rowc(ti * 16 + l15) * SQ + 4 * s + lq];
          const float b = Ks[rowc(tj * 16 + l15) * SQ + 4 * s + lq];
          acc = mfma_16x16x4(a, b, acc);
          const float a2 = Gs[rowc(ti * 16 + l15) * SV + 4 * s + lq];
          const float b2 = Vs[rowc(tj * 16 + l15) * SV + 4 * s + lq];
          acc2 = mfma_16x16x4(a2, b2, acc2);
        }
        p[ti][tj] = acc;
        dp[ti][tj] = acc2;
      }

    // ---- softmax (same arithmetic as the forward kernel), dS, bias gradient ----
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = ti * 16 + lq * 4 + r;
        float m = -3.0e38f;
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
          const int col = tj * 16 + l15;
          float v = p[ti][tj][r];
          if (col >= L) v = -3.0e38f;
          else if (!BATCH && bias && row < L) v += bias[((int64_t)head * L + row) * L + col];
          p[ti][tj][r] = v;
          m = fmaxf(m, v);
        }
#pragma unroll
        for (int x = 1; x < 16; x <<= 1) m = fmaxf(m, __shfl_xor(m, x));
        float sum = 0.f;
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
          const int col = tj * 16 + l15;
          const float e = col < L ? expf(p[ti][tj][r] - m) : 0.f;
          p[ti][tj][r] = e;
          sum += e;
        }
#pragma unroll
        for (int x = 1; x < 16; x <<= 1) sum += __shfl_xor(sum, x);
        float dot = 0.f;
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
          const float pv = p[ti][tj][r] / sum;
          p[ti][tj][r] = pv;
          if (LROWS == LP || row < LROWS) Ps[row * SP + tj * 16 + l15] = pv;
          dot += pv * dp[ti][tj][r];
        }
#pragma unroll
        for (int x = 1; x < 16; x <<= 1) dot += __shfl_xor(dot, x);
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
          const float ds = (valid && row < L) ? p[ti][tj][r] * (dp[ti][tj][r] - dot) : 0.f;
          dp[ti][tj][r] = ds;                     // dp now holds dS
          db[ti][tj][r] += ds;
        }
      }
    }
    lfdm_wave_lds_sync();

    // ---- dV = P^T dO ----
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
      f32x4 o[2];
      o[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      o[1] = o[0];
#pragma unroll
      for (int s = 0; s < LROWS / 4; ++s) {      // (tokens >= LROWS do not exist; rows in [L, LROWS) hold zeros)
        const float a = Ps[(4 * s + lq) * SP + tj * 16 + l15];      // A[m = token j][k = token i] = P[i][j]
        const float b0 = Gs[(4 * s + lq) * SV + l15];
        const float b1 = Gs[(4 * s + lq) * SV + 16 + l15];
        o[0] = mfma_16x16x4(a, b0, o[0]);
        o[1] = mfma_16x16x4(a, b1, o[1]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = tj * 16 + lq * 4 + r;
        if (valid && t < L) {
          float* dst = dqkv + (row0 + (int64_t)t * tstride) * QKV_LD + 2 * OUT_LD + head * DH;
          dst[l15] = o[0][r];
          dst[16 + l15] = o[1][r];
        }
      }
    }
    lfdm_wave_lds_sync();
    // ---- dS -> LDS (over P) ----
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
          if (LROWS == LP || ti * 16 + lq * 4 + r < LROWS) Ps[(ti * 16 + lq * 4 + r) * SP + tj * 16 + l15] = dp[ti][tj][r];
    lfdm_wave_lds_sync();

    // ---- dQr = dS K ; dKr = dS^T Q ; undo rotary ; store ----
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
      f32x4 oq[2], ok[2];
      oq[0] = oq[1] = ok[0] = ok[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // the tile's sixteen rotary factors are requested in front of its MFMAs (they were loaded one by one at their use, each behind a wait)
      float urc[4][2], urs[4][2];
      if (BATCH && rot_cos) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int t = ti * 16 + lq * 4 + r, tt = t < L ? t : 0;
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            urc[r][hh] = rot_cos[tt * 16 + ((16 * hh + l15) >> 1)];
            urs[r][hh] = rot_sin[tt * 16 + ((16 * hh + l15) >> 1)];
          }
        }
      }
#pragma unroll
      for (int s = 0; s < LROWS / 4; ++s) {      // (tokens >= LROWS do not exist; rows in [L, LROWS) hold zeros)
        const float aq = Ps[rowc(ti * 16 + l15) * SP + 4 * s + lq];      // dS[i][j = 4s+lq]
        const float ak = Ps[(4 * s + lq) * SP + ti * 16 + l15];      // dS[i = 4s+lq][j]
        const float bk0 = Ks[(4 * s + lq) * SQ + l15], bk1 = Ks[(4 * s + lq) * SQ + 16 + l15];
        const float bq0 = Qs[(4 * s + lq) * SQ + l15], bq1 = Qs[(4 * s + lq) * SQ + 16 + l15];
        oq[0] = mfma_16x16x4(aq, bk0, oq[0]);
        oq[1] = mfma_16x16x4(aq, bk1, oq[1]);
        ok[0] = mfma_16x16x4(ak, bq0, ok[0]);
        ok[1] = mfma_16x16x4(ak, bq1, ok[1]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = ti * 16 + lq * 4 + r;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          float gq = oq[hh][r], gk = ok[hh][r];
          if (rot_cos) {
            // feature d = 16*hh + l15; pair index d>>1; even lane holds x, odd lane holds y of the pair
            const float gq_o = __shfl_xor(gq, 1), gk_o = __shfl_xor(gk, 1);
            const int tt = t < L ? t : 0;
            const float c = BATCH ? urc[r][hh] : rot_cos[tt * 16 + ((16 * hh + l15) >> 1)], sn = BATCH ? urs[r][hh] : rot_sin[tt * 16 + ((16 * hh + l15) >> 1)];
            if ((l15 & 1) == 0) {          // dx = gx*c + gy*s
              gq = gq * c + gq_o * sn;
              gk = gk * c + gk_o * sn;
            } else {                        // dy = gy*c - gx*s
              gq = gq * c - gq_o * sn;
              gk = gk * c - gk_o * sn;
            }
          }
          if (valid && t < L) {
            float* dst = dqkv + (row0 + (int64_t)t * tstride) * QKV_LD + head * DH + 16 * hh + l15;
            dst[0] = gq * ATT_SCALE;
            dst[OUT_LD] = gk;
          }
        }
      }
    }
    lfdm_wave_lds_sync();
  }

  if (dbias_part) {
    // per-wave partial [LP][LP] rows; global wave id w = blockIdx.x*WPB + wave serves head w % 8
    float* dst = dbias_part + ((int64_t)blockIdx.x * WPB + wave) * (L * L);
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
          const int row = ti * 16 + lq * 4 + r, col = tj * 16 + l15;
          if (row < L && col < L) dst[row * L + col] = db[ti][tj][r];
        }
  }
}

bool attn_bwd_rows40_enabled() {          // experiment knob (tools/bench_attn_bwd.py): LFDM_ATTN_BWD_ROWS40=0 -> the 48-row / four-wave form
  const char* e = lfdm_knob("LFDM_ATTN_BWD_ROWS40");
  return !(e && e[0] == '0');
}

int attn_bwd_blocks(int64_t units, int wpb) {
  int64_t nb = (units + wpb - 1) / wpb;
  const int cap = 2048 / wpb;            // 2048 wavefronts in flight; a multiple of 8 per grid stride
  if (nb > cap) nb = cap;
  // grid*wpb must be a multiple of 8 so that a wavefront keeps its head
  while ((nb * wpb) % 8 != 0) ++nb;
  return (int)nb;
}

// ---------------- linear attention ----------------
constexpr int LA_WS = 2 * DH * DH + 4 * DH;   // per (frame, head): ctx | dctx | kmax | ksum | rowdot | pad

// grid (n_frames*8), 256 threads.  thread (d = tid>>3, e0 = 4*(tid&7)) owns ctx[d][e0..e0+3] and dctx likewise.
__global__ __launch_bounds__(256) void linattn_bwd_context_kernel(const float* __restrict__ qkv,
                                                                  const float* __restrict__ dout, int hw,
                                                                  float* __restrict__ ws) {
  __shared__ float red[4][32];
  __shared__ float kmax[32];
  __shared__ float ek[64][33], qs[64][33];
  __shared__ __attribute__((aligned(16))) float vv[64][36], go[64][36];
  const int tid = threadIdx.x;
  const int f = blockIdx.x >> 3, h = blockIdx.x & 7;
  const float* qbase = qkv + (int64_t)f * hw * QKV_LD + h * DH;
  const float* kbase = qbase + OUT_LD;
  const float* vbase = kbase + OUT_LD;
  const float* gbase = dout + (int64_t)f * hw * OUT_LD + h * DH;

  {  // pass 1: per-feature max of k over tokens
    const int c4 = tid & 7, part = tid >> 3;
    float4 m = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
    for (int n = part; n < hw; n += 32) {
      const float4 v = *reinterpret_cast<const float4*>(kbase + (int64_t)n * QKV_LD + 4 * c4);
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
#pragma unroll
    for (int x = 8; x <= 32; x <<= 1) {
      m.x = fmaxf(m.x, __shfl_xor(m.x, x)); m.y = fmaxf(m.y, __shfl_xor(m.y, x));
      m.z = fmaxf(m.z, __shfl_xor(m.z, x)); m.w = fmaxf(m.w, __shfl_xor(m.w, x));
    }
    if ((tid & 63) < 8) {
      const int w = tid >> 6;
      red[w][4 * c4 + 0] = m.x; red[w][4 * c4 + 1] = m.y; red[w][4 * c4 + 2] = m.z; red[w][4 * c4 + 3] = m.w;
    }
    __syncthreads();
    if (tid < 32) kmax[tid] = fmaxf(fmaxf(red[0][tid], red[1][tid]), fmaxf(red[2][tid], red[3][tid]));
    __syncthreads();
  }

  const int d = tid >> 3, e0 = (tid & 7) * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, dacc[4] = {0.f, 0.f, 0.f, 0.f};
  float ssum = 0.f;
  const int ln = tid >> 3, lc4 = tid & 7;
  for (int n0 = 0; n0 < hw; n0 += 64) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int nl = ln + 32 * u;
      const int n = n0 + nl;
      const bool ok = n < hw;
      float4 rk = make_float4(0.f, 0.f, 0.f, 0.f), rv = rk, rq = rk, rg = rk;
      if (ok) {
        rq = *reinterpret_cast<const float4*>(qbase + (int64_t)n * QKV_LD + 4 * lc4);
        rk = *reinterpret_cast<const float4*>(kbase + (int64_t)n * QKV_LD + 4 * lc4);
        rv = *reinterpret_cast<const float4*>(vbase + (int64_t)n * QKV_LD + 4 * lc4);
        rg = *reinterpret_cast<const float4*>(gbase + (int64_t)n * OUT_LD + 4 * lc4);
      }
      ek[nl][4 * lc4 + 0] = ok ? expf(rk.x - kmax[4 * lc4 + 0]) : 0.f;
      ek[nl][4 * lc4 + 1] = ok ? expf(rk.y - kmax[4 * lc4 + 1]) : 0.f;
      ek[nl][4 * lc4 + 2] = ok ? expf(rk.z - kmax[4 * lc4 + 2]) : 0.f;
      ek[nl][4 * lc4 + 3] = ok ? expf(rk.w - kmax[4 * lc4 + 3]) : 0.f;
      // q softmax over the 32 features of this token: the 8 lanes (lc4) that share the token
      float m = fmaxf(fmaxf(rq.x, rq.y), fmaxf(rq.z, rq.w));
#pragma unroll
      for (int x = 1; x < 8; x <<= 1) m = fmaxf(m, __shfl_xor(m, x));
      float4 eq;
      eq.x = expf(rq.x - m); eq.y = expf(rq.y - m); eq.z = expf(rq.z - m); eq.w = expf(rq.w - m);
      float sm = (eq.x + eq.y) + (eq.z + eq.w);
#pragma unroll
      for (int x = 1; x < 8; x <<= 1) sm += __shfl_xor(sm, x);
      qs[nl][4 * lc4 + 0] = ok ? eq.x / sm * ATT_SCALE : 0.f;
      qs[nl][4 * lc4 + 1] = ok ? eq.y / sm * ATT_SCALE : 0.f;
      qs[nl][4 * lc4 + 2] = ok ? eq.z / sm * ATT_SCALE : 0.f;
      qs[nl][4 * lc4 + 3] = ok ? eq.w / sm * ATT_SCALE : 0.f;
      *reinterpret_cast<float4*>(&vv[nl][4 * lc4]) = rv;
      *reinterpret_cast<float4*>(&go[nl][4 * lc4]) = rg;
    }
    __syncthreads();
#pragma unroll 8
    for (int n = 0; n < 64; ++n) {
      const float e = ek[n][d];
      const float qd = qs[n][d];
      const float4 v4 = *reinterpret_cast<const float4*>(&vv[n][e0]);
      const float4 g4 = *reinterpret_cast<const float4*>(&go[n][e0]);
      acc[0] = fmaf(e, v4.x, acc[0]); acc[1] = fmaf(e, v4.y, acc[1]);
      acc[2] = fmaf(e, v4.z, acc[2]); acc[3] = fmaf(e, v4.w, acc[3]);
      dacc[0] = fmaf(qd, g4.x, dacc[0]); dacc[1] = fmaf(qd, g4.y, dacc[1]);
      dacc[2] = fmaf(qd, g4.z, dacc[2]); dacc[3] = fmaf(qd, g4.w, dacc[3]);
      ssum += e;
    }
    __syncthreads();
  }
  float* w = ws + (int64_t)blockIdx.x * LA_WS;
  float rd = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float c = acc[i] / ssum;
    w[d * DH + e0 + i] = c;
    w[DH * DH + d * DH + e0 + i] = dacc[i];
    rd += c * dacc[i];
  }
#pragma unroll
  for (int x = 1; x < 8; x <<= 1) rd += __shfl_xor(rd, x);
  if ((tid & 7) == 0) {
    w[2 * DH * DH + d] = kmax[d];
    w[2 * DH * DH + DH + d] = ssum;
    w[2 * DH * DH + 2 * DH + d] = rd;
  }
}

// grid (ceil(hw / LA_TOK), n_frames); 256 threads = 8 heads x 32 features: a workgroup walks LA_TOK tokens of one frame and reads / writes
// whole 3 KB qkv rows (1 KB contiguous per operand and token - round 4's grid gave every (frame, head) its own workgroups, i.e. 128-byte pieces
// 3 KB apart: 2.1 TB/s).  Thread (head, feature j) keeps row j of ctx, row j of dctx and column j of dctx of its head in REGISTERS (96 floats,
// loaded once per workgroup; round 4 read them from LDS inside the loop, six ds_read_b32 per three FMAs); a token's dout / v / softmax(k)
// vectors are exchanged through a wave-private LDS row (the 32 lanes of a head are half a wavefront: a wave-level fence orders them) and read
// back as broadcast float4s.  Four tokens per trip keep 16 loads in flight per thread.
constexpr int LA_TOK = 64;
__global__ __launch_bounds__(256) void linattn_bwd_apply_kernel(const float* __restrict__ qkv,
                                                                const float* __restrict__ dout,
                                                                const float* __restrict__ ws, int hw,
                                                                float* __restrict__ dqkv) {
  __shared__ __attribute__((aligned(16))) float t_go[8][4][DH], t_v[8][4][DH], t_ks[8][4][DH];
  const int tid = threadIdx.x;
  const int f = blockIdx.y;
  const int h = tid >> 5, j = tid & 31;
  const float* w = ws + ((int64_t)f * HEADS + h) * LA_WS;
  float cj[DH], dj[DH], dtj[DH];
#pragma unroll
  for (int e4 = 0; e4 < DH / 4; ++e4) {
    const float4 a = *reinterpret_cast<const float4*>(w + j * DH + 4 * e4);
    const float4 b = *reinterpret_cast<const float4*>(w + DH * DH + j * DH + 4 * e4);
    cj[4 * e4] = a.x; cj[4 * e4 + 1] = a.y; cj[4 * e4 + 2] = a.z; cj[4 * e4 + 3] = a.w;
    dj[4 * e4] = b.x; dj[4 * e4 + 1] = b.y; dj[4 * e4 + 2] = b.z; dj[4 * e4 + 3] = b.w;
  }
#pragma unroll
  for (int e = 0; e < DH; ++e) dtj[e] = w[DH * DH + e * DH + j];
  const float kmax_j = w[2 * DH * DH + j], ksum_j = w[2 * DH * DH + DH + j], rowdot_j = w[2 * DH * DH + 2 * DH + j];
  const int n0 = blockIdx.x * LA_TOK;
  for (int it = 0; it < LA_TOK / 4; ++it) {
    float q[4], k[4], v[4], g[4];
    bool ok[4];
    int64_t row[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = n0 + 4 * it + u;
      ok[u] = n < hw;
      row[u] = (int64_t)f * hw + (ok[u] ? n : 0);
      const float* src = qkv + row[u] * QKV_LD + h * DH + j;
      q[u] = ok[u] ? src[0] : 0.f;
      k[u] = ok[u] ? src[OUT_LD] : 0.f;
      v[u] = ok[u] ? src[2 * OUT_LD] : 0.f;
      g[u] = ok[u] ? dout[row[u] * OUT_LD + h * DH + j] : 0.f;
    }
    float s[4], ks[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float m = q[u];
#pragma unroll
      for (int x = 1; x < 32; x <<= 1) m = fmaxf(m, __shfl_xor(m, x));
      const float eq = expf(q[u] - m);
      float sm = eq;
#pragma unroll
      for (int x = 1; x < 32; x <<= 1) sm += __shfl_xor(sm, x);
      s[u] = eq / sm;                                         // softmax_d(q)
      ks[u] = expf(k[u] - kmax_j) / ksum_j;                   // softmax_n(k)
    }
    lfdm_wave_lds_sync();                                     // (the previous trip's reads of these rows are done)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      t_go[h][u][j] = g[u];
      t_v[h][u][j] = v[u];
      t_ks[h][u][j] = ks[u];
    }
    lfdm_wave_lds_sync();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float dqs = 0.f, dks = 0.f, dv = 0.f;
#pragma unroll
      for (int e4 = 0; e4 < DH / 4; ++e4) {
        const float4 g4 = *reinterpret_cast<const float4*>(&t_go[h][u][4 * e4]);
        const float4 v4 = *reinterpret_cast<const float4*>(&t_v[h][u][4 * e4]);
        const float4 k4 = *reinterpret_cast<const float4*>(&t_ks[h][u][4 * e4]);
        dqs = fmaf(cj[4 * e4], g4.x, dqs); dqs = fmaf(cj[4 * e4 + 1], g4.y, dqs); dqs = fmaf(cj[4 * e4 + 2], g4.z, dqs); dqs = fmaf(cj[4 * e4 + 3], g4.w, dqs);
        dks = fmaf(dj[4 * e4], v4.x, dks); dks = fmaf(dj[4 * e4 + 1], v4.y, dks); dks = fmaf(dj[4 * e4 + 2], v4.z, dks); dks = fmaf(dj[4 * e4 + 3], v4.w, dks);
        dv = fmaf(dtj[4 * e4], k4.x, dv); dv = fmaf(dtj[4 * e4 + 1], k4.y, dv); dv = fmaf(dtj[4 * e4 + 2], k4.z, dv); dv = fmaf(dtj[4 * e4 + 3], k4.w, dv);
      }
      const float dl = dqs * ATT_SCALE;
      float dot = s[u] * dl;
#pragma unroll
      for (int x = 1; x < 32; x <<= 1) dot += __shfl_xor(dot, x);
      if (ok[u]) {
        float* dst = dqkv + row[u] * QKV_LD + h * DH + j;
        dst[0] = s[u] * (dl - dot);
        dst[OUT_LD] = ks[u] * (dks - rowdot_j);
        dst[2 * OUT_LD] = dv;
      }
    }
  }
}

}  // namespace

extern "C" size_t lfdm_attention_bwd_ws_bytes(int batch, int frames, int hw, int mode) {
  const int L = mode == 0 ? frames : hw;
  const bool rows40 = L > 32 && L <= 40 && attn_bwd_rows40_enabled();      // the 40-frame videos: 40-row tiles (10 instead of 12 k-steps in the token-contraction GEMMs)
  const int wpb = L > 48 ? 2 : 4;
  const int64_t nseq = mode == 0 ? (int64_t)batch * hw : (int64_t)batch * frames;
  const int nb = attn_bwd_blocks(nseq * HEADS, wpb);
  return (size_t)nb * wpb * L * L * sizeof(float);
}

extern "C" int lfdm_attention_bwd_cl_f32(const float* qkv, const float* dout, float* dqkv, int batch, int frames,
                                         int hw, int mode, const float* bias, const float* rot_cos,
                                         const float* rot_sin, float* dbias, void* ws, size_t ws_bytes,
                                         lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int L = mode == 0 ? frames : hw;
  if (!qkv || !dout || !dqkv || batch <= 0 || frames <= 0 || hw <= 0 || (mode != 0 && mode != 1) || L > 64 ||
      ((rot_cos == nullptr) != (rot_sin == nullptr)) || ((bias == nullptr) != (dbias == nullptr))) {
    lfdm_set_error("attention_bwd: unsupported arguments (sequence length must be <= 64; dbias iff bias)");
    return LFDM_EINVAL;
  }
  const bool rows40 = L > 32 && L <= 40 && attn_bwd_rows40_enabled();      // the 40-frame videos: 40-row tiles (10 instead of 12 k-steps in the token-contraction GEMMs)
  const int wpb = L > 48 ? 2 : 4;
  const int64_t nseq = mode == 0 ? (int64_t)batch * hw : (int64_t)batch * frames;
  const int nb = attn_bwd_blocks(nseq * HEADS, wpb);
  float* part = nullptr;
  if (dbias) {
    if (!ws || ws_bytes < lfdm_attention_bwd_ws_bytes(batch, frames, hw, mode)) {
      lfdm_set_error("attention_bwd: workspace too small");
      return LFDM_EWORKSPACE;
    }
    part = (float*)ws;
  }
  const dim3 grid(nb), block(64 * wpb);
  if (L <= 16) LFDM_LAUNCH((attention_bwd_kernel<16, 4>), grid, block, 0, stream, qkv, dout, dqkv, batch, frames, hw, mode, bias, rot_cos, rot_sin, part);
  else if (L <= 32) LFDM_LAUNCH((attention_bwd_kernel<32, 4>), grid, block, 0, stream, qkv, dout, dqkv, batch, frames, hw, mode, bias, rot_cos, rot_sin, part);
  else if (rows40) LFDM_LAUNCH((attention_bwd_kernel<48, 4, 40>), grid, block, 0, stream, qkv, dout, dqkv, batch, frames, hw, mode, bias, rot_cos, rot_sin, part);
  else if (L <= 48) LFDM_LAUNCH((attention_bwd_kernel<48, 4>), grid, block, 0, stream, qkv, dout, dqkv, batch, frames, hw, mode, bias, rot_cos, rot_sin, part);
  else LFDM_LAUNCH((attention_bwd_kernel<64, 2>), grid, block, 0, stream, qkv, dout, dqkv, batch, frames, hw, mode, bias, rot_cos, rot_sin, part);
  int rc = lfdm_check_launch("attention_bwd");
  if (rc) return rc;
  if (dbias) {
    // partial index w = q*8 + head  ->  [q][head][L][L]; sum over q
    return lfdm_sum_leading_f32(part, dbias, (int64_t)HEADS * L * L, nb * wpb / HEADS, stream_);
  }
  return LFDM_OK;
}

extern "C" size_t lfdm_linear_attention_bwd_ws_bytes(int n_frames) {
  return (size_t)n_frames * HEADS * LA_WS * sizeof(float);
}

extern "C" int lfdm_linear_attention_bwd_cl_f32(const float* qkv, const float* dout, float* dqkv, int n_frames,
                                                int hw, void* ws, size_t ws_bytes, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!qkv || !dout || !dqkv || n_frames <= 0 || hw <= 0) {
    lfdm_set_error("linear_attention_bwd: bad arguments");
    return LFDM_EINVAL;
  }
  if (!ws || ws_bytes < lfdm_linear_attention_bwd_ws_bytes(n_frames)) {
    lfdm_set_error("linear_attention_bwd: workspace too small");
    return LFDM_EWORKSPACE;
  }
  LFDM_LAUNCH(linattn_bwd_context_kernel, dim3(n_frames * HEADS), dim3(256), 0, stream, qkv, dout, hw, (float*)ws);
  LFDM_LAUNCH(linattn_bwd_apply_kernel, dim3((hw + LA_TOK - 1) / LA_TOK, n_frames), dim3(256), 0, stream, qkv, dout,
              (const float*)ws, hw, dqkv);
  return lfdm_check_launch("linear_attention_bwd");
}
