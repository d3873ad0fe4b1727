// Attention kernels of the 3D UNet on channels-last rows (include/lfdm_hip.h).
//
// 1. lfdm_attention_cl_f32 - softmax attention over short sequences (temporal: L = T frames per
//    pixel, with rotary + T5 relative-position bias; mid spatial: L = H*W tokens per frame).
//    One wavefront owns one (sequence, head): Q/K/V (L x 32 each) are staged in LDS with
//    coalesced 128-byte row reads (scale and rotary applied on the way in), S = QK^T and O = PV
//    run on v_mfma_f32_16x16x4_f32, the L x L scores never leave registers/LDS
//    (the reference materialises a 126 MB qkv re-layout and a 52 MB score tensor per call,
//    DM/modules/video_flow_diffusion.py:311-361).
// 2. lfdm_linear_attention_cl_f32 - SpatialLinearAttention core (:256-263): k-softmax over
//    tokens + context = k v^T (two-pass column reduction), then q-softmax over the 32-dim axis
//    and out = context^T q.
#include <stdlib.h>
#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

constexpr int HEADS = 8;
constexpr int DH = 32;
constexpr int QKV_LD = 3 * HEADS * DH;  // 768
constexpr int OUT_LD = HEADS * DH;      // 256

// One wavefront (= one 64-thread workgroup) per (sequence, head).  Nothing but P goes through LDS:
//  * the sum over the 32 head features is order independent, so MFMA k-slot `lq` of step s is given feature
//    k = 8*lq + s for BOTH operands: a lane's Q (and K) fragment for a 16-token tile is then 8 CONSECUTIVE floats of one
//    row = two 16-byte global loads straight into the MFMA operand registers (16 rows x 128 B per tile, fully
//    coalesced), with scale and rotary applied in registers (a rotation pair never leaves the lane);
//  * V is consumed as the B operand of P*V in its natural row layout (lane = feature, k-slot = tokens 16*t + 4*lq + r):
//    scalar loads, each element read exactly once;
//  * the scores are computed transposed (S^T = K Q^T), which makes the softmax a register reduction + two shuffles and
//    leaves P^T directly in the A-operand layout of P V: NO LDS and no barrier anywhere in the kernel.
// The first version staged Q, K, V and P in LDS (20 KB per wave, 8 waves per CU, 60 % of the wave cycles in s_waitcnt).
template <int LP>
__global__ __launch_bounds__(64) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int batch,
                                                       int frames, int hw, int mode, const float* __restrict__ bias,
                                                       const float* __restrict__ rot_cos,
                                                       const float* __restrict__ rot_sin) {
  constexpr int NT = LP / 16;

  const int lane = threadIdx.x & 63;
  const int l15 = lane & 15, lq = lane >> 4;
  const int L = mode == 0 ? frames : hw;
  const int64_t unit = blockIdx.x;
  const int64_t seq = unit / HEADS;
  const int head = (int)(unit - seq * HEADS);
  int64_t row0, tstride;
  if (mode == 0) {
    const int64_t b = seq / hw, pix = seq - b * hw;
    row0 = b * frames * hw + pix;
    tstride = hw;
  } else {
    row0 = seq * hw;
    tstride = 1;
  }
  const float scale = 0.17677669529663687f;  // 32^-0.5
  const float* base = qkv + head * DH;

  // ---- Q / K fragments: token tile*16 + l15, features 8*lq .. 8*lq+7 ----
  float qf[NT][8], kf[NT][8];
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    const int t = ti * 16 + l15;
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, k0 = q0, k1 = q0;
    if (t < L) {
      const float* src = base + (row0 + (int64_t)t * tstride) * QKV_LD + 8 * lq;
      q0 = *reinterpret_cast<const float4*>(src);
      q1 = *reinterpret_cast<const float4*>(src + 4);
      k0 = *reinterpret_cast<const float4*>(src + OUT_LD);
      k1 = *reinterpret_cast<const float4*>(src + OUT_LD + 4);
    }
    qf[ti][0] = q0.x * scale; qf[ti][1] = q0.y * scale; qf[ti][2] = q0.z * scale; qf[ti][3] = q0.w * scale;
    qf[ti][4] = q1.x * scale; qf[ti][5] = q1.y * scale; qf[ti][6] = q1.z * scale; qf[ti][7] = q1.w * scale;
    kf[ti][0] = k0.x; kf[ti][1] = k0.y; kf[ti][2] = k0.z; kf[ti][3] = k0.w;
    kf[ti][4] = k1.x; kf[ti][5] = k1.y; kf[ti][6] = k1.z; kf[ti][7] = k1.w;
  }
  // ---- V fragments (issued now, used after the softmax): vf[half][4*tj + r] = v[token 16*tj + 4*lq + r][16*half + l15]
  // = the B operand of P V for the token order t(lq, s) = 16*(s>>2) + 4*lq + (s&3) ----
  float vf[2][4 * NT];
#pragma unroll
  for (int tj = 0; tj < NT; ++tj)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = 16 * tj + 4 * lq + r;
      vf[0][4 * tj + r] = vf[1][4 * tj + r] = 0.f;
      if (t < L) {
        const float* src = base + (row0 + (int64_t)t * tstride) * QKV_LD + 2 * OUT_LD;
        vf[0][4 * tj + r] = src[l15];
        vf[1][4 * tj + r] = src[16 + l15];
      }
    }
  if (rot_cos) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
      const int t = ti * 16 + l15;
      if (t < L) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          const float c = rot_cos[t * 16 + 4 * lq + pr], sn = rot_sin[t * 16 + 4 * lq + pr];
          const float qx = qf[ti][2 * pr], qy = qf[ti][2 * pr + 1];
          const float kx = kf[ti][2 * pr], ky = kf[ti][2 * pr + 1];
          qf[ti][2 * pr] = qx * c - qy * sn;
          qf[ti][2 * pr + 1] = qy * c + qx * sn;
          kf[ti][2 * pr] = kx * c - ky * sn;
          kf[ti][2 * pr + 1] = ky * c + kx * sn;
        }
      }
    }
  }

  // ---- S^T = K Q^T: lane = query token 16*ti + l15, registers = key tokens 16*tj + 4*lq + r.  The softmax over the keys
  // of a query is then a reduction over the lane's registers plus two shuffles (the four k-slots), and the result is
  // already the A operand of P V in the token order above: the scores never touch LDS. ----
  f32x4 st[NT][NT];
#pragma unroll
  for (int ti = 0; ti < NT; ++ti)
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) acc = mfma_16x16x4(kf[tj][s], qf[ti][s], acc);
      st[ti][tj] = acc;
    }
  const bool bias_vec = bias && (L % 4 == 0) && ((((uintptr_t)bias) & 15) == 0);
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    const int qt = ti * 16 + l15;
    float m = -3.0e38f;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      const int key0 = tj * 16 + lq * 4;
      if (bias && qt < L && key0 < L) {
        const float* bp = bias + ((int64_t)head * L + qt) * L + key0;
        if (bias_vec) {
          const float4 b4 = *reinterpret_cast<const float4*>(bp);
          bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[r] = (key0 + r < L) ? bp[r] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = st[ti][tj][r];
        if (key0 + r >= L) v = -3.0e38f;
        else v += bv[r];
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
        const float e = (tj * 16 + lq * 4 + r) < L ? expf(st[ti][tj][r] - m) : 0.f;
        st[ti][tj][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[ti][tj][r] = st[ti][tj][r] / sum;
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
        float* dst = out + (row0 + (int64_t)t * tstride) * OUT_LD + head * DH;
        dst[l15] = o[0][r];
        dst[16 + l15] = o[1][r];
      }
    }
  }
}

// ---------------- linear attention ----------------
// grid (n_frames*8); ctx_out[(f*8+h)][d][e] = sum_n softmax_n(k)[n][d] * v[n][e]
__global__ __launch_bounds__(256) void linattn_context_kernel(const float* __restrict__ qkv, int hw,
                                                              float* __restrict__ ctx_out) {
  __shared__ float red[4][32];
  __shared__ float kmax[32];
  __shared__ float ek[64][33];
  __shared__ __attribute__((aligned(16))) float vv[64][36];
  const int tid = threadIdx.x;
  const int f = blockIdx.x >> 3, h = blockIdx.x & 7;
  const float* kbase = qkv + (int64_t)f * hw * QKV_LD + OUT_LD + h * DH;
  const float* vbase = kbase + OUT_LD;

  {  // pass 1: per-feature max over tokens (float4 per lane, 4 independent rows in flight)
    const int c4 = tid & 7, part = tid >> 3;          // 32 parts x 8 float4 columns
    float4 m = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
    for (int n = part; n < hw; n += 128) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int nn = n + 32 * u;
        v[u] = nn < hw ? *reinterpret_cast<const float4*>(kbase + (int64_t)nn * QKV_LD + 4 * c4) : m;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        m.x = fmaxf(m.x, v[u].x); m.y = fmaxf(m.y, v[u].y); m.z = fmaxf(m.z, v[u].z); m.w = fmaxf(m.w, v[u].w);
      }
    }
    // reduce over the 8 parts that share a wave-local (lane>>3), then across waves through LDS
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
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float ssum = 0.f;
  // pass 2: 64-token tiles, next tile's global loads in flight while this tile is accumulated
  const int ln = tid >> 3, lc4 = tid & 7;              // loader: token ln (+32), float4 column lc4
  float4 rk[2], rv[2];
  auto load_tile = [&](int n0) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int n = n0 + ln + 32 * u;
      rk[u] = rv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < hw) {
        rk[u] = *reinterpret_cast<const float4*>(kbase + (int64_t)n * QKV_LD + 4 * lc4);
        rv[u] = *reinterpret_cast<const float4*>(vbase + (int64_t)n * QKV_LD + 4 * lc4);
      }
    }
  };
  load_tile(0);
  for (int n0 = 0; n0 < hw; n0 += 64) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int nl = ln + 32 * u;
      const bool ok = n0 + nl < hw;
      ek[nl][4 * lc4 + 0] = ok ? expf(rk[u].x - kmax[4 * lc4 + 0]) : 0.f;
      ek[nl][4 * lc4 + 1] = ok ? expf(rk[u].y - kmax[4 * lc4 + 1]) : 0.f;
      ek[nl][4 * lc4 + 2] = ok ? expf(rk[u].z - kmax[4 * lc4 + 2]) : 0.f;
      ek[nl][4 * lc4 + 3] = ok ? expf(rk[u].w - kmax[4 * lc4 + 3]) : 0.f;
      *reinterpret_cast<float4*>(&vv[nl][4 * lc4]) = rv[u];
    }
    __syncthreads();
    if (n0 + 64 < hw) load_tile(n0 + 64);
#pragma unroll 8
    for (int n = 0; n < 64; ++n) {
      const float e = ek[n][d];
      const float4 v4 = *reinterpret_cast<const float4*>(&vv[n][e0]);
      acc[0] = fmaf(e, v4.x, acc[0]);
      acc[1] = fmaf(e, v4.y, acc[1]);
      acc[2] = fmaf(e, v4.z, acc[2]);
      acc[3] = fmaf(e, v4.w, acc[3]);
      ssum += e;
    }
    __syncthreads();
  }
  float* dst = ctx_out + ((int64_t)blockIdx.x * DH + d) * DH + e0;
  dst[0] = acc[0] / ssum;
  dst[1] = acc[1] / ssum;
  dst[2] = acc[2] / ssum;
  dst[3] = acc[3] / ssum;
}

// grid (ceil(hw/32), n_frames); thread = (token, head)
__global__ __launch_bounds__(256) void linattn_output_kernel(const float* __restrict__ qkv,
                                                             const float* __restrict__ ctx, int hw,
                                                             float* __restrict__ out) {
  constexpr int CS = DH * DH + 4;  // padded per-head stride
  __shared__ __attribute__((aligned(16))) float cs[HEADS * CS];
  const int tid = threadIdx.x;
  const int f = blockIdx.y;
  for (int i = tid; i < HEADS * DH * DH; i += 256) {
    const int h = i / (DH * DH), r = i - h * DH * DH;
    cs[h * CS + r] = ctx[(int64_t)f * HEADS * DH * DH + i];
  }
  __syncthreads();
  const int tok = blockIdx.x * 32 + (tid >> 3), h = tid & 7;
  if (tok >= hw) return;
  const int64_t row = (int64_t)f * hw + tok;
  const float* qp = qkv + row * QKV_LD + h * DH;
  float q[DH];
#pragma unroll
  for (int i = 0; i < DH / 4; ++i) {
    const float4 v = reinterpret_cast<const float4*>(qp)[i];
    q[4 * i] = v.x; q[4 * i + 1] = v.y; q[4 * i + 2] = v.z; q[4 * i + 3] = v.w;
  }
  float m = q[0];
#pragma unroll
  for (int i = 1; i < DH; ++i) m = fmaxf(m, q[i]);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < DH; ++i) {
    q[i] = expf(q[i] - m);
    sum += q[i];
  }
  const float scale = 0.17677669529663687f;
#pragma unroll
  for (int i = 0; i < DH; ++i) q[i] = q[i] / sum * scale;
  float* op = out + row * OUT_LD + h * DH;
  const float* ch = cs + h * CS;
#pragma unroll
  for (int e4 = 0; e4 < DH / 4; ++e4) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dd = 0; dd < DH; ++dd) {
      const float4 c = *reinterpret_cast<const float4*>(ch + dd * DH + 4 * e4);
      o.x = fmaf(c.x, q[dd], o.x);
      o.y = fmaf(c.y, q[dd], o.y);
      o.z = fmaf(c.z, q[dd], o.z);
      o.w = fmaf(c.w, q[dd], o.w);
    }
    reinterpret_cast<float4*>(op)[e4] = o;
  }
}

}  // namespace

extern "C" int lfdm_attention_cl_f32(const float* qkv, float* out, int batch, int frames, int hw,
                                     int mode, const float* bias, const float* rot_cos,
                                     const float* rot_sin, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int L = mode == 0 ? frames : hw;
  if (!qkv || !out || batch <= 0 || frames <= 0 || hw <= 0 || (mode != 0 && mode != 1) ||
      L > 64 || ((rot_cos == nullptr) != (rot_sin == nullptr))) {
    lfdm_set_error("attention: unsupported shape (sequence length must be <= 64)");
    return LFDM_EINVAL;
  }
  const int64_t nseq = mode == 0 ? (int64_t)batch * hw : (int64_t)batch * frames;
  const int64_t units = nseq * HEADS;
  const dim3 grid((unsigned)units), block(64);
  if (L <= 16) LFDM_LAUNCH((attention_kernel<16>), grid, block, 0, stream, qkv, out, batch, frames, hw, mode, bias, rot_cos, rot_sin);
  else if (L <= 32) LFDM_LAUNCH((attention_kernel<32>), grid, block, 0, stream, qkv, out, batch, frames, hw, mode, bias, rot_cos, rot_sin);
  else if (L <= 48) LFDM_LAUNCH((attention_kernel<48>), grid, block, 0, stream, qkv, out, batch, frames, hw, mode, bias, rot_cos, rot_sin);
  else LFDM_LAUNCH((attention_kernel<64>), grid, block, 0, stream, qkv, out, batch, frames, hw, mode, bias, rot_cos, rot_sin);
  return lfdm_check_launch("attention");
}

extern "C" size_t lfdm_linear_attention_ws_bytes(int n_frames) {
  return (size_t)n_frames * HEADS * DH * DH * sizeof(float);
}

extern "C" int lfdm_linear_attention_cl_f32(const float* qkv, float* out, int n_frames, int hw,
                                            void* ws, size_t ws_bytes, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!qkv || !out || n_frames <= 0 || hw <= 0) {
    lfdm_set_error("linear_attention: bad arguments");
    return LFDM_EINVAL;
  }
  if (!ws || ws_bytes < lfdm_linear_attention_ws_bytes(n_frames)) {
    lfdm_set_error("linear_attention: workspace too small");
    return LFDM_EWORKSPACE;
  }
  float* ctx = reinterpret_cast<float*>(ws);
  LFDM_LAUNCH(linattn_context_kernel, dim3(n_frames * HEADS), dim3(256), 0, stream, qkv, hw, ctx);
  LFDM_LAUNCH(linattn_output_kernel, dim3((hw + 31) / 32, n_frames), dim3(256), 0, stream, qkv,
              (const float*)ctx, hw, out);
  return lfdm_check_launch("linear_attention");
}
