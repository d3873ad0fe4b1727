// Normalisation / element-wise kernels on channels-last rows (HBM-bound):
//   GroupNorm(8) over (C/8, T, H, W) + scale/shift + SiLU   (reference Block.forward :200-211)
//   channel LayerNorm (gamma only)                          (reference LayerNorm :170-179)
//   per-channel affine + activation, 2x2 average pool, planar<->channels-last transposes (LFAE).
// GroupNorm statistics span all frames of a sample, so they are a two-stage reduction:
// per-workgroup partial (sum, sumsq) via wave shuffles -> a tiny finalize kernel that merges the
// partials in double and folds mean/rstd/gamma/beta/scale/shift into one per-(b,c) FMA ->
// a streaming apply pass.  Reduction order is fixed (no float atomics): bit-reproducible.
#include <stdlib.h>
#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

constexpr int GN_MAX_CHUNKS = 256;

__host__ __device__ inline int gn_num_chunks(int pixels) {
  // >= 16 pixels per workgroup; small tensors (low-resolution UNet levels) still get tens of
  // workgroups instead of a handful of long serial loops
  int n = (pixels + 15) / 16;
  if (n < 1) n = 1;
  if (n > GN_MAX_CHUNKS) n = GN_MAX_CHUNKS;
  return n;
}

// grid (nchunk, B), 256 threads.  partial[b][chunk][g] = (sum, sumsq)
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, int pixels,
                                                         int channels, int groups,
                                                         float* __restrict__ partial) {
  __shared__ float red_s[256], red_q[256];
  const int tid = threadIdx.x;
  const int c4n = channels >> 2;
  const int rows_per_iter = 256 / c4n;
  const int c4 = tid % c4n, prow = tid / c4n;
  const int nchunk = gridDim.x, chunk = blockIdx.x, b = blockIdx.y;
  const int per = (pixels + nchunk - 1) / nchunk;
  const int p0 = chunk * per;
  const int p1 = (p0 + per < pixels) ? p0 + per : pixels;
  const float* xb = x + (int64_t)b * pixels * channels;
  float s = 0.f, q = 0.f;
  if (prow < rows_per_iter) {
    for (int p = p0 + prow; p < p1; p += rows_per_iter) {
      const float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)p * channels + 4 * c4);
      s += (v.x + v.y) + (v.z + v.w);
      q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  }
  red_s[tid] = s;
  red_q[tid] = q;
  __syncthreads();
  if (tid < groups) {
    const int cg4 = c4n / groups;  // float4 per group per pixel
    float ts = 0.f, tq = 0.f;
    for (int r = 0; r < rows_per_iter; ++r)
      for (int k = 0; k < cg4; ++k) {
        const int t = r * c4n + tid * cg4 + k;
        ts += red_s[t];
        tq += red_q[t];
      }
    float* dst = partial + (((int64_t)b * nchunk + chunk) * groups + tid) * 2;
    dst[0] = ts;
    dst[1] = tq;
  }
}

// Finalize + apply in one launch.  grid (blocks_per_sample, B).  Every workgroup first merges the
// (sum, sumsq) partials of its sample in double (fixed order -> bit reproducible) and folds
// mean / rstd / gamma / beta / (scale+1, shift) into one FMA per channel kept in LDS, then streams its
// slice:  y = silu(x * A[c] + Bc[c]) (+ residual).
constexpr int GN_MAX_C = 1024;
// NT threads per workgroup (256 / 512 / 1024).  Every workgroup repeats the merge of the partials, so fewer, larger workgroups
// (LFDM_GN_BLOCK, LFDM_GN_F4) trade memory-level parallelism for less redundant L2 traffic: with 320 partial chunks x 16
// groups (the merged output heads) the 2560 256-thread workgroups of round 2 read 100 MB of partials for a 21 MB tensor.
template <int NT>
__global__ __launch_bounds__(NT) void gn_apply_kernel(
    const float* __restrict__ x, float* __restrict__ out, int pixels, int channels, int groups,
    const float* __restrict__ partial, int nchunk, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ scale_shift, int ss_ld, float eps, int silu,
    const float* __restrict__ residual, int xcd_r) {
  __shared__ float s_mean[64], s_rstd[64];
  __shared__ __attribute__((aligned(16))) float s_a[GN_MAX_C], s_b[GN_MAX_C];
  const int b = blockIdx.y, tid = threadIdx.x;
  // XCD affinity (round 4).  Workgroups are dealt round-robin to the 8 XCDs in linear-id order, and the convolution that produced x
  // (and the one that will read the result) runs the workgroup of 128-pixel tile block t on XCD t % 8 - its output is still in THAT L2
  // (written back at the kernel boundary, not invalidated).  With xcd_r = R > 0 (R runs of NT float4 per tile block, a power of two,
  // gridDim.x % 8R == 0 - launch_gn_apply checks) the runs are re-dealt so that the workgroup on XCD v walks runs of tile blocks = v
  // (mod 8): same-XCD reads instead of cross-XCD ones (MI355X_MICROARCH.md "handoff-payload": 1.7x).  Speed only.
  unsigned jb = blockIdx.x;
  if (xcd_r > 0) {
    const unsigned r8 = 8u * (unsigned)xcd_r, q = jb / r8, rem = jb - q * r8;
    jb = q * r8 + (rem & 7u) * (unsigned)xcd_r + (rem >> 3);
  }
  // the first two float4 of this thread are requested BEFORE the statistics are merged: their HBM latency runs under the
  // prologue instead of after it
  const int c4n = channels >> 2;
  const int64_t per_b = (int64_t)pixels * c4n;
  const float4* xb = reinterpret_cast<const float4*>(x) + (int64_t)b * per_b;
  float4* ob = reinterpret_cast<float4*>(out) + (int64_t)b * per_b;
  const float4* rb = residual ? reinterpret_cast<const float4*>(residual) + (int64_t)b * per_b : nullptr;
  const int64_t stride = (int64_t)gridDim.x * NT;
  const int64_t i0 = (int64_t)jb * NT + tid;
  float4 pre_v[2], pre_r[2];        // (four in flight measured 1.6x SLOWER: the selects below stop being register renames)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int64_t i = i0 + k * stride;
    pre_v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    pre_r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < per_b) {
      pre_v[k] = xb[i];
      if (rb) pre_r[k] = rb[i];
    }
  }
  // ... and so are this thread's per-channel parameters (round 6): read behind the merge's barrier they were one more dependent round trip of every launch
  constexpr int PREC = 2;            // channels tid, tid + NT (C <= 512 with 256 threads); further ones are read in place
  float p_g[PREC], p_b[PREC], p_sc[PREC], p_sh[PREC];
#pragma unroll
  for (int j = 0; j < PREC; ++j) {
    const int c = tid + j * NT, cc = c < channels ? c : 0;
    p_g[j] = gamma[cc];
    p_b[j] = beta[cc];
    p_sc[j] = scale_shift ? scale_shift[(int64_t)b * ss_ld + cc] : 0.f;
    p_sh[j] = scale_shift ? scale_shift[(int64_t)b * ss_ld + channels + cc] : 0.f;
  }
  // (a coalesced float2 walk with one thread per (chunk, group) item + an LDS tree was measured: 9.2 vs 9.1 us at 320 chunks and
  // 8.0 vs 6.8 us at 32 - the extra barrier costs more than the strided loads; removed)
  // LPG lanes walk one group's chunks: a whole wavefront per group once the workgroup has 64 threads per group
  const int lpg = (NT >= 64 * groups) ? 64 : 32;
  const int gpp = NT / lpg;           // groups per pass
  const int sub = tid & (lpg - 1);
  for (int g0 = 0; g0 < groups; g0 += gpp) {
    const int g = g0 + tid / lpg;
    double s = 0.0, q = 0.0;
    if (g < groups) {
      // four chunk loads in flight per lane (round 4: the one-load-per-trip walk paid an L2 round trip per chunk - 2.4 us of the launch
      // at 320 chunks, tools/ubench/launch_floor.py); same summation order as before: k ascending per lane.
      // Round 6, measured and NOT kept: all of a lane's loads in flight at once (sixteen 64-bit-addressed loads: 56 -> 135 VGPRs = three
      // waves per SIMD, +3.6 ms per video, profiles/r06_b_gn_merge_ab.txt) and eight per trip through a buffer descriptor with no single-
      // load tail (64 VGPRs, still +1.4 ms per video, profiles/r06_c_gn_merge8_ab.txt): in situ the walk is not what the launch waits for.
      const float2* src = reinterpret_cast<const float2*>(partial) + ((int64_t)b * nchunk) * groups + g;
      int k = sub;
      for (; k + 3 * lpg < nchunk; k += 4 * lpg) {
        const float2 v0 = src[(int64_t)k * groups], v1 = src[(int64_t)(k + lpg) * groups];
        const float2 v2 = src[(int64_t)(k + 2 * lpg) * groups], v3 = src[(int64_t)(k + 3 * lpg) * groups];
        s += (double)v0.x; q += (double)v0.y;
        s += (double)v1.x; q += (double)v1.y;
        s += (double)v2.x; q += (double)v2.y;
        s += (double)v3.x; q += (double)v3.y;
      }
      for (; k < nchunk; k += lpg) {
        const float2 v = src[(int64_t)k * groups];
        s += (double)v.x;
        q += (double)v.y;
      }
    }
    for (int m = lpg >> 1; m >= 1; m >>= 1) {
      s += __shfl_xor(s, m);
      q += __shfl_xor(q, m);
    }
    if (g < groups && sub == 0) {
      const double n = (double)pixels * (double)(channels / groups);
      const double mean = s / n;
      double var = q / n - mean * mean;
      if (var < 0.0) var = 0.0;
      s_mean[g] = (float)mean;
      s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
  const int cg = channels / groups;
  {
    int j = 0;
    for (int c = tid; c < channels; c += NT, ++j) {
      const int g = c / cg;
      const bool pre = j < PREC;
      const float a = s_rstd[g] * (pre ? (j == 0 ? p_g[0] : p_g[1]) : gamma[c]);
      float bb = (pre ? (j == 0 ? p_b[0] : p_b[1]) : beta[c]) - s_mean[g] * a;
      float aa = a;
      if (scale_shift) {
        const float sc = (pre ? (j == 0 ? p_sc[0] : p_sc[1]) : scale_shift[(int64_t)b * ss_ld + c]) + 1.0f;
        const float sh = pre ? (j == 0 ? p_sh[0] : p_sh[1]) : scale_shift[(int64_t)b * ss_ld + channels + c];
        aa = a * sc;
        bb = bb * sc + sh;
      }
      s_a[c] = aa;
      s_b[c] = bb;
    }
  }
  __syncthreads();
  int k = 0;
  for (int64_t i = i0; i < per_b; i += stride, ++k) {
    const int c4 = (int)(i % c4n);
    const float4 v = k == 0 ? pre_v[0] : k == 1 ? pre_v[1] : xb[i];
    const float4 a = *reinterpret_cast<const float4*>(s_a + 4 * c4);
    const float4 d = *reinterpret_cast<const float4*>(s_b + 4 * c4);
    float4 y;
    y.x = fmaf(v.x, a.x, d.x);
    y.y = fmaf(v.y, a.y, d.y);
    y.z = fmaf(v.z, a.z, d.z);
    y.w = fmaf(v.w, a.w, d.w);
    if (silu) {
      y.x = siluf_(y.x);
      y.y = siluf_(y.y);
      y.z = siluf_(y.z);
      y.w = siluf_(y.w);
    }
    if (rb) {
      const float4 r = k == 0 ? pre_r[0] : k == 1 ? pre_r[1] : rb[i];
      y.x += r.x;
      y.y += r.y;
      y.z += r.z;
      y.w += r.w;
    }
    ob[i] = y;
  }
}

// C = 64 / 128 (the two finest levels of the training forward): a wavefront serves 64 / C4N rows at once - lane = (row of the group, float4
// column), the row reductions are xor-shuffles inside C4N adjacent lanes - and two row groups are in flight per trip (one row per wavefront
// leaves 48 of 64 lanes idle at C = 64: train_norm.hip, layernorm_bwd_small_kernel).  Same arithmetic as the kernel below.
template <int C4N>
__global__ __launch_bounds__(256) void layernorm_small_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t rows,
                                                              const float* __restrict__ gamma, float eps) {
  constexpr int RPW = 64 / C4N, C = 4 * C4N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / C4N, f = lane % C4N;
  const float4 g = reinterpret_cast<const float4*>(gamma)[f];
  auto group_sum = [&](float v) {
#pragma unroll
    for (int m = 1; m < C4N; m <<= 1) v += __shfl_xor(v, m);
    return v;
  };
  const int64_t stride = (int64_t)gridDim.x * 4 * RPW * 2;
  for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * RPW * 2; row0 < rows; row0 += stride) {
    float4 v[2];
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t row = row0 + u * RPW + sub;
      live[u] = row < rows;
      v[u] = live[u] ? reinterpret_cast<const float4*>(x + row * C)[f] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float mean = group_sum((v[u].x + v[u].y) + (v[u].z + v[u].w)) / (float)C;
      const float a = v[u].x - mean, b = v[u].y - mean, c = v[u].z - mean, d = v[u].w - mean;
      const float var = group_sum((a * a + b * b) + (c * c + d * d)) / (float)C;
      const float denom = sqrtf(var + eps);
      if (live[u]) {
        float4 y;
        y.x = a / denom * g.x;
        y.y = b / denom * g.y;
        y.z = c / denom * g.z;
        y.w = d / denom * g.w;
        reinterpret_cast<float4*>(out + (row0 + u * RPW + sub) * C)[f] = y;
      }
    }
  }
}

// one wavefront per row, 4 rows per workgroup; C % 4 == 0, C <= 1024
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                        float* __restrict__ out, int64_t rows,
                                                        int channels, const float* __restrict__ gamma,
                                                        float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4n = channels >> 2;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * channels);
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = lane + 64 * i;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < c4n) {
        v[i] = xr[f];
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
    }
    const float mean = wave_sum(s) / (float)channels;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = lane + 64 * i;
      if (f < c4n) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
      }
    }
    const float var = wave_sum(q) / (float)channels;
    const float denom = sqrtf(var + eps);
    float4* orow = reinterpret_cast<float4*>(out + row * channels);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = lane + 64 * i;
      if (f < c4n) {
        const float4 g = reinterpret_cast<const float4*>(gamma)[f];
        float4 y;
        y.x = (v[i].x - mean) / denom * g.x;
        y.y = (v[i].y - mean) / denom * g.y;
        y.z = (v[i].z - mean) / denom * g.z;
        y.w = (v[i].w - mean) / denom * g.w;
        orow[f] = y;
      }
    }
  }
}

__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x,
                                                         float* __restrict__ out, int64_t rows,
                                                         int channels, int ldx, int ldo,
                                                         const float* __restrict__ a,
                                                         const float* __restrict__ b, int act) {
  const int c4n = channels >> 2;
  const int64_t total = rows * c4n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
    const float4 aa = *reinterpret_cast<const float4*>(a + c);
    const float4 bb = *reinterpret_cast<const float4*>(b + c);
    float4 y;
    y.x = apply_act(fmaf(v.x, aa.x, bb.x), act);
    y.y = apply_act(fmaf(v.y, aa.y, bb.y), act);
    y.z = apply_act(fmaf(v.z, aa.z, bb.z), act);
    y.w = apply_act(fmaf(v.w, aa.w, bb.w), act);
    *reinterpret_cast<float4*>(out + r * ldo + c) = y;
  }
}

__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ x,
                                                       float* __restrict__ out, int n_img, int h,
                                                       int w, int channels) {
  const int c4n = channels >> 2;
  const int ho = h >> 1, wo = w >> 1;
  const int64_t total = (int64_t)n_img * ho * wo * c4n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % c4n) * 4;
    int64_t pix = i / c4n;
    const int ox = (int)(pix % wo);
    pix /= wo;
    const int oy = (int)(pix % ho);
    const int n = (int)(pix / ho);
    const float* base = x + (((int64_t)n * h + 2 * oy) * w + 2 * ox) * channels + c;
    const float4 v00 = *reinterpret_cast<const float4*>(base);
    const float4 v01 = *reinterpret_cast<const float4*>(base + channels);
    const float4 v10 = *reinterpret_cast<const float4*>(base + (int64_t)w * channels);
    const float4 v11 = *reinterpret_cast<const float4*>(base + (int64_t)w * channels + channels);
    float4 y;
    y.x = ((v00.x + v01.x) + (v10.x + v11.x)) * 0.25f;
    y.y = ((v00.y + v01.y) + (v10.y + v11.y)) * 0.25f;
    y.z = ((v00.z + v01.z) + (v10.z + v11.z)) * 0.25f;
    y.w = ((v00.w + v01.w) + (v10.w + v11.w)) * 0.25f;
    *reinterpret_cast<float4*>(out + (((int64_t)n * ho + oy) * wo + ox) * channels + c) = y;
  }
}

// (n_img, C, hw) planar -> rows (n_img*hw, ldo).  32x32 LDS tile transpose; grid (hw/32, C/32, n)
__global__ __launch_bounds__(256) void planar_to_cl_kernel(const float* __restrict__ x,
                                                           float* __restrict__ out, int channels,
                                                           int hw, int ldo) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, p = p0 + tx;
    tile[i][tx] = (c < channels && p < hw) ? x[((int64_t)n * channels + c) * hw + p] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int p = p0 + i, c = c0 + tx;
    if (p < hw && c < channels) out[((int64_t)n * hw + p) * ldo + c] = tile[tx][i];
  }
}

__global__ __launch_bounds__(256) void cl_to_planar_kernel(const float* __restrict__ x,
                                                           float* __restrict__ out, int channels,
                                                           int hw, int ldx) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int p = p0 + i, c = c0 + tx;
    tile[i][tx] = (p < hw && c < channels) ? x[((int64_t)n * hw + p) * ldx + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, p = p0 + tx;
    if (c < channels && p < hw) out[((int64_t)n * channels + c) * hw + p] = tile[tx][i];
  }
}

inline unsigned grid_for(int64_t work_items, unsigned cap = 8192) {
  int64_t nb = (work_items + 255) / 256;
  if (nb < 1) nb = 1;
  if (nb > cap) nb = cap;
  return (unsigned)nb;
}

void launch_gn_apply(const float* x, float* out, int batch, int pixels, int channels, int groups,
                     const float* partial, int nchunk, const float* gamma, const float* beta,
                     const float* scale_shift, int ss_ld, float eps, int silu, const float* residual,
                     hipStream_t stream) {
  const int64_t per_b = (int64_t)pixels * (channels / 4);
  // float4 per thread and threads per workgroup: two per thread in 256-thread workgroups measured best in rounds 1 and 2 for the
  // 64 ... 512-channel tensors with <= 8 groups; LFDM_GN_F4 / LFDM_GN_BLOCK override (tools/bench_gn.py sweeps them)
  static const int env_f4 = [] { const char* e = lfdm_knob("LFDM_GN_F4"); return e ? atoi(e) : 0; }();
  static const int env_nt = [] { const char* e = lfdm_knob("LFDM_GN_BLOCK"); return e ? atoi(e) : 0; }();
  int f4 = 2, nt = 256;
  // many partials: merge them in fewer, larger workgroups (tools/bench_gn.py sweep, profiles/r03_a_sweep_gn.txt: the merged output
  // heads' 21 MB tensor with 320 x 16 partials 33.1 -> 16.8 us at 1024 threads x 8 float4; 640 x 8 partials on 2.6 MB 6.5 -> 5.6 us at 512)
  if ((int64_t)nchunk * groups >= 4096) {
    nt = per_b >= (1 << 20) ? 1024 : 512;
    f4 = per_b >= (1 << 20) ? 8 : 2;
  }
  if (env_f4 > 0) f4 = env_f4;
  if (env_nt == 256 || env_nt == 512 || env_nt == 1024) nt = env_nt;
  int64_t nb = (per_b + (int64_t)nt * f4 - 1) / ((int64_t)nt * f4);
  if (nb < 1) nb = 1;
  if (nb > 8192) nb = 8192;        // per sample (grid y = batch); the 2C-channel output heads need 5120 at 32x32 x 40 frames
  const dim3 grid((unsigned)nb, batch);
  // XCD-affine run order (see the kernel): R = runs of nt float4 per 128-pixel tile block, a power of two; whole groups of 8R runs only
  int xcd_r = 0;
  {
    const char* e = lfdm_knob("LFDM_GN_XCD");      // (read per call: A/B in tools)
    const int64_t blk_f4 = 128ll * (channels / 4);
    if (!(e && e[0] == '0') && blk_f4 % nt == 0) {
      const int64_t r = blk_f4 / nt;
      if (r >= 1 && r <= 64 && (r & (r - 1)) == 0 && nb % (8 * r) == 0 && (per_b % ((int64_t)nt)) == 0) xcd_r = (int)r;
    }
  }
  if (nt == 1024)
    LFDM_LAUNCH(gn_apply_kernel<1024>, grid, dim3(1024), 0, stream, x, out, pixels, channels, groups, partial, nchunk, gamma, beta,
                scale_shift, ss_ld, eps, silu, residual, xcd_r);
  else if (nt == 512)
    LFDM_LAUNCH(gn_apply_kernel<512>, grid, dim3(512), 0, stream, x, out, pixels, channels, groups, partial, nchunk, gamma, beta,
                scale_shift, ss_ld, eps, silu, residual, xcd_r);
  else
    LFDM_LAUNCH(gn_apply_kernel<256>, grid, dim3(256), 0, stream, x, out, pixels, channels, groups, partial, nchunk, gamma, beta,
                scale_shift, ss_ld, eps, silu, residual, xcd_r);
}


// Built, measured and REMOVED in round 6 (records: HISTORY.md rounds 3-4): GroupNorm straight from the raw split-K slabs (one workgroup per
// (group, sample): 8 workgroups pull 1 MB slabs at a tenth of the HBM rate) and the chip-wide cooperative reduce + statistics + apply launch
// (+6.4 us per block against conv + reduce + apply, profiles/r04_p_gn_coop_ab.json).

}  // namespace

extern "C" size_t lfdm_groupnorm_ws_bytes(int batch, int pixels, int channels) {
  const size_t partial = (size_t)batch * gn_num_chunks(pixels) * 64 * 2;
  const size_t ab = (size_t)batch * 2 * channels;
  return (partial + ab) * sizeof(float);
}

extern "C" int lfdm_groupnorm_silu_cl_f32(const float* x, float* out, int batch, int pixels,
                                          int channels, int groups, const float* gamma,
                                          const float* beta, const float* scale_shift, int ss_ld,
                                          const float* residual, float eps, int apply_silu, void* ws,
                                          size_t ws_bytes, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !out || !gamma || !beta || batch <= 0 || pixels <= 0 || channels <= 0 || groups <= 0 ||
      groups > 64 || (scale_shift && ss_ld < 2 * channels) || channels % groups != 0 || (channels / groups) % 4 != 0 ||
      256 % (channels / 4) != 0) {
    lfdm_set_error("groupnorm: unsupported shape (need C%G==0, (C/G)%4==0, 256%(C/4)==0)");
    return LFDM_EINVAL;
  }
  if (!ws || ws_bytes < lfdm_groupnorm_ws_bytes(batch, pixels, channels)) {
    lfdm_set_error("groupnorm: workspace too small");
    return LFDM_EWORKSPACE;
  }
  const int nchunk = gn_num_chunks(pixels);
  float* partial = reinterpret_cast<float*>(ws);
  LFDM_LAUNCH(gn_partial_kernel, dim3(nchunk, batch), dim3(256), 0, stream, x, pixels, channels,
              groups, partial);
  launch_gn_apply(x, out, batch, pixels, channels, groups, (const float*)partial, nchunk, gamma, beta,
                  scale_shift, ss_ld, eps, apply_silu, residual, stream);
  return lfdm_check_launch("groupnorm");
}

extern "C" int lfdm_groupnorm_apply_cl_f32(const float* x, float* out, int batch, int pixels,
                                           int channels, int groups, const float* gamma,
                                           const float* beta, const float* scale_shift, int ss_ld,
                                           const float* residual, float eps, int apply_silu,
                                           const float* partial, int nchunk, void* ws,
                                           size_t ws_bytes, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !out || !gamma || !beta || !partial || nchunk <= 0 || batch <= 0 || pixels <= 0 ||
      channels <= 0 || channels > GN_MAX_C || groups <= 0 || groups > 64 || channels % groups != 0 ||
      channels % 4 != 0 || (scale_shift && ss_ld < 2 * channels)) {
    lfdm_set_error("groupnorm_apply: bad arguments");
    return LFDM_EINVAL;
  }
  (void)ws;
  (void)ws_bytes;
  launch_gn_apply(x, out, batch, pixels, channels, groups, partial, nchunk, gamma, beta, scale_shift,
                  ss_ld, eps, apply_silu, residual, stream);
  return lfdm_check_launch("groupnorm_apply");
}

extern "C" int lfdm_layernorm_cl_f32(const float* x, float* out, int64_t rows, int channels,
                                     const float* gamma, float eps, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !out || !gamma || rows <= 0 || channels <= 0 || channels % 4 != 0 || channels > 1024) {
    lfdm_set_error("layernorm: unsupported shape (C%4==0, C<=1024)");
    return LFDM_EINVAL;
  }
  int64_t nb = (rows + 3) / 4;
  if (nb > 16384) nb = 16384;
  const bool al16 = ((((uintptr_t)x) | ((uintptr_t)out) | ((uintptr_t)gamma)) & 15) == 0;
  if ((channels == 64 || channels == 128) && al16 && rows >= 4096) {
    int64_t nbs = (rows + 4 * (256 / channels) * 2 * 4 - 1) / (4 * (256 / channels) * 2 * 4);      // ~4 trips per wavefront
    if (nbs > 8192) nbs = 8192;
    if (nbs < 1) nbs = 1;
    if (channels == 64) LFDM_LAUNCH((layernorm_small_kernel<16>), dim3((unsigned)nbs), dim3(256), 0, stream, x, out, rows, gamma, eps);
    else LFDM_LAUNCH((layernorm_small_kernel<32>), dim3((unsigned)nbs), dim3(256), 0, stream, x, out, rows, gamma, eps);
    return lfdm_check_launch("layernorm");
  }
  LFDM_LAUNCH(layernorm_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, out, rows, channels,
              gamma, eps);
  return lfdm_check_launch("layernorm");
}

extern "C" int lfdm_affine_act_cl_f32(const float* x, float* out, int64_t rows, int channels,
                                      int ldx, int ldo, const float* a, const float* b, int act,
                                      lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !out || !a || !b || rows <= 0 || channels % 4 != 0 || ldx % 4 != 0 || ldo % 4 != 0) {
    lfdm_set_error("affine_act: unsupported shape");
    return LFDM_EINVAL;
  }
  LFDM_LAUNCH(affine_act_kernel, dim3(grid_for(rows * (channels / 4))), dim3(256), 0, stream, x, out,
              rows, channels, ldx, ldo, a, b, act);
  return lfdm_check_launch("affine_act");
}

extern "C" int lfdm_avgpool2_cl_f32(const float* x, float* out, int n_img, int h, int w,
                                    int channels, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !out || n_img <= 0 || (h & 1) || (w & 1) || channels % 4 != 0) {
    lfdm_set_error("avgpool2: unsupported shape");
    return LFDM_EINVAL;
  }
  const int64_t total = (int64_t)n_img * (h / 2) * (w / 2) * (channels / 4);
  LFDM_LAUNCH(avgpool2_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, out, n_img, h, w,
              channels);
  return lfdm_check_launch("avgpool2");
}

extern "C" int lfdm_planar_to_cl_f32(const float* x, float* out, int n_img, int channels, int hw,
                                     int ldo, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !out || n_img <= 0 || channels <= 0 || hw <= 0 || ldo < channels) {
    lfdm_set_error("planar_to_cl: bad arguments");
    return LFDM_EINVAL;
  }
  LFDM_LAUNCH(planar_to_cl_kernel, dim3((hw + 31) / 32, (channels + 31) / 32, n_img), dim3(256), 0,
              stream, x, out, channels, hw, ldo);
  return lfdm_check_launch("planar_to_cl");
}

extern "C" int lfdm_cl_to_planar_f32(const float* x, float* out, int n_img, int channels, int hw,
                                     int ldx, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !out || n_img <= 0 || channels <= 0 || hw <= 0 || ldx < channels) {
    lfdm_set_error("cl_to_planar: bad arguments");
    return LFDM_EINVAL;
  }
  LFDM_LAUNCH(cl_to_planar_kernel, dim3((hw + 31) / 32, (channels + 31) / 32, n_img), dim3(256), 0,
              stream, x, out, channels, hw, ldx);
  return lfdm_check_launch("cl_to_planar");
}
