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
  for (int c = tid; c < channels; c += NT) {
    const int g = c / cg;
    const float a = s_rstd[g] * gamma[c];
    float bb = beta[c] - s_mean[g] * a;
    float aa = a;
    if (scale_shift) {
      const float sc = scale_shift[(int64_t)b * ss_ld + c] + 1.0f;
      const float sh = scale_shift[(int64_t)b * ss_ld + channels + c];
      aa = a * sc;
      bb = bb * sc + sh;
    }
    s_a[c] = aa;
    s_b[c] = bb;
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
  static const int env_f4 = [] { const char* e = getenv("LFDM_GN_F4"); return e ? atoi(e) : 0; }();
  static const int env_nt = [] { const char* e = getenv("LFDM_GN_BLOCK"); return e ? atoi(e) : 0; }();
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
    const char* e = getenv("LFDM_GN_XCD");      // (read per call: A/B in tools)
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


// GroupNorm straight from split-K slabs: one workgroup per (group, sample) sums the slabs of its rows x channels into registers, reduces the
// statistics (double), and writes y = silu(x * A[c] + B[c]) (+ residual).  1024 threads, GSK_MAX float4 per thread.
constexpr int GSK_MAX = 20;
__global__ __launch_bounds__(1024) void gn_splitk_apply_kernel(const float* __restrict__ partial, int ksplit, int64_t slab_stride, int coutp,
                                                               const float* __restrict__ bias, float* __restrict__ out, int pixels,
                                                               int channels, int groups, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float* __restrict__ scale_shift,
                                                               int ss_ld, const float* __restrict__ residual, float eps, int silu) {
  __shared__ double red_s[16], red_q[16];
  __shared__ float s_stat[2];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int cg = channels / groups, cg4 = cg >> 2;              // 1024 % cg4 == 0: a thread keeps its channel quad
  const int items = pixels * cg4;
  const int q = tid % cg4, c0 = g * cg + 4 * q;
  float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) bb = *reinterpret_cast<const float4*>(bias + c0);
  float4 v[GSK_MAX];
  float ls = 0.f, lq = 0.f;
#pragma unroll
  for (int it = 0; it < GSK_MAX; ++it) {
    const int i = tid + it * 1024;
    v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < items) {
      const int64_t row = (int64_t)b * pixels + i / cg4;
      const float* src = partial + row * coutp + c0;
      float4 a = bb;
      for (int z = 0; z < ksplit; ++z) {
        const float4 t = *reinterpret_cast<const float4*>(src + (int64_t)z * slab_stride);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
      }
      v[it] = a;
      ls += (a.x + a.y) + (a.z + a.w);
      lq += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
    }
  }
  double ds = (double)ls, dq = (double)lq;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    ds += __shfl_xor(ds, m);
    dq += __shfl_xor(dq, m);
  }
  if ((tid & 63) == 0) {
    red_s[tid >> 6] = ds;
    red_q[tid >> 6] = dq;
  }
  __syncthreads();
  if (tid == 0) {
    double ts = 0.0, tq = 0.0;
    for (int w = 0; w < 16; ++w) {
      ts += red_s[w];
      tq += red_q[w];
    }
    const double n = (double)pixels * (double)cg;
    const double mean = ts / n;
    double var = tq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    s_stat[0] = (float)mean;
    s_stat[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const float mean = s_stat[0], rstd = s_stat[1];
  float aa[4], ab[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = c0 + e;
    const float a = rstd * gamma[c];
    float bsh = beta[c] - mean * a, asc = a;
    if (scale_shift) {
      const float sc = scale_shift[(int64_t)b * ss_ld + c] + 1.0f;
      const float sh = scale_shift[(int64_t)b * ss_ld + channels + c];
      asc = a * sc;
      bsh = bsh * sc + sh;
    }
    aa[e] = asc;
    ab[e] = bsh;
  }
#pragma unroll
  for (int it = 0; it < GSK_MAX; ++it) {
    const int i = tid + it * 1024;
    if (i < items) {
      const int64_t row = (int64_t)b * pixels + i / cg4;
      float4 y;
      y.x = fmaf(v[it].x, aa[0], ab[0]);
      y.y = fmaf(v[it].y, aa[1], ab[1]);
      y.z = fmaf(v[it].z, aa[2], ab[2]);
      y.w = fmaf(v[it].w, aa[3], ab[3]);
      if (silu) {
        y.x = siluf_(y.x);
        y.y = siluf_(y.y);
        y.z = siluf_(y.z);
        y.w = siluf_(y.w);
      }
      if (residual) {
        const float4 r = *reinterpret_cast<const float4*>(residual + row * channels + c0);
        y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
      }
      *reinterpret_cast<float4*>(out + row * channels + c0) = y;
    }
  }
}


// ---- split-K reduce + GroupNorm statistics + apply in ONE launch, chip wide (round 4) ----
// The low-resolution ResnetBlocks of a B = 1 step ran conv (split-K slabs) -> reduce pass (7 us) -> GroupNorm apply (5 us): two launches
// that move a 1.3 MB tensor at a tenth of the HBM rate.  The one-workgroup-per-group form above has 8 workgroups pull the slabs (slower);
// here the reduce pass's own grid - one workgroup per 16 rows x 64 channels, one float4 per thread - keeps its reduced values in
// REGISTERS, publishes its (sum, sum of squares) of every group it touches as ONE 8-byte agent-scope granule, counts itself in on the
// group's arrival counter, waits until all `need` workgroups of the (sample, group) have arrived (relaxed polls, bounded), sums the
// granules in a fixed order (double: bit reproducible) and applies the normalisation to the values it still holds.  No fences: every
// shared word is an 8- / 4-byte agent-scope atomic on both sides.  The last workgroup to LEAVE a group zeroes its two counters, so a
// workspace zeroed once serves every later launch.  Correct for any placement; needs all workgroups co-resident (the launcher admits at
// most 1024 = four per CU) - a timeout (2^20 polls) sets ws[0] and lets the launch finish rather than hang.
constexpr int COOP_ROWS = 16, COOP_C4 = 16;       // 16 rows x 16 float4 columns = 256 threads, one item each
template <int KS>
__global__ __launch_bounds__(256) void gn_splitk_coop_kernel(const float* __restrict__ partial, int64_t slab_stride, int coutp,
                                                             const float* __restrict__ bias, float* __restrict__ out, int pixels, int channels,
                                                             int groups, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ scale_shift, int ss_ld, const float* __restrict__ residual,
                                                             float eps, int silu, unsigned* __restrict__ ws) {
  __shared__ float s_ps[4][8], s_pq[4][8];        // per wave, per local group
  __shared__ float s_mean[8], s_rstd[8];
  __shared__ int s_ok;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = tid >> 4, q = tid & 15;           // row of the block, float4 column of the block
  const int rb = blockIdx.x, cy = blockIdx.y;     // row block (over batch * pixels / 16), 64-channel column block
  const int nrb = pixels / COOP_ROWS;             // row blocks per sample (pixels % 16 == 0: host)
  const int b = rb / nrb, rbs = rb - b * nrb;
  const int64_t row = (int64_t)rb * COOP_ROWS + r;
  const int c = cy * 64 + 4 * q;
  const int cg = channels / groups;               // 64 % cg == 0 or cg % 64 == 0 (host)
  const int gpw = cg >= 64 ? 1 : 64 / cg;         // groups inside this workgroup's 64 channels
  const int wpg = cg >= 64 ? cg / 64 : 1;         // column blocks per group
  const int lg = cg >= 64 ? 0 : (4 * q) / cg;     // this thread's local group
  const int g0 = (cy * 64) / cg;                  // first group of the workgroup
  const int need = nrb * wpg;                     // workgroups per (sample, group)
  // workspace: [0] timeout flag | per (sample, group): arrive, leave | granules [batch * groups][need]
  unsigned* const cnt = ws + 4 + 2 * (b * groups + g0);
  unsigned long long* const gran = reinterpret_cast<unsigned long long*>(ws + 4 + 2 * (gridDim.x / nrb) * groups) +
                                   (int64_t)(b * groups + g0) * need + rbs * wpg + (cg >= 64 ? cy % wpg : 0);

  // ---- reduce: all KS slab loads in flight, then the per-channel inputs of the apply ----
  const float* pp = partial + row * coutp + c;
  float4 v[KS];
#pragma unroll
  for (int z = 0; z < KS; ++z) v[z] = *reinterpret_cast<const float4*>(pp + z * slab_stride);
  float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), rr = bb, sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = bb;
  if (bias) bb = *reinterpret_cast<const float4*>(bias + c);
  const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
  if (scale_shift) {
    sc = *reinterpret_cast<const float4*>(scale_shift + (int64_t)b * ss_ld + c);
    sh = *reinterpret_cast<const float4*>(scale_shift + (int64_t)b * ss_ld + channels + c);
    sc.x += 1.f; sc.y += 1.f; sc.z += 1.f; sc.w += 1.f;
  }
  if (residual) rr = *reinterpret_cast<const float4*>(residual + row * channels + c);
  float4 x = v[0];
#pragma unroll
  for (int z = 1; z < KS; ++z) { x.x += v[z].x; x.y += v[z].y; x.z += v[z].z; x.w += v[z].w; }
  x.x += bb.x; x.y += bb.y; x.z += bb.z; x.w += bb.w;
  float ls = (x.x + x.y) + (x.z + x.w), lq = (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
  // lanes of a wave: (row & 3) << 4 | q: sum over the rows (masks 16, 32) and over the quads of one group (masks < cg / 4)
#pragma unroll
  for (int m = 1; m <= 32; m <<= 1) {
    const bool same_group = m >= 16 || m < (cg >= 64 ? 16 : cg / 4);
    const float os = __shfl_xor(ls, m), oq = __shfl_xor(lq, m);
    if (same_group) { ls += os; lq += oq; }
  }
  if ((lane >> 4) == 0 && (4 * q) % (cg >= 64 ? 64 : cg) == 0) {
    s_ps[wave][lg] = ls;
    s_pq[wave][lg] = lq;
  }
  __syncthreads();
  if (tid < gpw) {       // publish: one granule per (workgroup, group), then count in
    const float ts = (s_ps[0][tid] + s_ps[1][tid]) + (s_ps[2][tid] + s_ps[3][tid]);
    const float tq = (s_pq[0][tid] + s_pq[1][tid]) + (s_pq[2][tid] + s_pq[3][tid]);
    lfdm_agent_store_u64(gran + (int64_t)tid * need, ((unsigned long long)__float_as_uint(tq) << 32) | __float_as_uint(ts));
    LFDM_DRAIN_STORES();
    lfdm_ticket_take(cnt + 2 * tid);
  }
  // ---- wait for the whole (sample, group) cohort: one lane per group polls ----
  if (tid < gpw) {
    bool ok = true;
    unsigned spins = 0;
    while (lfdm_agent_load_u32(cnt + 2 * tid) < (unsigned)need) {
      lfdm_sleep();
      if (++spins > (1u << 20)) { ok = false; break; }
    }
    if (!ok) ws[0] = 1u;
  }
  __syncthreads();
  // ---- statistics: wave w sums the granules of local groups w, w + 4 in a fixed order ----
  for (int g = wave; g < gpw; g += 4) {
    const unsigned long long* src = gran + (int64_t)g * need - (rbs * wpg + (cg >= 64 ? cy % wpg : 0));
    double ds = 0.0, dq = 0.0;
    for (int k = lane; k < need; k += 64) {
      const unsigned long long u = lfdm_agent_load_u64(src + k);
      ds += (double)__uint_as_float((unsigned)u);
      dq += (double)__uint_as_float((unsigned)(u >> 32));
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      ds += __shfl_xor(ds, m);
      dq += __shfl_xor(dq, m);
    }
    if (lane == 0) {
      const double n = (double)pixels * (double)cg;
      const double mean = ds / n;
      double var = dq / n - mean * mean;
      if (var < 0.0) var = 0.0;
      s_mean[g] = (float)mean;
      s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
  if (tid < gpw) {       // count out; the last one to leave re-arms the group's counters for the next launch
    if (lfdm_ticket_take(cnt + 2 * tid + 1) == (unsigned)need - 1) {
      lfdm_ticket_reset(cnt + 2 * tid);
      lfdm_ticket_reset(cnt + 2 * tid + 1);
    }
  }
  // ---- apply: y = silu(x * A[c] + B[c]) (+ residual), the arithmetic of gn_apply_kernel ----
  const float mean = s_mean[lg], rstd = s_rstd[lg];
  const float xs[4] = {x.x, x.y, x.z, x.w}, gs[4] = {ga.x, ga.y, ga.z, ga.w}, bs[4] = {be.x, be.y, be.z, be.w};
  const float scs[4] = {sc.x, sc.y, sc.z, sc.w}, shs[4] = {sh.x, sh.y, sh.z, sh.w}, rs[4] = {rr.x, rr.y, rr.z, rr.w};
  float y[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float a = rstd * gs[e];
    float bbv = bs[e] - mean * a, aa = a;
    if (scale_shift) {
      aa = a * scs[e];
      bbv = bbv * scs[e] + shs[e];
    }
    float t = fmaf(xs[e], aa, bbv);
    if (silu) t = siluf_(t);
    y[e] = t + rs[e];
  }
  *reinterpret_cast<float4*>(out + row * channels + c) = make_float4(y[0], y[1], y[2], y[3]);
}
}  // namespace

extern "C" int lfdm_groupnorm_splitk_ok(int pixels, int channels, int groups) {
  if (pixels <= 0 || channels <= 0 || groups <= 0 || groups > 64 || channels % groups != 0) return 0;
  const int cg = channels / groups;
  if (cg % 4 != 0 || 1024 % (cg / 4) != 0) return 0;
  return (int64_t)pixels * (cg / 4) <= (int64_t)GSK_MAX * 1024 ? 1 : 0;
}

extern "C" int lfdm_groupnorm_splitk_apply_cl_f32(const float* partial, int ksplit, long long slab_stride, int coutp, const float* bias,
                                                  float* out, int batch, int pixels, int channels, int groups, const float* gamma,
                                                  const float* beta, const float* scale_shift, int ss_ld, const float* residual,
                                                  float eps, int apply_silu, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!partial || !out || !gamma || !beta || ksplit < 1 || batch <= 0 || coutp < channels || coutp % 4 != 0 ||
      slab_stride < (long long)batch * pixels * coutp || (scale_shift && ss_ld < 2 * channels) ||
      !lfdm_groupnorm_splitk_ok(pixels, channels, groups) || (((uintptr_t)partial | (uintptr_t)out) & 15) != 0 ||
      (bias && ((uintptr_t)bias & 15) != 0) || (residual && ((uintptr_t)residual & 15) != 0)) {
    lfdm_set_error("groupnorm_splitk_apply: bad arguments (see lfdm_groupnorm_splitk_ok)");
    return LFDM_EINVAL;
  }
  LFDM_LAUNCH(gn_splitk_apply_kernel, dim3((unsigned)groups, (unsigned)batch), dim3(1024), 0, stream, partial, ksplit, (int64_t)slab_stride,
              coutp, bias, out, pixels, channels, groups, gamma, beta, scale_shift, ss_ld, residual, eps, apply_silu);
  return lfdm_check_launch("groupnorm_splitk_apply");
}

extern "C" int lfdm_groupnorm_splitk_coop_ok(int batch, int pixels, int channels, int groups, int ksplit) {
  if (batch <= 0 || pixels <= 0 || channels <= 0 || groups <= 0 || groups > 64 || channels % groups != 0 || ksplit < 2 || ksplit > 8) return 0;
  const int cg = channels / groups;
  if (pixels % COOP_ROWS != 0 || channels % 64 != 0 || cg % 4 != 0 || !((cg <= 64 && 64 % cg == 0 && 64 / cg <= 8) || cg % 64 == 0)) return 0;
  // every workgroup must be resident while its cohort gathers: at most four 256-thread workgroups per CU (MI355X_MICROARCH.md "Residency")
  return (int64_t)batch * (pixels / COOP_ROWS) * (channels / 64) <= 1024 ? 1 : 0;
}

extern "C" size_t lfdm_groupnorm_splitk_coop_ws_bytes(int batch, int pixels, int channels, int groups) {
  if (batch <= 0 || pixels <= 0 || channels <= 0 || groups <= 0 || channels % groups != 0) return 0;
  const int cg = channels / groups;
  const size_t need = (size_t)(pixels / COOP_ROWS) * (cg >= 64 ? cg / 64 : 1);
  return (4 + 2 * (size_t)batch * groups) * sizeof(unsigned) + (size_t)batch * groups * need * sizeof(unsigned long long) + 16;
}

extern "C" int lfdm_groupnorm_splitk_coop_cl_f32(const float* partial, int ksplit, long long slab_stride, int coutp, const float* bias,
                                                 float* out, int batch, int pixels, int channels, int groups, const float* gamma,
                                                 const float* beta, const float* scale_shift, int ss_ld, const float* residual,
                                                 float eps, int apply_silu, void* sync_ws, size_t sync_ws_bytes, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!partial || !out || !gamma || !beta || !sync_ws || coutp < channels || coutp % 4 != 0 ||
      slab_stride < (long long)batch * pixels * coutp || (scale_shift && (ss_ld < 2 * channels || ss_ld % 4 != 0)) ||
      !lfdm_groupnorm_splitk_coop_ok(batch, pixels, channels, groups, ksplit) ||
      (((uintptr_t)partial | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)sync_ws) & 15) != 0 || (slab_stride % 4) != 0 ||
      (bias && ((uintptr_t)bias & 15) != 0) || (residual && ((uintptr_t)residual & 15) != 0) || (scale_shift && ((uintptr_t)scale_shift & 15) != 0)) {
    lfdm_set_error("groupnorm_splitk_coop: bad arguments (see lfdm_groupnorm_splitk_coop_ok; 16-byte aligned pointers)");
    return LFDM_EINVAL;
  }
  if (sync_ws_bytes < lfdm_groupnorm_splitk_coop_ws_bytes(batch, pixels, channels, groups)) {
    lfdm_set_error("groupnorm_splitk_coop: sync workspace too small (lfdm_groupnorm_splitk_coop_ws_bytes; zero it ONCE before the first use)");
    return LFDM_EWORKSPACE;
  }
#if defined(LFDM_EMU_BUILD)
  // the fiber emulator runs the workgroups of a launch one after the other per OS thread: a workgroup that waits for its cohort would wait
  // for ever.  The emulation build serves this entry point with the one-workgroup-per-group kernel (same arithmetic order per element,
  // statistics summed in another order) so that callers can be exercised without a GPU; the cooperative kernel itself is tested on the GPU.
  (void)sync_ws_bytes;
  LFDM_LAUNCH(gn_splitk_apply_kernel, dim3((unsigned)groups, (unsigned)batch), dim3(1024), 0, stream, partial, ksplit, (int64_t)slab_stride,
              coutp, bias, out, pixels, channels, groups, gamma, beta, scale_shift, ss_ld, residual, eps, apply_silu);
  return lfdm_check_launch("groupnorm_splitk_coop");
#else
  const dim3 grid((unsigned)(batch * (pixels / COOP_ROWS)), (unsigned)(channels / 64));
  unsigned* ws = reinterpret_cast<unsigned*>(sync_ws);
#define LFDM_COOP(KS) LFDM_LAUNCH(gn_splitk_coop_kernel<KS>, grid, dim3(256), 0, stream, partial, (int64_t)slab_stride, coutp, bias, out, pixels, channels, \
                                  groups, gamma, beta, scale_shift, ss_ld, residual, eps, apply_silu, ws)
  switch (ksplit) {
    case 2: LFDM_COOP(2); break;
    case 3: LFDM_COOP(3); break;
    case 4: LFDM_COOP(4); break;
    case 5: LFDM_COOP(5); break;
    case 6: LFDM_COOP(6); break;
    case 7: LFDM_COOP(7); break;
    default: LFDM_COOP(8); break;
  }
#undef LFDM_COOP
  return lfdm_check_launch("groupnorm_splitk_coop");
#endif
}

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
