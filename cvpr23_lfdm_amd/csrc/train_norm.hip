// Backward of the normalisation kernels (include/lfdm_hip.h, training section):
//   GroupNorm(8) + (scale+1, shift) + SiLU   (reference Block.forward, video_flow_diffusion.py:200-211)
//   channel LayerNorm                         (reference LayerNorm, :170-179)
//
// GroupNorm.  Forward: n = (x - mean_g) * rstd_g,  z = n * a_c + b_c  with a = gamma*(scale+1),
// b = beta*(scale+1) + shift,  y = silu(z).  With dz = dy * silu'(z) and the per-(sample, channel) sums
//   S1 = sum_pix dz,   S2 = sum_pix dz * n
// everything follows:  dgamma = sum_b (scale+1) S2,  dbeta = sum_b (scale+1) S1,
//   dscale = gamma S2 + beta S1,  dshift = S1,
//   G1 = sum_{c in g} a S1,  G2 = sum_{c in g} a S2,  N = pixels * C/G,
//   dx = rstd * (dz * a - G1/N - n * G2/N).
// Three streaming passes over (x, dy): per-chunk (S1, S2) partials with a fixed summation order
// (no float atomics), a tiny per-sample finalize, and the dx pass.  mean/rstd are re-derived from the
// forward's (sum, sumsq) partials exactly as the forward kernel does (double merge), so forward and
// backward see the same statistics.
#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

constexpr int GNB_MAX_C = 1024;
constexpr int GNB_MAX_CHUNKS = 256;

__host__ __device__ inline int gnb_num_chunks(int pixels) {
  int n = (pixels + 31) / 32;
  if (n < 1) n = 1;
  if (n > GNB_MAX_CHUNKS) n = GNB_MAX_CHUNKS;
  return n;
}

// group statistics from the forward partials -> s_mean / s_rstd (same arithmetic as gn_apply_kernel)
__device__ __forceinline__ void group_stats(const float* partial, int nchunk, int b, int groups, int pixels,
                                            int channels, float eps, float* s_mean, float* s_rstd) {
  const int tid = threadIdx.x, sub = tid & 31;
  for (int g0 = 0; g0 < groups; g0 += 8) {
    const int g = g0 + (tid >> 5);
    double s = 0.0, q = 0.0;
    if (g < groups) {
      for (int k = sub; k < nchunk; k += 32) {
        const float* src = partial + (((int64_t)b * nchunk + k) * groups + g) * 2;
        s += (double)src[0];
        q += (double)src[1];
      }
    }
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
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
}

__device__ __forceinline__ float dact(float z, int silu) {
  if (!silu) return 1.f;
  const float sg = sigmoidf_(z);
  return sg * (1.f + z * (1.f - sg));
}

struct GnArgs {
  const float* x;
  const float* dy;
  int pixels, channels, groups;
  const float* partial;
  int nchunk;
  const float* gamma;
  const float* beta;
  const float* scale_shift;
  int ss_ld;
  float eps;
  int silu;
};

// per-channel forward coefficients in LDS: z = x*A + B, n = x*Rn + Mn
__device__ __forceinline__ void channel_coeffs(const GnArgs& a, int b, const float* s_mean, const float* s_rstd,
                                               float* s_A, float* s_B, float* s_Rn, float* s_Mn) {
  const int cg = a.channels / a.groups;
  for (int c = threadIdx.x; c < a.channels; c += 256) {
    const int g = c / cg;
    float sc = 1.f, sh = 0.f;
    if (a.scale_shift) {
      sc = a.scale_shift[(int64_t)b * a.ss_ld + c] + 1.0f;
      sh = a.scale_shift[(int64_t)b * a.ss_ld + a.channels + c];
    }
    const float ac = a.gamma[c] * sc;
    s_Rn[c] = s_rstd[g];
    s_Mn[c] = -s_mean[g] * s_rstd[g];
    s_A[c] = s_rstd[g] * ac;
    s_B[c] = a.beta[c] * sc + sh - s_mean[g] * s_rstd[g] * ac;
  }
}

// pass 1: grid (nchunk2, B).  part2[b][chunk][c] = (S1, S2) over the chunk's pixels.
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(GnArgs a, float* __restrict__ part2) {
  __shared__ float s_mean[64], s_rstd[64];
  __shared__ __attribute__((aligned(16))) float s_A[GNB_MAX_C], s_B[GNB_MAX_C], s_Rn[GNB_MAX_C], s_Mn[GNB_MAX_C];
  __shared__ __attribute__((aligned(16))) float red1[1024], red2[1024];
  const int b = blockIdx.y, tid = threadIdx.x;
  group_stats(a.partial, a.nchunk, b, a.groups, a.pixels, a.channels, a.eps, s_mean, s_rstd);
  __syncthreads();
  channel_coeffs(a, b, s_mean, s_rstd, s_A, s_B, s_Rn, s_Mn);
  __syncthreads();
  const int c4n = a.channels >> 2;
  const int nchunk2 = gridDim.x, chunk = blockIdx.x;
  const int per = (a.pixels + nchunk2 - 1) / nchunk2;
  const int p0 = chunk * per;
  const int p1 = (p0 + per < a.pixels) ? p0 + per : a.pixels;
  // thread -> (pixel lane, float4 column); columns beyond 256 threads are covered by a loop
  for (int cbase = 0; cbase < c4n; cbase += 256) {
    const int cols = (c4n - cbase) < 256 ? (c4n - cbase) : 256;     // float4 columns in this sweep
    const int rows_per_iter = 256 / cols;
    const int c4 = cbase + tid % cols, prow = tid / cols;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (prow < rows_per_iter) {
      const float4 A = *reinterpret_cast<const float4*>(s_A + 4 * c4), B = *reinterpret_cast<const float4*>(s_B + 4 * c4);
      const float4 R = *reinterpret_cast<const float4*>(s_Rn + 4 * c4), Mn = *reinterpret_cast<const float4*>(s_Mn + 4 * c4);
      for (int p = p0 + prow; p < p1; p += rows_per_iter) {
        const int64_t off = ((int64_t)b * a.pixels + p) * a.channels + 4 * c4;
        const float4 xv = *reinterpret_cast<const float4*>(a.x + off);
        const float4 gv = *reinterpret_cast<const float4*>(a.dy + off);
        float dz;
        dz = gv.x * dact(fmaf(xv.x, A.x, B.x), a.silu); s1.x += dz; s2.x += dz * fmaf(xv.x, R.x, Mn.x);
        dz = gv.y * dact(fmaf(xv.y, A.y, B.y), a.silu); s1.y += dz; s2.y += dz * fmaf(xv.y, R.y, Mn.y);
        dz = gv.z * dact(fmaf(xv.z, A.z, B.z), a.silu); s1.z += dz; s2.z += dz * fmaf(xv.z, R.z, Mn.z);
        dz = gv.w * dact(fmaf(xv.w, A.w, B.w), a.silu); s1.w += dz; s2.w += dz * fmaf(xv.w, R.w, Mn.w);
      }
    }
    __syncthreads();
    *reinterpret_cast<float4*>(red1 + 4 * tid) = s1;
    *reinterpret_cast<float4*>(red2 + 4 * tid) = s2;
    __syncthreads();
    if (tid < cols) {
      float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
      for (int r = 0; r < rows_per_iter; ++r) {
        const float4 u = *reinterpret_cast<const float4*>(red1 + 4 * (r * cols + tid));
        const float4 v = *reinterpret_cast<const float4*>(red2 + 4 * (r * cols + tid));
        t1.x += u.x; t1.y += u.y; t1.z += u.z; t1.w += u.w;
        t2.x += v.x; t2.y += v.y; t2.z += v.z; t2.w += v.w;
      }
      float* dst = part2 + (((int64_t)b * nchunk2 + chunk) * a.channels + 4 * (cbase + tid)) * 2;
      dst[0] = t1.x; dst[1] = t2.x; dst[2] = t1.y; dst[3] = t2.y;
      dst[4] = t1.z; dst[5] = t2.z; dst[6] = t1.w; dst[7] = t2.w;
    }
  }
}

// pass 2: grid (ceil(C / 64), B).  sums[b][c] = (S1, S2);  dss[b] = [dscale | dshift];  dgb_part[b] = [dgamma part | dbeta part].
// 256 threads = 64 channels x 4 interleaved chunk ranges (part q sums chunks q, q+4, ...; four independent 8-byte loads in flight), folded in
// the fixed order 0..3 through LDS.  (Round 4: one workgroup per sample, every thread walking all chunks serially - 8 workgroups and a
// dependent load chain per launch: 39 us on average, 1.6 ms of a B = 8 training step for 40 launches.)
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(GnArgs a, const float* __restrict__ part2, int nchunk2,
                                                              float* __restrict__ sums, float* __restrict__ dss, int dss_ld,
                                                              float* __restrict__ dgb_part) {
  __shared__ float s_p[4][64][2];
  const int b = blockIdx.y;
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s1 = 0.f, s2 = 0.f;
  if (c < a.channels) {
    const float* base = part2 + ((int64_t)b * nchunk2 * a.channels + c) * 2;
    int k = q;
    for (; k + 12 < nchunk2; k += 16) {
      const float2 v0 = *reinterpret_cast<const float2*>(base + (int64_t)k * a.channels * 2);
      const float2 v1 = *reinterpret_cast<const float2*>(base + (int64_t)(k + 4) * a.channels * 2);
      const float2 v2 = *reinterpret_cast<const float2*>(base + (int64_t)(k + 8) * a.channels * 2);
      const float2 v3 = *reinterpret_cast<const float2*>(base + (int64_t)(k + 12) * a.channels * 2);
      s1 += v0.x; s2 += v0.y;
      s1 += v1.x; s2 += v1.y;
      s1 += v2.x; s2 += v2.y;
      s1 += v3.x; s2 += v3.y;
    }
    for (; k < nchunk2; k += 4) {
      const float2 v = *reinterpret_cast<const float2*>(base + (int64_t)k * a.channels * 2);
      s1 += v.x; s2 += v.y;
    }
  }
  s_p[q][cl][0] = s1;
  s_p[q][cl][1] = s2;
  __syncthreads();
  if (q != 0 || c >= a.channels) return;
  s1 = ((s_p[0][cl][0] + s_p[1][cl][0]) + s_p[2][cl][0]) + s_p[3][cl][0];
  s2 = ((s_p[0][cl][1] + s_p[1][cl][1]) + s_p[2][cl][1]) + s_p[3][cl][1];
  sums[((int64_t)b * a.channels + c) * 2 + 0] = s1;
  sums[((int64_t)b * a.channels + c) * 2 + 1] = s2;
  float sc = 1.f;
  if (a.scale_shift) {
    sc = a.scale_shift[(int64_t)b * a.ss_ld + c] + 1.0f;
    dss[(int64_t)b * dss_ld + c] = a.gamma[c] * s2 + a.beta[c] * s1;
    dss[(int64_t)b * dss_ld + a.channels + c] = s1;
  }
  dgb_part[(int64_t)b * 2 * a.channels + c] = sc * s2;
  dgb_part[(int64_t)b * 2 * a.channels + a.channels + c] = sc * s1;
}

// pass 3: grid (blocks, B).  dx = dz*K1[c] - (C0[g] + x*C1[g])
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GnArgs a, const float* __restrict__ sums, float* __restrict__ dx) {
  __shared__ float s_mean[64], s_rstd[64], s_g1[64], s_g2[64];
  __shared__ __attribute__((aligned(16))) float s_A[GNB_MAX_C], s_B[GNB_MAX_C], s_K1[GNB_MAX_C], s_C0[GNB_MAX_C], s_C1[GNB_MAX_C];
  const int b = blockIdx.y, tid = threadIdx.x;
  group_stats(a.partial, a.nchunk, b, a.groups, a.pixels, a.channels, a.eps, s_mean, s_rstd);
  __syncthreads();
  channel_coeffs(a, b, s_mean, s_rstd, s_A, s_B, s_C0, s_C1);       // C0/C1 used as scratch for Rn/Mn here
  __syncthreads();
  const int cg = a.channels / a.groups;
  if (tid < a.groups) {
    float g1 = 0.f, g2 = 0.f;
    for (int c = tid * cg; c < (tid + 1) * cg; ++c) {
      const float ac = s_A[c] / s_rstd[tid];          // a_c = gamma*(scale+1)
      g1 += ac * sums[((int64_t)b * a.channels + c) * 2 + 0];
      g2 += ac * sums[((int64_t)b * a.channels + c) * 2 + 1];
    }
    const float invn = 1.0f / ((float)a.pixels * (float)cg);
    s_g1[tid] = g1 * invn;
    s_g2[tid] = g2 * invn;
  }
  __syncthreads();
  for (int c = tid; c < a.channels; c += 256) {
    const int g = c / cg;
    const float r = s_rstd[g];
    s_K1[c] = s_A[c];                                   // rstd * a_c
    const float c1 = r * r * s_g2[g];
    s_C1[c] = c1;
    s_C0[c] = r * s_g1[g] - s_mean[g] * c1;
  }
  __syncthreads();
  const int c4n = a.channels >> 2;
  const int64_t per_b = (int64_t)a.pixels * c4n;
  const float4* xb = reinterpret_cast<const float4*>(a.x) + (int64_t)b * per_b;
  const float4* gb = reinterpret_cast<const float4*>(a.dy) + (int64_t)b * per_b;
  float4* ob = reinterpret_cast<float4*>(dx) + (int64_t)b * per_b;
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < per_b; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % c4n) * 4;
    const float4 xv = xb[i], gv = gb[i];
    const float4 A = *reinterpret_cast<const float4*>(s_A + c), B = *reinterpret_cast<const float4*>(s_B + c);
    const float4 K1 = *reinterpret_cast<const float4*>(s_K1 + c);
    const float4 C0 = *reinterpret_cast<const float4*>(s_C0 + c), C1 = *reinterpret_cast<const float4*>(s_C1 + c);
    float4 o;
    o.x = gv.x * dact(fmaf(xv.x, A.x, B.x), a.silu) * K1.x - fmaf(xv.x, C1.x, C0.x);
    o.y = gv.y * dact(fmaf(xv.y, A.y, B.y), a.silu) * K1.y - fmaf(xv.y, C1.y, C0.y);
    o.z = gv.z * dact(fmaf(xv.z, A.z, B.z), a.silu) * K1.z - fmaf(xv.z, C1.z, C0.z);
    o.w = gv.w * dact(fmaf(xv.w, A.w, B.w), a.silu) * K1.w - fmaf(xv.w, C1.w, C0.w);
    ob[i] = o;
  }
}

// ---------------- channel LayerNorm backward: one wavefront per row ----------------
// y = (x - mean)/sqrt(var + eps) * gamma ;  dn = dy*gamma ;  dx = (dn - mean(dn) - nh*mean(dn*nh)) / denom
// dgamma partial per workgroup: part[blk][c] = sum over the workgroup's rows of dy*nh
// dx_add (may be null): a second gradient of x (the residual path around the normalised branch) summed into dx here instead of by an ATen add.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dx, int64_t rows, int channels,
                                                            const float* __restrict__ gamma, float eps,
                                                            float* __restrict__ part, const float* __restrict__ dx_add) {
  __shared__ __attribute__((aligned(16))) float red[4][GNB_MAX_C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4n = channels >> 2;
  float4 dg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * channels);
    const float4* gr = reinterpret_cast<const float4*>(dy + row * channels);
    float4 v[4], g[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = lane + 64 * i;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      g[i] = v[i];
      if (f < c4n) {
        v[i] = xr[f];
        g[i] = gr[f];
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
    const float inv = 1.0f / sqrtf(var + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = lane + 64 * i;
      if (f < c4n) {
        const float4 gm = reinterpret_cast<const float4*>(gamma)[f];
        // v <- nh (normalised), g stays dy; dn = dy*gamma
        v[i].x = (v[i].x - mean) * inv; v[i].y = (v[i].y - mean) * inv;
        v[i].z = (v[i].z - mean) * inv; v[i].w = (v[i].w - mean) * inv;
        dg[i].x += g[i].x * v[i].x; dg[i].y += g[i].y * v[i].y; dg[i].z += g[i].z * v[i].z; dg[i].w += g[i].w * v[i].w;
        g[i].x *= gm.x; g[i].y *= gm.y; g[i].z *= gm.z; g[i].w *= gm.w;
        m1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        m2 += (g[i].x * v[i].x + g[i].y * v[i].y) + (g[i].z * v[i].z + g[i].w * v[i].w);
      }
    }
    m1 = wave_sum(m1) / (float)channels;
    m2 = wave_sum(m2) / (float)channels;
    float4* orow = reinterpret_cast<float4*>(dx + row * channels);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = lane + 64 * i;
      if (f < c4n) {
        float4 o;
        o.x = (g[i].x - m1 - v[i].x * m2) * inv;
        o.y = (g[i].y - m1 - v[i].y * m2) * inv;
        o.z = (g[i].z - m1 - v[i].z * m2) * inv;
        o.w = (g[i].w - m1 - v[i].w * m2) * inv;
        if (dx_add) {
          const float4 r = reinterpret_cast<const float4*>(dx_add + row * channels)[f];
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        orow[f] = o;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = lane + 64 * i;
    if (f < c4n) *reinterpret_cast<float4*>(&red[wave][4 * f]) = dg[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < channels; c += 256)
    part[(int64_t)blockIdx.x * channels + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// The same for C = 64 / 128 (the two finest levels: 16 / 32 float4 per row): a wavefront serves 64 / C4N rows at once - lane = (row of the group,
// float4 column), the row reductions are xor-shuffles inside C4N adjacent lanes - and two row groups are in flight per trip.  One row per
// wavefront left 48 of 64 lanes idle at C = 64 and walked 327 680 rows of 256 bytes behind four full-wave reductions each: 343 us for 252 MB
// (0.73 TB/s) in the B = 8 training step (profiles/r05_m_train_kernel_stats.txt).
template <int C4N>
__global__ __launch_bounds__(256) void layernorm_bwd_small_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  float* __restrict__ dx, int64_t rows, const float* __restrict__ gamma, float eps,
                                                                  float* __restrict__ part, const float* __restrict__ dx_add) {
  constexpr int RPW = 64 / C4N, C = 4 * C4N;            // rows per wavefront and trip, channels
  __shared__ __attribute__((aligned(16))) float red[4 * RPW][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / C4N, f = lane % C4N;
  const float4 gm = reinterpret_cast<const float4*>(gamma)[f];
  float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
  auto group_sum = [&](float v) {
#pragma unroll
    for (int m = 1; m < C4N; m <<= 1) v += __shfl_xor(v, m);
    return v;
  };
  const int64_t stride = (int64_t)gridDim.x * 4 * RPW * 2;
  for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * RPW * 2; row0 < rows; row0 += stride) {
    float4 v[2], g[2];
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t row = row0 + u * RPW + sub;
      live[u] = row < rows;
      v[u] = g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live[u]) {
        v[u] = reinterpret_cast<const float4*>(x + row * C)[f];
        g[u] = reinterpret_cast<const float4*>(dy + row * C)[f];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float mean = group_sum((v[u].x + v[u].y) + (v[u].z + v[u].w)) / (float)C;
      const float a = v[u].x - mean, b = v[u].y - mean, c = v[u].z - mean, d = v[u].w - mean;
      const float var = group_sum((a * a + b * b) + (c * c + d * d)) / (float)C;
      const float inv = 1.0f / sqrtf(var + eps);
      float4 nh = make_float4(a * inv, b * inv, c * inv, d * inv);
      float4 dn = g[u];
      dg.x += dn.x * nh.x; dg.y += dn.y * nh.y; dg.z += dn.z * nh.z; dg.w += dn.w * nh.w;      // (rows past the end: dy = 0)
      dn.x *= gm.x; dn.y *= gm.y; dn.z *= gm.z; dn.w *= gm.w;
      const float m1 = group_sum((dn.x + dn.y) + (dn.z + dn.w)) / (float)C;
      const float m2 = group_sum((dn.x * nh.x + dn.y * nh.y) + (dn.z * nh.z + dn.w * nh.w)) / (float)C;
      if (live[u]) {
        float4 o;
        o.x = (dn.x - m1 - nh.x * m2) * inv;
        o.y = (dn.y - m1 - nh.y * m2) * inv;
        o.z = (dn.z - m1 - nh.z * m2) * inv;
        o.w = (dn.w - m1 - nh.w * m2) * inv;
        if (dx_add) {
          const float4 r = reinterpret_cast<const float4*>(dx_add + (row0 + u * RPW + sub) * C)[f];
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        reinterpret_cast<float4*>(dx + (row0 + u * RPW + sub) * C)[f] = o;
      }
    }
  }
  *reinterpret_cast<float4*>(&red[wave * RPW + sub][4 * f]) = dg;
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 4 * RPW; ++k) sum += red[k][c];        // fixed order
    part[(int64_t)blockIdx.x * C + c] = sum;
  }
}

int ln_bwd_blocks(int64_t rows) {
  int64_t nb = (rows + 3) / 4;
  if (nb > 512) nb = 512;
  return (int)nb;
}

}  // namespace

extern "C" size_t lfdm_groupnorm_bwd_ws_bytes(int batch, int pixels, int channels) {
  const size_t part2 = (size_t)batch * gnb_num_chunks(pixels) * channels * 2;
  const size_t sums = (size_t)batch * channels * 2;
  const size_t dgb = (size_t)batch * channels * 2;
  return (part2 + sums + dgb) * sizeof(float);
}

extern "C" int lfdm_groupnorm_silu_bwd_cl_f32(const float* x, const float* dy, float* dx, int batch, int pixels,
                                              int channels, int groups, const float* gamma, const float* beta,
                                              const float* scale_shift, int ss_ld, float eps, int apply_silu,
                                              const float* partial, int nchunk, float* dgamma_dbeta,
                                              float* dscale_shift, int dss_ld, void* ws, size_t ws_bytes,
                                              lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !dy || !dx || !gamma || !beta || !partial || !dgamma_dbeta || nchunk <= 0 || batch <= 0 || pixels <= 0 ||
      channels <= 0 || channels > GNB_MAX_C || groups <= 0 || groups > 64 || channels % groups != 0 || channels % 4 != 0 ||
      (scale_shift && (ss_ld < 2 * channels || !dscale_shift || dss_ld < 2 * channels)) ||
      !((channels / 4) >= 256 ? (channels / 4) % 256 == 0 : 256 % (channels / 4) == 0)) {
    lfdm_set_error("groupnorm_bwd: bad arguments (C%4==0, C<=1024, C/4 divides 256)");
    return LFDM_EINVAL;
  }
  if (!ws || ws_bytes < lfdm_groupnorm_bwd_ws_bytes(batch, pixels, channels)) {
    lfdm_set_error("groupnorm_bwd: workspace too small");
    return LFDM_EWORKSPACE;
  }
  const int nchunk2 = gnb_num_chunks(pixels);
  float* part2 = (float*)ws;
  float* sums = part2 + (size_t)batch * nchunk2 * channels * 2;
  float* dgb = sums + (size_t)batch * channels * 2;
  GnArgs a;
  a.x = x; a.dy = dy; a.pixels = pixels; a.channels = channels; a.groups = groups; a.partial = partial;
  a.nchunk = nchunk; a.gamma = gamma; a.beta = beta; a.scale_shift = scale_shift; a.ss_ld = ss_ld; a.eps = eps;
  a.silu = apply_silu;
  LFDM_LAUNCH(gn_bwd_reduce_kernel, dim3(nchunk2, batch), dim3(256), 0, stream, a, part2);
  LFDM_LAUNCH(gn_bwd_finalize_kernel, dim3((unsigned)((channels + 63) / 64), (unsigned)batch), dim3(256), 0, stream, a, (const float*)part2,
              nchunk2, sums, dscale_shift, dss_ld, dgb);
  int rc = lfdm_sum_leading_f32(dgb, dgamma_dbeta, 2 * (int64_t)channels, batch, stream_);
  if (rc) return rc;
  const int64_t per_b = (int64_t)pixels * (channels / 4);
  int64_t nb = (per_b + 255) / 256;
  const int64_t cap = batch >= 8 ? 128 : 1024 / batch;
  if (nb > cap) nb = cap;
  LFDM_LAUNCH(gn_bwd_apply_kernel, dim3((unsigned)nb, batch), dim3(256), 0, stream, a, (const float*)sums, dx);
  return lfdm_check_launch("groupnorm_bwd");
}

extern "C" size_t lfdm_layernorm_bwd_ws_bytes(int64_t rows, int channels) {
  return (size_t)ln_bwd_blocks(rows) * channels * sizeof(float);
}

extern "C" int lfdm_layernorm_bwd_cl_f32(const float* x, const float* dy, float* dx, int64_t rows, int channels,
                                         const float* gamma, float eps, float* dgamma, void* ws, size_t ws_bytes,
                                         lfdm_stream_t stream_) {
  return lfdm_layernorm_bwd_add_cl_f32(x, dy, nullptr, dx, rows, channels, gamma, eps, dgamma, ws, ws_bytes, stream_);
}

extern "C" int lfdm_layernorm_bwd_add_cl_f32(const float* x, const float* dy, const float* dx_add, float* dx, int64_t rows, int channels,
                                             const float* gamma, float eps, float* dgamma, void* ws, size_t ws_bytes,
                                             lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !dy || !dx || !gamma || !dgamma || rows <= 0 || channels <= 0 || channels % 4 != 0 || channels > GNB_MAX_C) {
    lfdm_set_error("layernorm_bwd: unsupported shape (C%4==0, C<=1024)");
    return LFDM_EINVAL;
  }
  if (!ws || ws_bytes < lfdm_layernorm_bwd_ws_bytes(rows, channels)) {
    lfdm_set_error("layernorm_bwd: workspace too small");
    return LFDM_EWORKSPACE;
  }
  const int nb = ln_bwd_blocks(rows);
  const bool al16 = ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)gamma) | ((uintptr_t)dx_add)) & 15) == 0;
  if (channels == 64 && al16) LFDM_LAUNCH((layernorm_bwd_small_kernel<16>), dim3(nb), dim3(256), 0, stream, x, dy, dx, rows, gamma, eps, (float*)ws, dx_add);
  else if (channels == 128 && al16) LFDM_LAUNCH((layernorm_bwd_small_kernel<32>), dim3(nb), dim3(256), 0, stream, x, dy, dx, rows, gamma, eps, (float*)ws, dx_add);
  else LFDM_LAUNCH(layernorm_bwd_kernel, dim3(nb), dim3(256), 0, stream, x, dy, dx, rows, channels, gamma, eps, (float*)ws, dx_add);
  int rc = lfdm_check_launch("layernorm_bwd");
  if (rc) return rc;
  return lfdm_sum_leading_f32((const float*)ws, dgamma, channels, nb, stream_);
}
