// Small / boundary kernels of the UNet (include/lfdm_hip.h): conditioning MLPs, sinusoidal
// timestep embedding (timestep read from device memory for graph replay), the planar-input
// small-C_in direct convolution (UNet init conv's step-dependent 3 channels, LFAE first conv) and
// the two 1x1x1 output heads writing the planar (B,3,T,H,W) prediction.
#include <stdlib.h>
#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

__device__ __forceinline__ float geluf_(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

__device__ __forceinline__ float small_act(float v, int act) {
  if (act == LFDM_ACT_GELU) return geluf_(v);
  return apply_act(v, act);
}

// grid (ceil(N/4), B): one wavefront per output feature
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           float* __restrict__ y, int k, int n,
                                                           int ldx, int ldw, int ldy, int act_in,
                                                           int act_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 4 + wave;
  const int b = blockIdx.y;
  if (col >= n) return;
  const float* xr = x + (int64_t)b * ldx;
  const float* wr = w + (int64_t)col * ldw;
  float s = 0.f;
  for (int i = lane; i < k; i += 64) s = fmaf(small_act(xr[i], act_in), wr[i], s);
  s = wave_sum(s);
  if (lane == 0) {
    if (bias) s += bias[col];
    y[(int64_t)b * ldy + col] = small_act(s, act_out);
  }
}

__global__ __launch_bounds__(64) void sinusoidal_kernel(const int32_t* __restrict__ t_dev,
                                                        int t_stride, const float* __restrict__ freqs,
                                                        float* __restrict__ out, int dim, int ldo) {
  const int b = blockIdx.x;
  const int half = dim / 2;
  const float tt = (float)t_dev[(int64_t)b * t_stride];
  for (int i = threadIdx.x; i < half; i += 64) {
    const float a = tt * freqs[i];
    out[(int64_t)b * ldo + i] = sinf(a);
    out[(int64_t)b * ldo + half + i] = cosf(a);
  }
}

// Direct conv, planar input with few channels -> CL rows.  grid (ceil(B*T*H*W/128), cout/64);
// thread = (pixel = tid&127, 32 output channels = tid>>7).  Weights [K][64-slice] sit in LDS (staged once per 128 pixels with
// 16-byte loads) and are read wave-uniformly (broadcast ds_read_b128); the planar input is read coalesced along W.
// (Round 1: 64 pixels x 16 channels per thread - the 37 KB weight staging per 64 pixels dominated: 46 us for the sampler's
// 3-channel 7x7 stem at 40 x 32 x 32 pixels.)
constexpr int CPI_MAX_K = 7 * 7 * 4;
constexpr int CPI_PIX = 128;
// (A variant that loads the 21 input values of a filter row back to back before their FMAs was measured: 50 vs 42 us - slower;
// what bounds this kernel is not the load latency.)
__global__ __launch_bounds__(256) void conv_planar_in_kernel(
    const float* __restrict__ x, int batch, int cin, int cin_total, int frames, int h, int w,
    const float* __restrict__ wgt, int kh, int kw, int cout, const float* __restrict__ bias,
    const float* __restrict__ add_term, float* __restrict__ out, int ldo, int act) {
  __shared__ __attribute__((aligned(16))) float ws[CPI_MAX_K * 64];
  const int tid = threadIdx.x;
  const int K = kh * kw * cin;
  const int co0 = blockIdx.y * 64;
  if ((cout & 3) == 0 && ((((uintptr_t)wgt) & 15) == 0)) {
    for (int i = tid; i < K * 16; i += 256) {                  // float4 item i: row kk = i / 16, columns 4*(i % 16)
      const int kk = i >> 4, j4 = i & 15;
      *reinterpret_cast<float4*>(ws + kk * 64 + 4 * j4) = *reinterpret_cast<const float4*>(wgt + (int64_t)kk * cout + co0 + 4 * j4);
    }
  } else {
    for (int i = tid; i < K * 64; i += 256) {
      const int kk = i >> 6, j = i & 63;
      ws[i] = wgt[(int64_t)kk * cout + co0 + j];
    }
  }
  __syncthreads();
  const int hw = h * w;
  const int64_t total = (int64_t)batch * frames * hw;
  const int64_t gp = (int64_t)blockIdx.x * CPI_PIX + (tid & (CPI_PIX - 1));
  if (gp >= total) return;
  const int cg = tid >> 7;                                     // which 32 of the 64 output channels
  const int64_t bt = gp / hw;
  const int pix = (int)(gp - bt * hw);
  const int b = (int)(bt / frames), t = (int)(bt - (int64_t)b * frames);
  const int oy = pix / w, ox = pix - oy * w;
  const int py = kh / 2, px = kw / 2;
  float acc[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) acc[j] = 0.f;
  for (int ky = 0; ky < kh; ++ky) {
    const int iy = oy + ky - py;
    if (iy < 0 || iy >= h) continue;
    for (int kx = 0; kx < kw; ++kx) {
      const int ix = ox + kx - px;
      if (ix < 0 || ix >= w) continue;
      for (int c = 0; c < cin; ++c) {
        const float v = x[(((int64_t)b * cin_total + c) * frames + t) * hw + iy * w + ix];
        const float4* wp = reinterpret_cast<const float4*>(ws + ((ky * kw + kx) * cin + c) * 64 + cg * 32);
#pragma unroll
        for (int j4 = 0; j4 < 8; ++j4) {
          const float4 ww = wp[j4];
          acc[4 * j4 + 0] = fmaf(v, ww.x, acc[4 * j4 + 0]);
          acc[4 * j4 + 1] = fmaf(v, ww.y, acc[4 * j4 + 1]);
          acc[4 * j4 + 2] = fmaf(v, ww.z, acc[4 * j4 + 2]);
          acc[4 * j4 + 3] = fmaf(v, ww.w, acc[4 * j4 + 3]);
        }
      }
    }
  }
  const int cbase = co0 + cg * 32;
  float* orow = out + gp * ldo + cbase;
  const float* arow = add_term ? add_term + ((int64_t)b * hw + pix) * cout + cbase : nullptr;
  const bool vec = (ldo & 3) == 0 && (cout & 3) == 0 && ((((uintptr_t)out) & 15) == 0) && (!arow || (((uintptr_t)add_term) & 15) == 0) &&
                   (!bias || (((uintptr_t)bias) & 15) == 0);
  if (vec) {
#pragma unroll
    for (int j4 = 0; j4 < 8; ++j4) {
      float4 v = make_float4(acc[4 * j4], acc[4 * j4 + 1], acc[4 * j4 + 2], acc[4 * j4 + 3]);
      if (bias) {
        const float4 bb = *reinterpret_cast<const float4*>(bias + cbase + 4 * j4);
        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
      }
      if (arow) {
        const float4 aa = *reinterpret_cast<const float4*>(arow + 4 * j4);
        v.x += aa.x; v.y += aa.y; v.z += aa.z; v.w += aa.w;
      }
      v.x = apply_act(v.x, act); v.y = apply_act(v.y, act); v.z = apply_act(v.z, act); v.w = apply_act(v.w, act);
      *reinterpret_cast<float4*>(orow + 4 * j4) = v;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      float v = acc[j];
      if (bias) v += bias[cbase + j];
      if (arow) v += arow[j];
      orow[j] = apply_act(v, act);
    }
  }
}

// one thread per pixel row; weights (3 x C) in LDS.
// EXTRA (round 4, lfdm_heads_res_cl_to_planar_f32): the heads' ResnetBlocks end in h + res_conv(cat(x, r)) and the heads are linear, so
// W1 (h + Wres [x|r] + bres) + b1 = W1 h + (W1 Wres) [x|r] + (W1 bres + b1): the 1x1 res_conv launch (27 us at 32x32: a 40 960 x 128 x 128
// GEMM) becomes three more dot products per pixel against the composed 3 x (c0 + c1) matrix `w_extra` (rows: flow x, flow y, occlusion).
template <bool EXTRA>
__global__ __launch_bounds__(256) void heads_kernel(const float* __restrict__ y_flow,
                                                    const float* __restrict__ y_occ, int channels, int ld,
                                                    const float* __restrict__ w_flow,
                                                    const float* __restrict__ b_flow,
                                                    const float* __restrict__ w_occ,
                                                    const float* __restrict__ b_occ,
                                                    float* __restrict__ out, int batch, int frames,
                                                    int hw, const float* __restrict__ x0, int ld0, int c0,
                                                    const float* __restrict__ x1, int ld1, int c1,
                                                    const float* __restrict__ w_extra) {
  __shared__ __attribute__((aligned(16))) float wl[3 * 256];
  __shared__ __attribute__((aligned(16))) float we[EXTRA ? 3 * 512 : 4];
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * channels; i += 256) wl[i] = w_flow[i];
  for (int i = tid; i < channels; i += 256) wl[2 * channels + i] = w_occ[i];
  if (EXTRA)
    for (int i = tid; i < 3 * (c0 + c1); i += 256) we[i] = w_extra[i];
  __syncthreads();
  const int64_t total = (int64_t)batch * frames * hw;
  const int64_t row = (int64_t)blockIdx.x * 256 + tid;
  if (row >= total) return;
  const float4* fr = reinterpret_cast<const float4*>(y_flow + row * ld);
  const float4* orr = reinterpret_cast<const float4*>(y_occ + row * ld);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int i = 0; i < channels / 4; ++i) {
    const float4 a = fr[i];
    const float4 o = orr[i];
    const float4 w0 = reinterpret_cast<const float4*>(wl)[i];
    const float4 w1 = reinterpret_cast<const float4*>(wl + channels)[i];
    const float4 w2 = reinterpret_cast<const float4*>(wl + 2 * channels)[i];
    s0 += (a.x * w0.x + a.y * w0.y) + (a.z * w0.z + a.w * w0.w);
    s1 += (a.x * w1.x + a.y * w1.y) + (a.z * w1.z + a.w * w1.w);
    s2 += (o.x * w2.x + o.y * w2.y) + (o.z * w2.z + o.w * w2.w);
  }
  if (EXTRA) {
    const int ce = c0 + c1;
    for (int src = 0; src < 2; ++src) {
      const int cs = src == 0 ? c0 : c1, base = src == 0 ? 0 : c0;
      const float4* xr = reinterpret_cast<const float4*>(src == 0 ? x0 + row * ld0 : x1 + row * ld1);
      for (int i = 0; i < cs / 4; ++i) {
        const float4 a = xr[i];
        const float4 w0 = *reinterpret_cast<const float4*>(we + base + 4 * i);
        const float4 w1 = *reinterpret_cast<const float4*>(we + ce + base + 4 * i);
        const float4 w2 = *reinterpret_cast<const float4*>(we + 2 * ce + base + 4 * i);
        s0 += (a.x * w0.x + a.y * w0.y) + (a.z * w0.z + a.w * w0.w);
        s1 += (a.x * w1.x + a.y * w1.y) + (a.z * w1.z + a.w * w1.w);
        s2 += (a.x * w2.x + a.y * w2.y) + (a.z * w2.z + a.w * w2.w);
      }
    }
  }
  const int64_t bt = row / hw;
  const int pix = (int)(row - bt * hw);
  const int b = (int)(bt / frames), t = (int)(bt - (int64_t)b * frames);
  const int64_t fhw = (int64_t)frames * hw;
  float* ob = out + (int64_t)b * 3 * fhw + (int64_t)t * hw + pix;
  ob[0] = s0 + b_flow[0];
  ob[fhw] = s1 + b_flow[1];
  ob[2 * fhw] = s2 + b_occ[0];
}

// The output heads with the LAST GroupNorm folded in (round 6; lfdm_heads_gn_res_cl_to_planar_f32).  The merged heads block ends in
// conv -> GroupNorm(16 groups over 2C channels) + SiLU -> [1x1 heads + folded res_conv] (unet.py run_trunk): the GroupNorm apply was a launch of
// its own that wrote 21 MB for the heads kernel to read back.  Here a workgroup merges the statistics partials of its sample exactly as
// gn_apply_kernel does (double, fixed order), keeps A[c], B[c] in LDS and normalises + activates each value on its way into the three dot products.
// Eight lanes share a pixel row: lane (row r8 = lane >> 3, segment = lane & 7) walks the float4 columns segment, segment + 8, ... of the row
// (one 128-byte line per eight lanes and load instruction - the one-thread-per-row kernel above touches 64 different lines per instruction), the three
// partial sums meet through three xor-shuffles.
__global__ __launch_bounds__(256) void heads_gn_kernel(const float* __restrict__ y, int ld, int channels,
                                                       const float* __restrict__ partial, int nchunk, int groups,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                       const float* __restrict__ w_flow, const float* __restrict__ b_flow,
                                                       const float* __restrict__ w_occ, const float* __restrict__ b_occ,
                                                       const float* __restrict__ x0, int ld0, int c0, const float* __restrict__ x1, int ld1, int c1,
                                                       const float* __restrict__ w_extra, float* __restrict__ out, int frames, int hw, int rows_per_wg) {
  __shared__ float s_mean[64], s_rstd[64];
  __shared__ __attribute__((aligned(16))) float s_a[512], s_b[512];
  __shared__ __attribute__((aligned(16))) float wl[3 * 512];                 // per channel of y: the two flow weights | the occlusion weight (0 where not used)
  __shared__ __attribute__((aligned(16))) float we[3 * 512];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int c2 = 2 * channels;                                                // channels of y: [flow C | occlusion C]
  const int pixels = frames * hw;
  // (round 6) everything the prologue reads that does NOT depend on the statistics - the thread's head weights, extra-term weights, gamma / beta - is
  // requested HERE, in front of the merge, instead of in loops behind it (each of those was a dependent round trip of its own)
  const int ce = c0 + c1;
  const bool pi_ok = tid < c2;
  const float pw_f0 = (pi_ok && tid < channels) ? w_flow[tid] : 0.f, pw_f1 = (pi_ok && tid < channels) ? w_flow[channels + tid] : 0.f;
  const float pw_oc = (pi_ok && tid >= channels) ? w_occ[tid - channels] : 0.f;
  const float pe0 = tid < 3 * ce ? w_extra[tid] : 0.f, pe1 = tid + 256 < 3 * ce ? w_extra[tid + 256] : 0.f;
  const float pre_gamma = gamma[pi_ok ? tid : 0], pre_beta = beta[pi_ok ? tid : 0];
  // ---- statistics of this sample: gn_apply_kernel's merge (32 lanes walk one group's chunks, four loads in flight, k ascending) ----
  {
    const int lpg = 32, gpp = 256 / lpg, sub = tid & (lpg - 1);
    for (int g0 = 0; g0 < groups; g0 += gpp) {
      const int g = g0 + tid / lpg;
      double sm = 0.0, sq = 0.0;
      if (g < groups) {
        const float2* src = reinterpret_cast<const float2*>(partial) + ((int64_t)b * nchunk) * groups + g;
        int k = sub;
        for (; k + 3 * lpg < nchunk; k += 4 * lpg) {
          const float2 v0 = src[(int64_t)k * groups], v1 = src[(int64_t)(k + lpg) * groups];
          const float2 v2 = src[(int64_t)(k + 2 * lpg) * groups], v3 = src[(int64_t)(k + 3 * lpg) * groups];
          sm += (double)v0.x; sq += (double)v0.y;
          sm += (double)v1.x; sq += (double)v1.y;
          sm += (double)v2.x; sq += (double)v2.y;
          sm += (double)v3.x; sq += (double)v3.y;
        }
        for (; k < nchunk; k += lpg) {
          const float2 v = src[(int64_t)k * groups];
          sm += (double)v.x;
          sq += (double)v.y;
        }
      }
      for (int m = lpg >> 1; m >= 1; m >>= 1) {
        sm += __shfl_xor(sm, m);
        sq += __shfl_xor(sq, m);
      }
      if (g < groups && sub == 0) {
        const double n = (double)pixels * (double)(c2 / groups);
        const double mean = sm / n;
        double var = sq / n - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[g] = (float)mean;
        s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
      }
    }
  }
  for (int i = tid; i < c2; i += 256) {                                       // head weights per y channel
    const bool pre = i == tid;
    wl[i] = pre ? pw_f0 : (i < channels ? w_flow[i] : 0.f);
    wl[512 + i] = pre ? pw_f1 : (i < channels ? w_flow[channels + i] : 0.f);
    wl[1024 + i] = pre ? pw_oc : (i < channels ? 0.f : w_occ[i - channels]);
  }
  for (int i = tid; i < 3 * ce; i += 256) we[(i / ce) * 512 + (i % ce)] = i == tid ? pe0 : (i == tid + 256 ? pe1 : w_extra[i]);
  __syncthreads();
  const int cg = c2 / groups;
  for (int c = tid; c < c2; c += 256) {
    const int g = c / cg;
    const float a = s_rstd[g] * (c == tid ? pre_gamma : gamma[c]);
    s_a[c] = a;
    s_b[c] = (c == tid ? pre_beta : beta[c]) - s_mean[g] * a;
  }
  __syncthreads();
  const int r8 = tid >> 3, seg = tid & 7;
  const float bf0 = b_flow[0], bf1 = b_flow[1], bo = b_occ[0];
  const int64_t fhw = (int64_t)frames * hw;
  auto finish = [&](int pix, float s0, float s1, float s2) {
#pragma unroll
    for (int m = 1; m <= 4; m <<= 1) {
      s0 += __shfl_xor(s0, m);
      s1 += __shfl_xor(s1, m);
      s2 += __shfl_xor(s2, m);
    }
    if (seg == 0) {
      const int t = pix / hw, px = pix - t * hw;
      float* ob = out + (int64_t)b * 3 * fhw + (int64_t)t * hw + px;
      ob[0] = s0 + bf0;
      ob[fhw] = s1 + bf1;
      ob[2 * fhw] = s2 + bo;
    }
  };
  if (c2 == 128 && c0 == 64 && c1 == 64) {
    // The UNet's own shape (dim 64): a lane's float4 columns are the same for every row it visits - its A / B / head-weight fragments live in
    // registers (32 float4) instead of five LDS reads per loaded float4 (the first version: 25 us for 42 MB - LDS-read bound)
    float4 ra[4], rb[4], w0[4], w1[4], w2[4], e0[4], e1[4], e2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = seg + 8 * j;
      ra[j] = *reinterpret_cast<const float4*>(s_a + 4 * k);
      rb[j] = *reinterpret_cast<const float4*>(s_b + 4 * k);
      w0[j] = *reinterpret_cast<const float4*>(wl + 4 * k);
      w1[j] = *reinterpret_cast<const float4*>(wl + 512 + 4 * k);
      w2[j] = *reinterpret_cast<const float4*>(wl + 1024 + 4 * k);
      // extra term: j = 0, 1 -> x0 columns seg, seg + 8; j = 2, 3 -> x1 columns seg, seg + 8 (offset c0 in w_extra)
      const int ke = (j < 2 ? 0 : 64) + 4 * (seg + 8 * (j & 1));
      e0[j] = *reinterpret_cast<const float4*>(we + ke);
      e1[j] = *reinterpret_cast<const float4*>(we + 512 + ke);
      e2[j] = *reinterpret_cast<const float4*>(we + 1024 + ke);
    }
    for (int p0 = 0; p0 < rows_per_wg; p0 += 32) {
      const int pix = blockIdx.x * rows_per_wg + p0 + r8;
      if (pix >= pixels) break;
      const int64_t row = (int64_t)b * pixels + pix;
      const float4* yr = reinterpret_cast<const float4*>(y + row * ld);
      const float4* x0r = reinterpret_cast<const float4*>(x0 + row * ld0);
      const float4* x1r = reinterpret_cast<const float4*>(x1 + row * ld1);
      float4 v[4], xv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = yr[seg + 8 * j];
      xv[0] = x0r[seg]; xv[1] = x0r[seg + 8]; xv[2] = x1r[seg]; xv[3] = x1r[seg + 8];
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t0 = siluf_(fmaf(v[j].x, ra[j].x, rb[j].x)), t1 = siluf_(fmaf(v[j].y, ra[j].y, rb[j].y));
        const float t2 = siluf_(fmaf(v[j].z, ra[j].z, rb[j].z)), t3 = siluf_(fmaf(v[j].w, ra[j].w, rb[j].w));
        s0 += (t0 * w0[j].x + t1 * w0[j].y) + (t2 * w0[j].z + t3 * w0[j].w);
        s1 += (t0 * w1[j].x + t1 * w1[j].y) + (t2 * w1[j].z + t3 * w1[j].w);
        s2 += (t0 * w2[j].x + t1 * w2[j].y) + (t2 * w2[j].z + t3 * w2[j].w);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s0 += (xv[j].x * e0[j].x + xv[j].y * e0[j].y) + (xv[j].z * e0[j].z + xv[j].w * e0[j].w);
        s1 += (xv[j].x * e1[j].x + xv[j].y * e1[j].y) + (xv[j].z * e1[j].z + xv[j].w * e1[j].w);
        s2 += (xv[j].x * e2[j].x + xv[j].y * e2[j].y) + (xv[j].z * e2[j].z + xv[j].w * e2[j].w);
      }
      finish(pix, s0, s1, s2);
    }
    return;
  }
  for (int p0 = 0; p0 < rows_per_wg; p0 += 32) {
    const int pix = blockIdx.x * rows_per_wg + p0 + r8;                       // pixel of the sample (frame-major)
    if (pix >= pixels) break;                                                 // (whole 8-lane groups leave together)
    const int64_t row = (int64_t)b * pixels + pix;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    const float4* yr = reinterpret_cast<const float4*>(y + row * ld);
    for (int k = seg; k < c2 / 4; k += 8) {
      const float4 v = yr[k];
      const float4 a = *reinterpret_cast<const float4*>(s_a + 4 * k), d = *reinterpret_cast<const float4*>(s_b + 4 * k);
      const float4 w0 = *reinterpret_cast<const float4*>(wl + 4 * k), w1 = *reinterpret_cast<const float4*>(wl + 512 + 4 * k);
      const float4 w2 = *reinterpret_cast<const float4*>(wl + 1024 + 4 * k);
      const float t0 = siluf_(fmaf(v.x, a.x, d.x)), t1 = siluf_(fmaf(v.y, a.y, d.y)), t2 = siluf_(fmaf(v.z, a.z, d.z)), t3 = siluf_(fmaf(v.w, a.w, d.w));
      s0 += (t0 * w0.x + t1 * w0.y) + (t2 * w0.z + t3 * w0.w);
      s1 += (t0 * w1.x + t1 * w1.y) + (t2 * w1.z + t3 * w1.w);
      s2 += (t0 * w2.x + t1 * w2.y) + (t2 * w2.z + t3 * w2.w);
    }
#pragma unroll
    for (int src = 0; src < 2; ++src) {
      const int cs = src == 0 ? c0 : c1, base = src == 0 ? 0 : c0;
      if (cs == 0) continue;
      const float4* xr = reinterpret_cast<const float4*>(src == 0 ? x0 + row * ld0 : x1 + row * ld1);
      for (int k = seg; k < cs / 4; k += 8) {
        const float4 v = xr[k];
        const float4 w0 = *reinterpret_cast<const float4*>(we + base + 4 * k), w1 = *reinterpret_cast<const float4*>(we + 512 + base + 4 * k);
        const float4 w2 = *reinterpret_cast<const float4*>(we + 1024 + base + 4 * k);
        s0 += (v.x * w0.x + v.y * w0.y) + (v.z * w0.z + v.w * w0.w);
        s1 += (v.x * w1.x + v.y * w1.y) + (v.z * w1.z + v.w * w1.w);
        s2 += (v.x * w2.x + v.y * w2.y) + (v.z * w2.z + v.w * w2.w);
      }
    }
    finish(pix, s0, s1, s2);
  }
}

// out[b][i] = step_table[*step][i] + batch_base[b][i]
__global__ __launch_bounds__(256) void step_cond_kernel(const float* __restrict__ step_table,
                                                        const float* __restrict__ batch_base,
                                                        const int32_t* __restrict__ step_dev,
                                                        float* __restrict__ out, int batch, int n) {
  const float* row = step_table + (int64_t)(*step_dev) * n;
  const int64_t total = (int64_t)batch * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % n);
    out[i] = row[c] + batch_base[i];
  }
}

// Direct-form filter pack (lfdm_pack_conv_weight_f32): one thread per element of out[(parity)][chunk][n][kk]
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(const float* __restrict__ w, int n_o, int n_i, int taps,
                                                               int64_t stride_o, int64_t stride_i, int mode, int K, int N,
                                                               int np, float* __restrict__ out) {
  const int64_t per = (int64_t)((K + 31) / 32) * np * 32;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= per * (mode == 2 ? 4 : 1)) return;
  const int q = (int)(idx / per);                      // parity (mode 2)
  const int64_t r = idx - q * per;
  const int kk = (int)(r & 31);
  const int n = (int)((r >> 5) % np);
  const int k = (int)((r >> 5) / np) * 32 + kk;
  float v = 0.f;
  if (k < K && n < N) {
    if (mode == 0) {
      const int tap = k / n_i, i = k - tap * n_i;
      v = w[n * stride_o + i * stride_i + tap];
    } else if (mode == 1) {
      const int tap = k / n_o, o = k - tap * n_o;
      v = w[o * stride_o + n * stride_i + (taps - 1 - tap)];
    } else {
      const int t = k / n_o, ci = k - t * n_o;
      const int ky = (3 - (q >> 1)) - 2 * (t >> 1), kx = (3 - (q & 1)) - 2 * (t & 1);
      v = w[ci * stride_o + n * stride_i + ky * 4 + kx];
    }
  }
  out[idx] = v;
}

}  // namespace

extern "C" int lfdm_step_cond_f32(const float* step_table, const float* batch_base,
                                  const int32_t* step_dev, float* out, int batch, int n,
                                  lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!step_table || !batch_base || !step_dev || !out || batch <= 0 || n <= 0) {
    lfdm_set_error("step_cond: bad arguments");
    return LFDM_EINVAL;
  }
  int64_t nb = ((int64_t)batch * n + 255) / 256;
  if (nb > 1024) nb = 1024;
  LFDM_LAUNCH(step_cond_kernel, dim3((unsigned)nb), dim3(256), 0, stream, step_table, batch_base,
              step_dev, out, batch, n);
  return lfdm_check_launch("step_cond");
}

extern "C" int lfdm_linear_small_f32(const float* x, const float* w, const float* bias, float* y,
                                     int batch, int k, int n, int ldx, int ldw, int ldy,
                                     int act_in, int act_out, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !w || !y || batch <= 0 || k <= 0 || n <= 0 || ldx < k || ldw < k || ldy < n) {
    lfdm_set_error("linear_small: bad arguments");
    return LFDM_EINVAL;
  }
  LFDM_LAUNCH(linear_small_kernel, dim3((n + 3) / 4, batch), dim3(256), 0, stream, x, w, bias, y, k,
              n, ldx, ldw, ldy, act_in, act_out);
  return lfdm_check_launch("linear_small");
}

extern "C" int lfdm_sinusoidal_f32(const int32_t* t_dev, int t_stride, const float* freqs,
                                   float* out, int batch, int dim, int ldo, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!t_dev || !freqs || !out || batch <= 0 || dim < 4 || (dim & 1) || ldo < dim) {
    lfdm_set_error("sinusoidal: bad arguments");
    return LFDM_EINVAL;
  }
  LFDM_LAUNCH(sinusoidal_kernel, dim3(batch), dim3(64), 0, stream, t_dev, t_stride, freqs, out, dim, ldo);
  return lfdm_check_launch("sinusoidal");
}

// The same convolution on the matrix pipe: M = pixels, N = 64 output channels, K = kh*kw*cin (147 for the UNet's 7x7 stem over the
// three step-dependent channels) on v_mfma_f32_32x32x2_f32.  One workgroup = a 4-row x 32-column patch of one frame; its input
// window ((4 + kh - 1) x (32 + kw - 1) per channel, zeros outside the image) and the 64-column filter slice sit in LDS; wavefront
// r owns output row r of the patch: lane (pixel l&31, k-slot l>>5) reads its A element lds[c][r + ky][pixel + kx] - consecutive
// lanes, consecutive addresses - and the two B elements w[k][32*nt + l&31].  The k index walks (ky, kx, c) in the weight's own
// order; the D layout has lane = channel, so a register of the accumulator is one pixel's 128-byte row segment: stores need no
// transposition.  (The VALU form above: 42 us for the sampler's stem; it spends 64 FMAs per loaded input value on one lane.)
constexpr int CPM_ROWS = 4, CPM_COLS = 32;
__global__ __launch_bounds__(256) void conv_planar_in_mfma_kernel(
    const float* __restrict__ x, int batch, int cin, int cin_total, int frames, int h, int w,
    const float* __restrict__ wgt, int kh, int kw, int cout, const float* __restrict__ bias,
    const float* __restrict__ add_term, float* __restrict__ out, int ldo, int act, int tiles_x, int tiles_y) {
  __shared__ __attribute__((aligned(16))) float ws[(CPI_MAX_K + 1) * 64];
  __shared__ float win[8 * (CPM_ROWS + 6) * (CPM_COLS + 6)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, khalf = lane >> 5;
  const int K = kh * kw * cin;
  const int Kp = (K + 1) & ~1;
  const int co0 = blockIdx.y * 64;
  const int tile = blockIdx.x % (tiles_x * tiles_y);
  const int bt = blockIdx.x / (tiles_x * tiles_y);
  const int b = bt / frames, t = bt - b * frames;
  const int oy0 = (tile / tiles_x) * CPM_ROWS, ox0 = (tile % tiles_x) * CPM_COLS;
  const int py = kh / 2, px = kw / 2;
  const int wr = CPM_ROWS + kh - 1, wc = CPM_COLS + kw - 1;
  // filter slice [Kp][64] (row K of an odd K: zeros) and input window (zero outside the image): every load of the workgroup is
  // requested before the first one is written to LDS (the first version waited for each of its ~15 round trips in turn)
  constexpr int WMAX = ((CPI_MAX_K + 1) * 16 + 255) / 256;                       // float4 items per thread
  constexpr int XMAX = (8 * (CPM_ROWS + 6) * (CPM_COLS + 6) + 255) / 256;
  float4 wv[WMAX];
  float xv[XMAX];
#pragma unroll
  for (int j = 0; j < WMAX; ++j) {
    const int i = tid + 256 * j, kk = i >> 4, j4 = i & 15;
    wv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kk < K) wv[j] = *reinterpret_cast<const float4*>(wgt + (int64_t)kk * cout + co0 + 4 * j4);
  }
  const int hw = h * w;
  const int nwin = cin * wr * wc;
#pragma unroll
  for (int j = 0; j < XMAX; ++j) {
    const int i = tid + 256 * j;
    const int c = i / (wr * wc), rem = i - c * (wr * wc);
    const int ry = rem / wc, rx = rem - ry * wc;
    const int iy = oy0 + ry - py, ix = ox0 + rx - px;
    xv[j] = 0.f;
    if (i < nwin && iy >= 0 && iy < h && ix >= 0 && ix < w) xv[j] = x[(((int64_t)b * cin_total + c) * frames + t) * hw + iy * w + ix];
  }
#pragma unroll
  for (int j = 0; j < WMAX; ++j) {
    const int i = tid + 256 * j;
    if (i < Kp * 16) *reinterpret_cast<float4*>(ws + (i >> 4) * 64 + 4 * (i & 15)) = wv[j];
  }
#pragma unroll
  for (int j = 0; j < XMAX; ++j) {
    const int i = tid + 256 * j;
    if (i < nwin) win[i] = xv[j];
  }
  __syncthreads();
  f32x16 acc[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  // The contraction order is free: k-slot 0 (lanes 0-31) walks k = 0 .. Kp/2-1, k-slot 1 the upper half - each lane half then steps
  // through (ky, kx, c) in the weight's own order by +1 with two selects (the first version interleaved the slots, k = 2*step + slot,
  // and spent ~700 cycles per step in divergent carry loops: 32 us for this kernel, no faster than the VALU form)
  const int half_k = Kp >> 1;
  if (kh == 7 && kw == 7 && cin == 3) {
    // The UNet's stem (7x7 over the three step-dependent channels, K = 147): every index of the walk is a compile-time constant - the
    // generic loop below spends ~10 dependent VALU instructions and three LDS reads with computed addresses per step, which the 128
    // matrix-pipe cycles of a step do not cover with one or two workgroups per CU (round 6: 31 -> see profiles/r06_k_*).  Fully unrolled:
    // one select between the two k-slots' window offsets per step, LDS reads at immediate offsets the compiler can issue ahead.
    constexpr int KH = 7, KW = 7, CIN = 3, WR = CPM_ROWS + KH - 1, WC = CPM_COLS + KW - 1, HALF = (KH * KW * CIN + 1) / 2;
    const float* wbase = win + wave * WC + l31;
    const float* bbase = ws + (khalf * HALF) * 64 + l31;
#pragma unroll
    for (int st = 0; st < HALF; ++st) {
      const int k0 = st, k1 = HALF + st;
      const int ky0 = k0 / (KW * CIN), kx0 = (k0 - ky0 * KW * CIN) / CIN, c0 = k0 - (ky0 * KW + kx0) * CIN;
      const int ky1r = k1 / (KW * CIN), kx1 = (k1 - ky1r * KW * CIN) / CIN, c1 = k1 - (ky1r * KW + kx1) * CIN;
      const int ky1 = ky1r < KH ? ky1r : KH - 1;                  // (index K of the odd K: any valid address, its filter row is zero)
      const int o0 = (c0 * WR + ky0) * WC + kx0, o1 = (c1 * WR + ky1) * WC + kx1;
      const float a = wbase[khalf ? o1 : o0];
      const float b0 = bbase[st * 64], b1 = bbase[st * 64 + 32];
      acc[0] = mfma_32x32x2(a, b0, acc[0]);
      acc[1] = mfma_32x32x2(a, b1, acc[1]);
    }
  } else {
  int k = khalf * half_k;
  int ky = k / (kw * cin), kx = (k - ky * kw * cin) / cin, c = k - (ky * kw + kx) * cin;
  for (int st = 0; st < half_k; ++st, ++k) {
    const int kyc = ky < kh ? ky : kh - 1;                     // (index K of an odd K: any valid address, its filter row is zero)
    const float a = win[(c * wr + wave + kyc) * wc + l31 + kx];
    const float b0 = ws[k * 64 + l31], b1 = ws[k * 64 + 32 + l31];
    acc[0] = mfma_32x32x2(a, b0, acc[0]);
    acc[1] = mfma_32x32x2(a, b1, acc[1]);
    const bool cw_ = c + 1 == cin;
    c = cw_ ? 0 : c + 1;
    const bool kw_ = cw_ && kx + 1 == kw;
    kx = kw_ ? 0 : (cw_ ? kx + 1 : kx);
    ky = kw_ ? ky + 1 : ky;
  }
  }
  const int oy = oy0 + wave;
  if (oy >= h) return;
  // the 32 add-term values of this lane are requested together (one load -> wait -> store round trip per value cost more than the
  // whole contraction), then bias / activation / 128-byte row-segment stores
  float addv[2][16];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      const int oxc = ox < w ? ox : w - 1;
      addv[nt][r] = add_term ? add_term[((int64_t)b * hw + oy * w + oxc) * cout + co0 + 32 * nt + l31] : 0.f;
    }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int ch = co0 + 32 * nt + l31;
    const float bv = bias ? bias[ch] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      if (ox >= w) continue;
      const float v = acc[nt][r] + bv + addv[nt][r];
      out[(((int64_t)b * frames + t) * hw + oy * w + ox) * ldo + ch] = apply_act(v, act);
    }
  }
}

extern "C" int lfdm_conv_planar_in_cl_f32(const float* x, int batch, int cin, int cin_total,
                                          int frames, int h, int w, const float* wgt, int kh,
                                          int kw, int cout, const float* bias,
                                          const float* add_term, float* out, int ldo, int act,
                                          lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !wgt || !out || batch <= 0 || cin <= 0 || cin > cin_total || frames <= 0 || h <= 0 ||
      w <= 0 || kh <= 0 || kw <= 0 || !(kh & 1) || !(kw & 1) || cout <= 0 || cout % 64 != 0 ||
      kh * kw * cin > CPI_MAX_K || ldo < cout) {
    lfdm_set_error("conv_planar_in: unsupported shape (odd kernel, kh*kw*cin<=196, cout%64==0)");
    return LFDM_EINVAL;
  }
  const int64_t total = (int64_t)batch * frames * h * w;
  static const bool mfma_off = [] { const char* e = lfdm_knob("LFDM_STEM_MFMA"); return e && e[0] == '0'; }();
  const int tiles_x = (w + CPM_COLS - 1) / CPM_COLS, tiles_y = (h + CPM_ROWS - 1) / CPM_ROWS;
  const int64_t nblk = (int64_t)batch * frames * tiles_x * tiles_y;
  if (!mfma_off && cin <= 8 && kh <= 7 && kw <= 7 && (cout & 3) == 0 && ((((uintptr_t)wgt) & 15) == 0) && nblk < (1ll << 31)) {
    LFDM_LAUNCH(conv_planar_in_mfma_kernel, dim3((unsigned)nblk, cout / 64), dim3(256), 0, stream, x, batch, cin, cin_total, frames, h,
                w, wgt, kh, kw, cout, bias, add_term, out, ldo, act, tiles_x, tiles_y);
    return lfdm_check_launch("conv_planar_in_mfma");
  }
  const dim3 grid((unsigned)((total + CPI_PIX - 1) / CPI_PIX), cout / 64);
  LFDM_LAUNCH(conv_planar_in_kernel, grid, dim3(256), 0, stream, x, batch, cin, cin_total, frames, h, w, wgt, kh, kw, cout, bias,
              add_term, out, ldo, act);
  return lfdm_check_launch("conv_planar_in");
}

extern "C" int lfdm_heads_cl_to_planar_f32(const float* y_flow, const float* y_occ, int channels, int ld,
                                           const float* w_flow, const float* b_flow,
                                           const float* w_occ, const float* b_occ, float* out,
                                           int batch, int frames, int hw, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!y_flow || !y_occ || !w_flow || !b_flow || !w_occ || !b_occ || !out || channels <= 0 ||
      channels > 256 || channels % 4 != 0 || ld < channels || ld % 4 != 0 || batch <= 0 || frames <= 0 || hw <= 0) {
    lfdm_set_error("heads: bad arguments");
    return LFDM_EINVAL;
  }
  const int64_t total = (int64_t)batch * frames * hw;
  LFDM_LAUNCH(heads_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, y_flow,
              y_occ, channels, ld, w_flow, b_flow, w_occ, b_occ, out, batch, frames, hw, (const float*)nullptr, 0, 0, (const float*)nullptr, 0, 0,
              (const float*)nullptr);
  return lfdm_check_launch("heads");
}

extern "C" int lfdm_heads_res_cl_to_planar_f32(const float* y_flow, const float* y_occ, int channels, int ld, const float* w_flow,
                                               const float* b_flow, const float* w_occ, const float* b_occ, const float* x0, int ld0, int c0,
                                               const float* x1, int ld1, int c1, const float* w_extra, float* out, int batch, int frames,
                                               int hw, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!y_flow || !y_occ || !w_flow || !b_flow || !w_occ || !b_occ || !out || !x0 || !w_extra || channels <= 0 || channels > 256 ||
      channels % 4 != 0 || ld < channels || ld % 4 != 0 || batch <= 0 || frames <= 0 || hw <= 0 || c0 <= 0 || c0 % 4 != 0 || ld0 < c0 ||
      ld0 % 4 != 0 || c1 < 0 || c1 % 4 != 0 || (c1 > 0 && (!x1 || ld1 < c1 || ld1 % 4 != 0)) || c0 + c1 > 512 ||
      (((uintptr_t)y_flow | (uintptr_t)y_occ | (uintptr_t)x0 | (uintptr_t)x1) & 15) != 0) {
    lfdm_set_error("heads_res: bad arguments (C % 4 == 0, c0 + c1 <= 512, 16-byte aligned rows)");
    return LFDM_EINVAL;
  }
  const int64_t total = (int64_t)batch * frames * hw;
  LFDM_LAUNCH(heads_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, y_flow, y_occ, channels, ld, w_flow, b_flow,
              w_occ, b_occ, out, batch, frames, hw, x0, ld0, c0, x1 ? x1 : x0, ld1, c1, w_extra);
  return lfdm_check_launch("heads_res");
}

extern "C" int lfdm_heads_gn_res_cl_to_planar_f32(const float* y, int ld, int channels, const float* partial, int nchunk, int groups,
                                                  const float* gamma, const float* beta, float eps, const float* w_flow, const float* b_flow,
                                                  const float* w_occ, const float* b_occ, const float* x0, int ld0, int c0, const float* x1, int ld1,
                                                  int c1, const float* w_extra, float* out, int batch, int frames, int hw, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int c2 = 2 * channels;
  if (!y || !partial || !gamma || !beta || !w_flow || !b_flow || !w_occ || !b_occ || !out || !x0 || !w_extra || channels <= 0 || c2 > 512 ||
      channels % 4 != 0 || ld < c2 || ld % 4 != 0 || batch <= 0 || batch > 65535 || frames <= 0 || hw <= 0 || nchunk <= 0 || groups <= 0 || groups > 64 ||
      c2 % groups != 0 || c0 <= 0 || c0 % 4 != 0 || ld0 < c0 || ld0 % 4 != 0 || c1 < 0 || c1 % 4 != 0 ||
      (c1 > 0 && (!x1 || ld1 < c1 || ld1 % 4 != 0)) || c0 + c1 > 512 || (((uintptr_t)y | (uintptr_t)x0 | (uintptr_t)x1) & 15) != 0) {
    lfdm_set_error("heads_gn_res: bad arguments (2C <= 512 and % groups == 0, C % 4 == 0, c0 + c1 <= 512, 16-byte aligned rows, <= 65535 samples)");
    return LFDM_EINVAL;
  }
  const int64_t pixels = (int64_t)frames * hw;
  const int rows_per_wg = pixels >= 32768 ? 128 : 32;                         // four passes per workgroup once that still leaves >= 256 workgroups per sample
  LFDM_LAUNCH(heads_gn_kernel, dim3((unsigned)((pixels + rows_per_wg - 1) / rows_per_wg), (unsigned)batch), dim3(256), 0, stream, y, ld, channels, partial,
              nchunk, groups, gamma, beta, eps, w_flow, b_flow, w_occ, b_occ, x0, ld0, c0, x1 ? x1 : x0, ld1, c1, w_extra, out, frames, hw, rows_per_wg);
  return lfdm_check_launch("heads_gn_res");
}

extern "C" int lfdm_pack_conv_weight_f32(const float* w, int n_o, int n_i, int taps, int64_t stride_o, int64_t stride_i,
                                         int mode, float* out, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!w || !out || n_o <= 0 || n_i <= 0 || taps <= 0 || mode < 0 || mode > 2 || (mode == 2 && taps != 16) ||
      stride_o <= 0 || stride_i <= 0) {
    lfdm_set_error("pack_conv_weight: bad arguments (mode 0 conv, 1 data gradient, 2 ConvTranspose k4 s2 p1 parity packs)");
    return LFDM_EINVAL;
  }
  const int K = mode == 0 ? taps * n_i : mode == 1 ? taps * n_o : 4 * n_o;
  const int N = mode == 0 ? n_o : n_i;
  const int np = (N + 31) / 32 * 32;
  const int64_t total = (int64_t)((K + 31) / 32) * np * 32 * (mode == 2 ? 4 : 1);
  LFDM_LAUNCH(pack_conv_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, n_o, n_i, taps,
              stride_o, stride_i, mode, K, N, np, out);
  return lfdm_check_launch("pack_conv_weight");
}
