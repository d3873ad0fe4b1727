// Sampler step on the planar latent (include/lfdm_hip.h: lfdm_sampler_step_f32):
//   x0 = c_x*x - c_eps*eps;  s = max(1, quantile_0.9(|x0|) per sample);  x0 = clamp(x0,-s,s)/s;
//   x <- k_x0*x0 + k_eps*eps + k_x*x + k_noise*noise
// (reference GaussianDiffusion.ddim_sample :791-827, p_sample/p_mean_variance :712-746).
// torch.quantile sorts; here the two order statistics the linear interpolation needs are found
// exactly by a 3-pass (11/11/10 bit) radix select on the IEEE bit pattern of |x0|, with
// LDS-privatised histograms merged by integer atomics (order independent -> deterministic).
// All step-dependent scalars come from a device table indexed by a device counter so the captured
// hipGraph of one step can be replayed for every step.
#include <stdlib.h>
#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

constexpr int NB = 2048;  // bins per histogram (11 bits)
// workspace layout per sample: hist0[NB] | h1a[NB] | h1b[NB] | h2a[NB] | h2b[NB]  (uint32)
constexpr int HIST_PER_SAMPLE = 5 * NB;

struct Ranks {
  unsigned lo, hi;  // zero-based ranks of the two order statistics
  float frac;
};

// Finds the bin holding zero-based rank `k` in hist[0..nbins) and the rank inside that bin.
// All 256 threads call it; result is broadcast through LDS.
__device__ void find_bin(const unsigned* hist, int nbins, unsigned k, unsigned& bin, unsigned& krem,
                         unsigned* s_part /*[256]*/, unsigned* s_res /*[2]*/) {
  // one wavefront: each lane sums nbins/64 consecutive bins, a shuffle scan locates the lane holding rank k, that lane
  // walks its own bins.  (The first version let thread 0 walk 256 partial sums serially - 6 of these per workgroup cost
  // more than streaming the data.)
  (void)s_part;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const int per = nbins / 64;
    unsigned local = 0;
    for (int i = 0; i < per; ++i) local += hist[lane * per + i];
    unsigned incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned up = __shfl(incl, lane >= d ? lane - d : lane);
      if (lane >= d) incl += up;
    }
    const unsigned excl = incl - local;
    if (k >= excl && k < incl) {
      unsigned run = excl;
      int b = lane * per;
      for (int i = 0; i < per; ++i) {
        const unsigned c = hist[lane * per + i];
        b = lane * per + i;
        if (run + c > k) break;
        run += c;
      }
      s_res[0] = (unsigned)b;
      s_res[1] = k - run;
    }
  }
  __syncthreads();
  bin = s_res[0];
  krem = s_res[1];
  __syncthreads();
}

__device__ __forceinline__ void flush_hist(unsigned* lds, unsigned* glob, int nbins) {
  for (int i = threadIdx.x; i < nbins; i += 256) {
    const unsigned c = lds[i];
    if (c) atomicAdd(glob + i, c);
  }
}

// pass 0: x0 (optional) + histogram of bits [31:21].  grid (nblk, B)
__global__ __launch_bounds__(256) void quantile_pass0_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ eps,
                                                             float* __restrict__ x0buf, int64_t n,
                                                             const float* __restrict__ coef,
                                                             const int32_t* __restrict__ step_dev,
                                                             unsigned* __restrict__ hists) {
  __shared__ unsigned h[NB];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < NB; i += 256) h[i] = 0;
  __syncthreads();
  float cx = 1.f, ce = 0.f;
  if (coef) {
    const float* c = coef + (int64_t)(*step_dev) * 6;
    cx = c[0];
    ce = c[1];
  }
  const float* xb = x + (int64_t)b * n;
  const float* eb = eps ? eps + (int64_t)b * n : nullptr;
  float* ob = x0buf ? x0buf + (int64_t)b * n : nullptr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v = xb[i];
    if (eb) v = cx * v - ce * eb[i];
    if (ob) ob[i] = v;
    const unsigned u = __float_as_uint(fabsf(v));
    atomicAdd(&h[u >> 21], 1u);
  }
  __syncthreads();
  flush_hist(h, hists + (int64_t)b * HIST_PER_SAMPLE, NB);
}

// pass 1 (shift 10, 11 bits) and pass 2 (shift 0, 10 bits).  grid (nblk, B)
template <int PASS>
__global__ __launch_bounds__(256) void quantile_pass_kernel(const float* __restrict__ v, int64_t n,
                                                            Ranks rk, unsigned* __restrict__ hists) {
  __shared__ unsigned ha[NB], hb[NB];
  __shared__ unsigned s_part[256], s_res[2];
  const int b = blockIdx.y;
  unsigned* hs = hists + (int64_t)b * HIST_PER_SAMPLE;
  for (int i = threadIdx.x; i < NB; i += 256) { ha[i] = 0; hb[i] = 0; }
  unsigned pa, pb, ka, kb;
  find_bin(hs, NB, rk.lo, pa, ka, s_part, s_res);
  find_bin(hs, NB, rk.hi, pb, kb, s_part, s_res);
  if (PASS == 2) {
    unsigned qa, qb, t0, t1;
    find_bin(hs + NB, NB, ka, qa, t0, s_part, s_res);
    find_bin(hs + 2 * NB, NB, kb, qb, t1, s_part, s_res);
    pa = (pa << 11) | qa;
    pb = (pb << 11) | qb;
  }
  __syncthreads();
  const float* vb = v + (int64_t)b * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const unsigned u = __float_as_uint(fabsf(vb[i]));
    if (PASS == 1) {
      const unsigned top = u >> 21, mid = (u >> 10) & 2047u;
      if (top == pa) atomicAdd(&ha[mid], 1u);
      if (top == pb) atomicAdd(&hb[mid], 1u);
    } else {
      const unsigned top = u >> 10, low = u & 1023u;
      if (top == pa) atomicAdd(&ha[low], 1u);
      if (top == pb) atomicAdd(&hb[low], 1u);
    }
  }
  __syncthreads();
  flush_hist(ha, hs + (PASS == 1 ? 1 : 3) * NB, NB);
  flush_hist(hb, hs + (PASS == 1 ? 2 : 4) * NB, NB);
}

// resolves the two order statistics from the histograms (all threads of a block)
__device__ float resolve_quantile(const unsigned* hs, Ranks rk, unsigned* s_part, unsigned* s_res) {
  unsigned a0, b0, ka, kb, a1, b1, a2, b2, t;
  find_bin(hs, NB, rk.lo, a0, ka, s_part, s_res);
  find_bin(hs, NB, rk.hi, b0, kb, s_part, s_res);
  find_bin(hs + NB, NB, ka, a1, ka, s_part, s_res);
  find_bin(hs + 2 * NB, NB, kb, b1, kb, s_part, s_res);
  find_bin(hs + 3 * NB, NB, ka, a2, t, s_part, s_res);
  find_bin(hs + 4 * NB, NB, kb, b2, t, s_part, s_res);
  const float lo = __uint_as_float((a0 << 21) | (a1 << 10) | a2);
  const float hi = __uint_as_float((b0 << 21) | (b1 << 10) | b2);
  return lo + rk.frac * (hi - lo);  // aten lerp form for weight < 0.5
}

__global__ __launch_bounds__(256) void quantile_out_kernel(Ranks rk, const unsigned* __restrict__ hists,
                                                           float* __restrict__ q_out) {
  __shared__ unsigned s_part[256], s_res[2];
  const int b = blockIdx.x;
  const float q = resolve_quantile(hists + (int64_t)b * HIST_PER_SAMPLE, rk, s_part, s_res);
  if (threadIdx.x == 0) q_out[b] = q;
}

// grid (nblk, B)
__global__ __launch_bounds__(256) void sampler_update_kernel(float* __restrict__ x,
                                                             const float* __restrict__ eps,
                                                             const float* __restrict__ noise,
                                                             float* __restrict__ x0buf,
                                                             float* __restrict__ x0_out, int64_t n,
                                                             const float* __restrict__ coef,
                                                             const int32_t* __restrict__ step_dev,
                                                             Ranks rk, const unsigned* __restrict__ hists) {
  __shared__ unsigned s_part[256], s_res[2];
  const int b = blockIdx.y;
  float s = resolve_quantile(hists + (int64_t)b * HIST_PER_SAMPLE, rk, s_part, s_res);
  s = fmaxf(s, 1.0f);
  const float* c = coef + (int64_t)(*step_dev) * 6;
  const float k_x0 = c[2], k_eps = c[3], k_x = c[4], k_noise = c[5];
  float* xb = x + (int64_t)b * n;
  const float* eb = eps + (int64_t)b * n;
  const float* nb = noise ? noise + (int64_t)b * n : nullptr;
  const float* x0b = x0buf + (int64_t)b * n;
  float* x0o = x0_out ? x0_out + (int64_t)b * n : nullptr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float x0 = x0b[i];
    x0 = fminf(fmaxf(x0, -s), s) / s;
    if (x0o) x0o[i] = x0;
    float v = k_x0 * x0 + k_eps * eb[i];
    if (k_x != 0.f) v += k_x * xb[i];
    if (k_noise != 0.f && nb) v += k_noise * nb[i];
    xb[i] = v;
  }
}

// classifier-free guidance: out = null + (cond - null) * scale   (reference :525-526)
__global__ __launch_bounds__(256) void cfg_combine_kernel(const float* __restrict__ cond_eps,
                                                          const float* __restrict__ null_eps,
                                                          float scale, float* __restrict__ out,
                                                          int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float a = null_eps[i];
    out[i] = a + (cond_eps[i] - a) * scale;
  }
}

// Histogram clear.  NOT hipMemsetAsync: a memset node inside a captured hipGraph stopped writing zeros
// after a few replays on ROCm 7.2 / gfx950 (it filled a stale 32-bit pattern instead), which silently
// corrupted every quantile of the second video onwards; a plain kernel node replays correctly.
__global__ __launch_bounds__(256) void zero_u32_kernel(unsigned* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0u;
}

__global__ void advance_step_kernel(int32_t* step_dev) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *step_dev += 1;
}

Ranks make_ranks(int64_t n, float quantile) {
  // torch.quantile: rank = q * (n - 1) evaluated in the input dtype (fp32), then floor / lerp
  const float pos = quantile * (float)(n - 1);
  const float fl = floorf(pos);
  Ranks r;
  r.lo = (unsigned)fl;
  r.hi = r.lo + 1 < (unsigned)n ? r.lo + 1 : (unsigned)(n - 1);
  r.frac = pos - fl;
  return r;
}

unsigned blocks_for(int64_t n) {
  int64_t nb = (n + 256 * 8 - 1) / (256 * 8);
  if (nb < 1) nb = 1;
  if (nb > 512) nb = 512;
  return (unsigned)nb;
}

int run_select(const float* v, int batch, int64_t n, Ranks rk, unsigned* hists, hipStream_t stream) {
  const dim3 grid(blocks_for(n), batch), block(256);
  LFDM_LAUNCH((quantile_pass_kernel<1>), grid, block, 0, stream, v, n, rk, hists);
  LFDM_LAUNCH((quantile_pass_kernel<2>), grid, block, 0, stream, v, n, rk, hists);
  return 0;
}

}  // namespace

extern "C" int lfdm_cfg_combine_f32(const float* cond_eps, const float* null_eps, float scale,
                                    float* out, int64_t n, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!cond_eps || !null_eps || !out || n <= 0) {
    lfdm_set_error("cfg_combine: bad arguments");
    return LFDM_EINVAL;
  }
  int64_t nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  LFDM_LAUNCH(cfg_combine_kernel, dim3((unsigned)nb), dim3(256), 0, stream, cond_eps, null_eps, scale,
              out, n);
  return lfdm_check_launch("cfg_combine");
}

extern "C" size_t lfdm_sampler_ws_bytes(int batch, int64_t n) {
  return (size_t)batch * HIST_PER_SAMPLE * sizeof(unsigned) + (size_t)batch * (size_t)n * sizeof(float);
}

extern "C" int lfdm_abs_quantile_f32(const float* x, int batch, int64_t n, float quantile,
                                     float* q_out, void* ws, size_t ws_bytes, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !q_out || batch <= 0 || n <= 0 || n >= (1 << 24) || quantile < 0.f || quantile > 1.f) {
    lfdm_set_error("abs_quantile: bad arguments (n < 2^24)");
    return LFDM_EINVAL;
  }
  if (!ws || ws_bytes < (size_t)batch * HIST_PER_SAMPLE * sizeof(unsigned)) {
    lfdm_set_error("abs_quantile: workspace too small");
    return LFDM_EWORKSPACE;
  }
  unsigned* hists = reinterpret_cast<unsigned*>(ws);
  {
    const int64_t nz = (int64_t)batch * HIST_PER_SAMPLE;
    LFDM_LAUNCH(zero_u32_kernel, dim3((unsigned)((nz + 255) / 256 > 64 ? 64 : (nz + 255) / 256)), dim3(256), 0, stream,
                hists, nz);
  }
  const Ranks rk = make_ranks(n, quantile);
  LFDM_LAUNCH(quantile_pass0_kernel, dim3(blocks_for(n), batch), dim3(256), 0, stream, x,
              (const float*)nullptr, (float*)nullptr, n, (const float*)nullptr,
              (const int32_t*)nullptr, hists);
  run_select(x, batch, n, rk, hists, stream);
  LFDM_LAUNCH(quantile_out_kernel, dim3(batch), dim3(256), 0, stream, rk, (const unsigned*)hists, q_out);
  return lfdm_check_launch("abs_quantile");
}

extern "C" int lfdm_sampler_step_f32(float* x, const float* eps, const float* noise, float* x0_out,
                                     int batch, int64_t n, const float* coef, int32_t* step_dev,
                                     float quantile, int advance, void* ws, size_t ws_bytes,
                                     lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !eps || !coef || !step_dev || batch <= 0 || n <= 0 || n >= (1 << 24)) {
    lfdm_set_error("sampler_step: bad arguments (n < 2^24)");
    return LFDM_EINVAL;
  }
  if (!ws || ws_bytes < lfdm_sampler_ws_bytes(batch, n)) {
    lfdm_set_error("sampler_step: workspace too small");
    return LFDM_EWORKSPACE;
  }
  unsigned* hists = reinterpret_cast<unsigned*>(ws);
  float* x0buf = reinterpret_cast<float*>(hists + (size_t)batch * HIST_PER_SAMPLE);
  const Ranks rk = make_ranks(n, quantile);
  {
    const int64_t nz = (int64_t)batch * HIST_PER_SAMPLE;
    LFDM_LAUNCH(zero_u32_kernel, dim3((unsigned)((nz + 255) / 256 > 64 ? 64 : (nz + 255) / 256)), dim3(256), 0, stream,
                hists, nz);
  }
  const dim3 grid(blocks_for(n), batch), block(256);
  LFDM_LAUNCH(quantile_pass0_kernel, grid, block, 0, stream, (const float*)x, eps, x0buf, n, coef,
              (const int32_t*)step_dev, hists);
  run_select(x0buf, batch, n, rk, hists, stream);
  LFDM_LAUNCH(sampler_update_kernel, grid, block, 0, stream, x, eps, noise, x0buf, x0_out, n, coef,
              (const int32_t*)step_dev, rk, (const unsigned*)hists);
  if (advance) LFDM_LAUNCH(advance_step_kernel, dim3(1), dim3(64), 0, stream, step_dev);
  return lfdm_check_launch("sampler_step");
}
