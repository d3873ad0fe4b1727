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

// A 2048-bin histogram held eight consecutive bins per thread (256 threads).  load_bins only ISSUES the two 16-byte loads: the
// callers request every histogram they will need before the first one is searched, so a workgroup pays the global round trip
// once instead of once per search (the histograms were written by the previous kernel's atomics).
struct Bins {
  uint4 a, b;
};
__device__ __forceinline__ Bins load_bins(const unsigned* hist) {
  const uint4* p = reinterpret_cast<const uint4*>(hist) + 2 * threadIdx.x;
  Bins h;
  h.a = p[0];
  h.b = p[1];
  return h;
}

// Finds, for up to two zero-based ranks (k[0], k[1]; NK = 1 or 2) of the same histogram, the bin holding the rank and the rank
// inside that bin.  All 256 threads call it: wave-level shuffle scans of the per-thread sums, the four wave totals through LDS,
// the owning thread walks its eight registers.  (The first version let thread 0 walk 256 partial sums serially, the second one
// wavefront walk 32 bins per lane with dependent loads - six of those per workgroup cost more than streaming the data.)
template <int NK>
__device__ void find_bins(const Bins& h, const unsigned (&k)[NK], unsigned (&bin)[NK], unsigned (&krem)[NK],
                          unsigned* s_part /*[256]*/, unsigned* s_res /*[4]*/) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned v[8] = {h.a.x, h.a.y, h.a.z, h.a.w, h.b.x, h.b.y, h.b.z, h.b.w};
  const unsigned local = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  unsigned incl = local;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned up = __shfl(incl, lane >= d ? lane - d : lane);
    if (lane >= d) incl += up;
  }
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();
  unsigned base = 0;
#pragma unroll
  for (int w = 0; w < 3; ++w)
    if (w < wave) base += s_part[w];
  incl += base;
  const unsigned excl = incl - local;
#pragma unroll
  for (int q = 0; q < NK; ++q)
    if (k[q] >= excl && k[q] < incl) {
      unsigned run = excl, bsel = 7, rsel = 0;
      bool found = false;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (!found && run + v[i] > k[q]) {
          found = true;
          bsel = (unsigned)i;
          rsel = k[q] - run;
        }
        if (!found) run += v[i];
      }
      s_res[2 * q] = 8u * (unsigned)tid + bsel;
      s_res[2 * q + 1] = rsel;
    }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NK; ++q) {
    bin[q] = s_res[2 * q];
    krem[q] = s_res[2 * q + 1];
  }
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
  __shared__ unsigned s_part[256], s_res[4];
  const int b = blockIdx.y;
  unsigned* hs = hists + (int64_t)b * HIST_PER_SAMPLE;
  for (int i = threadIdx.x; i < NB; i += 256) { ha[i] = 0; hb[i] = 0; }
  const Bins h0 = load_bins(hs);
  Bins h1a = h0, h1b = h0;
  if (PASS == 2) {
    h1a = load_bins(hs + NB);
    h1b = load_bins(hs + 2 * NB);
  }
  unsigned pa, pb;
  {
    const unsigned k0[2] = {rk.lo, rk.hi};
    unsigned bin0[2], rem0[2];
    find_bins<2>(h0, k0, bin0, rem0, s_part, s_res);
    pa = bin0[0];
    pb = bin0[1];
    if (PASS == 2) {
      const unsigned ka[1] = {rem0[0]}, kb[1] = {rem0[1]};
      unsigned qa[1], qb[1], t[1];
      find_bins<1>(h1a, ka, qa, t, s_part, s_res);
      find_bins<1>(h1b, kb, qb, t, s_part, s_res);
      pa = (pa << 11) | qa[0];
      pb = (pb << 11) | qb[0];
    }
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
  const Bins h0 = load_bins(hs), h1a = load_bins(hs + NB), h1b = load_bins(hs + 2 * NB), h2a = load_bins(hs + 3 * NB),
             h2b = load_bins(hs + 4 * NB);                   // all five in flight before the first search
  const unsigned k0[2] = {rk.lo, rk.hi};
  unsigned bin0[2], rem0[2], a1[1], b1[1], a2[1], b2[1], ra[1], rb[1], t[1];
  find_bins<2>(h0, k0, bin0, rem0, s_part, s_res);
  const unsigned ka[1] = {rem0[0]}, kb[1] = {rem0[1]};
  find_bins<1>(h1a, ka, a1, ra, s_part, s_res);
  find_bins<1>(h1b, kb, b1, rb, s_part, s_res);
  find_bins<1>(h2a, ra, a2, t, s_part, s_res);
  find_bins<1>(h2b, rb, b2, t, s_part, s_res);
  const float lo = __uint_as_float((bin0[0] << 21) | (a1[0] << 10) | a2[0]);
  const float hi = __uint_as_float((bin0[1] << 21) | (b1[0] << 10) | b2[0]);
  return lo + rk.frac * (hi - lo);  // aten lerp form for weight < 0.5
}

__global__ __launch_bounds__(256) void quantile_out_kernel(Ranks rk, const unsigned* __restrict__ hists,
                                                           float* __restrict__ q_out) {
  __shared__ unsigned s_part[256], s_res[4];
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
                                                             const int32_t* step_dev,      // (no __restrict__: advance_dev is the same word)
                                                             Ranks rk, unsigned* __restrict__ hists, int hist_samples,
                                                             unsigned* __restrict__ ticket, int32_t* advance_dev) {
  __shared__ unsigned s_part[256], s_res[4];
  const int b = blockIdx.y;
  float* xb = x + (int64_t)b * n;
  const float* eb = eps + (int64_t)b * n;
  const float* nb = noise ? noise + (int64_t)b * n : nullptr;
  const float* x0b = x0buf + (int64_t)b * n;
  float* x0o = x0_out ? x0_out + (int64_t)b * n : nullptr;
  // The first PRE elements of this thread (all of them at the C2 latent: 122 880 / (60 x 256) = 8) are requested BEFORE the histograms are
  // searched: five scans with barriers stand between the kernel's start and the threshold, the operands' round trip runs under them (round 6).
  constexpr int PRE = 8;
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  float p_x0[PRE], p_e[PRE], p_x[PRE], p_n[PRE];
#pragma unroll
  for (int k = 0; k < PRE; ++k) {
    const int64_t i = i0 + k * stride;
    const bool in = i < n;
    p_x0[k] = in ? x0b[i] : 0.f;
    p_e[k] = in ? eb[i] : 0.f;
    p_x[k] = in ? xb[i] : 0.f;
    p_n[k] = (in && nb) ? nb[i] : 0.f;
  }
  float s = 1.0f;                       // rk.frac < 0: static clipping, x0.clamp(-1, 1) (use_dynamic_thres=False, :729-732)
  if (rk.frac >= 0.f) {
    s = resolve_quantile(hists + (int64_t)b * HIST_PER_SAMPLE, rk, s_part, s_res);
    s = fmaxf(s, 1.0f);
  }
  const float* c = coef + (int64_t)(*step_dev) * 6;
  const float k_x0 = c[2], k_eps = c[3], k_x = c[4], k_noise = c[5];
#pragma unroll
  for (int k = 0; k < PRE; ++k) {
    const int64_t i = i0 + k * stride;
    if (i < n) {
      float x0 = fminf(fmaxf(p_x0[k], -s), s) / s;
      if (x0o) x0o[i] = x0;
      float v = k_x0 * x0 + k_eps * p_e[k];
      if (k_x != 0.f) v += k_x * p_x[k];
      if (k_noise != 0.f && nb) v += k_noise * p_n[k];
      xb[i] = v;
    }
  }
  for (int64_t i = i0 + PRE * stride; i < n; i += stride) {
    float x0 = x0b[i];
    x0 = fminf(fmaxf(x0, -s), s) / s;
    if (x0o) x0o[i] = x0;
    float v = k_x0 * x0 + k_eps * eb[i];
    if (k_x != 0.f) v += k_x * xb[i];
    if (k_noise != 0.f && nb) v += k_noise * nb[i];
    xb[i] = v;
  }
  // End-of-step housekeeping by the workgroup that finishes last (every workgroup has read the histograms and the step counter before
  // it takes its ticket): clear the histograms for the next step and advance the step counter - two launches (zero_u32, advance_step)
  // of every replayed step otherwise.  The ticket word is left at zero again.
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned total = gridDim.x * gridDim.y;
    const bool last = lfdm_ticket_take(ticket) == total - 1;
    if (last) lfdm_ticket_reset(ticket);
    s_res[0] = last ? 1u : 0u;
  }
  __syncthreads();
  if (s_res[0]) {
    const int64_t nz = (int64_t)hist_samples * HIST_PER_SAMPLE;
    for (int64_t i = threadIdx.x; i < nz; i += 256) hists[i] = 0u;
    if (advance_dev && threadIdx.x == 0) *advance_dev += 1;
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
  return (size_t)batch * HIST_PER_SAMPLE * sizeof(unsigned) + (size_t)batch * (size_t)n * sizeof(float) + 64;     // + the ticket word
}

extern "C" int lfdm_sampler_ws_init(void* ws, size_t ws_bytes, int batch, int64_t n, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!ws || batch <= 0 || n <= 0 || ws_bytes < lfdm_sampler_ws_bytes(batch, n)) {
    lfdm_set_error("sampler_ws_init: workspace too small");
    return LFDM_EWORKSPACE;
  }
  unsigned* hists = reinterpret_cast<unsigned*>(ws);
  const int64_t nz = (int64_t)batch * HIST_PER_SAMPLE;
  LFDM_LAUNCH(zero_u32_kernel, dim3((unsigned)((nz + 255) / 256 > 64 ? 64 : (nz + 255) / 256)), dim3(256), 0, stream, hists, nz);
  unsigned* ticket = reinterpret_cast<unsigned*>(reinterpret_cast<float*>(hists + (size_t)batch * HIST_PER_SAMPLE) + (size_t)batch * n);
  LFDM_LAUNCH(zero_u32_kernel, dim3(1), dim3(256), 0, stream, ticket, (int64_t)16);
  return lfdm_check_launch("sampler_ws_init");
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
  {   // leave the histograms cleared: lfdm_sampler_step_f32 relies on that when it is handed the same workspace
    const int64_t nz = (int64_t)batch * HIST_PER_SAMPLE;
    LFDM_LAUNCH(zero_u32_kernel, dim3((unsigned)((nz + 255) / 256 > 64 ? 64 : (nz + 255) / 256)), dim3(256), 0, stream, hists, nz);
  }
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
  const bool dynamic = quantile >= 0.f;           // quantile < 0: static clipping to [-1, 1] (GaussianDiffusion's own default)
  if (dynamic && quantile > 1.f) {
    lfdm_set_error("sampler_step: quantile must lie in [0, 1] (or be negative for the static clamp to [-1, 1])");
    return LFDM_EINVAL;
  }
  Ranks rk = make_ranks(n, dynamic ? quantile : 0.f);
  if (!dynamic) rk.frac = -1.f;
  // the histograms are clear on entry: lfdm_sampler_ws_init, then the last workgroup of every update kernel - for one or two samples
  // (a single workgroup clearing more would lengthen the kernel's tail: larger batches keep the clearing launch)
  const bool fold_clear = batch <= 2;
  if (!fold_clear) {
    const int64_t nz = (int64_t)batch * HIST_PER_SAMPLE;
    LFDM_LAUNCH(zero_u32_kernel, dim3((unsigned)((nz + 255) / 256 > 64 ? 64 : (nz + 255) / 256)), dim3(256), 0, stream, hists, nz);
  }
  unsigned* ticket = reinterpret_cast<unsigned*>(x0buf + (size_t)batch * n);
  const dim3 grid(blocks_for(n), batch), block(256);
  LFDM_LAUNCH(quantile_pass0_kernel, grid, block, 0, stream, (const float*)x, eps, x0buf, n, coef,
              (const int32_t*)step_dev, hists);         // (also the x0 = c_x*x - c_eps*eps pass)
  if (dynamic) run_select(x0buf, batch, n, rk, hists, stream);
  LFDM_LAUNCH(sampler_update_kernel, grid, block, 0, stream, x, eps, noise, x0buf, x0_out, n, coef,
              (const int32_t*)step_dev, rk, hists, fold_clear ? batch : 0, ticket, advance ? step_dev : (int32_t*)nullptr);
  return lfdm_check_launch("sampler_step");
}
