// Native glue of LFAE stage-1 training (include/lfdm_hip.h, ABI version 10): what connects the convolutions of
// LFAE/modules/model.py:141-217 - everything the round-4 trainer left to MIOpen / composable_kernel / ATen kernels.
//
//  lfdm_batchnorm_train_{fwd,bwd}_cl_f32  nn.BatchNorm2d with BATCH statistics (+ ReLU) on channels-last rows (util.py:70-150: ResBlock2d,
//      UpBlock2d, DownBlock2d, SameBlock2d), forward and backward as TWO launches each: a row-chunk reduce whose last workgroup per
//      32-chunk group folds its group (ticket, fixed order: run-to-run identical), and an apply pass whose workgroups finish the merge for
//      their 64 channels.  Replaces 6 MIOpenBatchNorm* + clamp + threshold kernels x 62 layers per step.  HBM-bound: forward reads x
//      twice and writes y (12 B / element), backward reads x and dy twice and writes dx (20 B / element).
//  lfdm_blur_down_{fwd,bwd}_f32           AntiAliasInterpolation2d (util.py:217-264) and the ImagePyramide levels (model.py:62-82): depth-wise
//      Gaussian + every s-th pixel, any input / output strides (an NCHW image or the 4-channel rows a convolution wants), optional per-
//      channel affine epilogue (the VGG input normalisation, model.py:52).  Only kept outputs are computed.
//  lfdm_warp_bwd_f32 (+ absmax / fixed-point finalize / resize adjoint)   backward of deform_input + apply_optical (generator.py:59-88):
//      gradient of grid_sample (bilinear, zeros) w.r.t. the sampled tensor, the flow, the occlusion map and the blended tensor.  The
//      scatter into the sampled tensor's gradient accumulates 64-bit FIXED-POINT integers (integer atomics commute: the result does not
//      depend on the order the workgroups arrive in, unlike ATen's float atomics), scaled per call from max |dout| so that the rounding
//      step is ~2^-40 of that maximum.
//  lfdm_grid_sample_{fwd,bwd}_f32         F.grid_sample on small-channel planar tensors with an explicit grid (pixelwise_flow_predictor.py:95-
//      102 deformed sources - gradient w.r.t. the grid only; model.py:118-122 Transform.transform_frame with reflection padding).
//  lfdm_svd2x2_sym_bwd_f32                backward of torch.svd on the 2x2 region covariances (region_predictor.py:16-26), closed form.
#include <math.h>

#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

// =====================================================================================================================================
// BatchNorm (batch statistics) + ReLU
// =====================================================================================================================================
constexpr int BN_GROUP = 32;          // chunks folded by one level-1 finalizer
constexpr int BN_MAX_CHUNKS = 1024;   // -> at most 32 level-2 partials per channel
constexpr int BN_MIN_ROWS = 128;      // rows per chunk, at least

struct BnGeom {
  int q, lanes, tw, ctiles, nchunk, chunk_rows, ngroups;
};

BnGeom bn_geom(int64_t rows, int c) {
  BnGeom g;
  g.q = c > 32 ? 16 : (c > 16 ? 8 : 4);            // float4 lanes per row inside a workgroup
  g.lanes = 256 / g.q;
  g.tw = 4 * g.q;                                  // channels per workgroup column
  g.ctiles = (c + g.tw - 1) / g.tw;
  int64_t want = (rows + BN_MIN_ROWS - 1) / BN_MIN_ROWS;
  if (want < 1) want = 1;
  if (want > BN_MAX_CHUNKS) want = BN_MAX_CHUNKS;
  g.chunk_rows = (int)((rows + want - 1) / want);
  g.nchunk = (int)((rows + g.chunk_rows - 1) / g.chunk_rows);
  g.ngroups = (g.nchunk + BN_GROUP - 1) / BN_GROUP;
  return g;
}

struct BnArgs {
  const float* x;
  const float* dy;          // backward only
  const float* add;         // backward, may be null: a second gradient of x (the block's skip path), summed into dx; row stride ldadd
  int ldadd;
  float* out;               // y (forward) / dx (backward)
  int64_t rows;
  int c, ldx, lddy, ldo;
  const float* gamma;
  const float* beta;
  float* stat;              // [mean (c) | rstd (c)]: written by the forward, read by the backward
  float* running_mean;      // forward, may be null
  float* running_var;
  float momentum, eps;
  int relu;
  int q, chunk_rows, nchunk, ngroups;
  float* part1;             // [segment][ctile][chunk][tw] pairs (sum, sum) as 8-byte words
  double* part2;            // [segment][ctile][group][2][tw] double
  unsigned* tickets;        // [segment][ctile][group], zero between launches
  // segments: `segments` independent batches of `rows` rows stacked along the row axis, each normalised with its own statistics
  // (blockIdx.z); stat is [segment][2][c]; the running statistics take the segments' updates one after the other, and dgamma / dbeta
  // are the sums over the segments.  This is N module calls on N batches as one launch pair.
  int segments;
  int64_t seg_p1, seg_p2;   // strides of part1 / part2 per segment (elements)
  int seg_tk;
  float* dgamma;            // backward
  float* dbeta;
};

// MODE 0: (sum (x - x[row 0]), sum (x - x[row 0])^2).  MODE 1: (sum g, sum g * xhat) with g = dy masked by the ReLU of the recomputed forward.
// grid (nchunk, ctiles), 256 threads = `lanes` row lanes x q float4 columns.
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(BnArgs a) {
  __shared__ float sm[2][256 * 4];
  __shared__ int s_last;
  const int tid = threadIdx.x, q = a.q, lanes = 256 / q, tw = 4 * q;
  const int quad = tid & (q - 1), rl = tid / q;
  const int chunk = blockIdx.x, ct = blockIdx.y, seg = blockIdx.z;
  a.x += seg * a.rows * a.ldx;
  if (MODE == 1) {
    a.dy += seg * a.rows * a.lddy;
    a.stat += (int64_t)seg * 2 * a.c;
  }
  a.part1 += seg * a.seg_p1;
  a.part2 += seg * a.seg_p2;
  a.tickets += seg * a.seg_tk;
  const int ch = ct * tw + quad * 4;
  const bool cvalid = ch < a.c;
  const int64_t r0 = (int64_t)chunk * a.chunk_rows;
  const int64_t r1 = r0 + a.chunk_rows < a.rows ? r0 + a.chunk_rows : a.rows;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  float4 mu = s0, rs = s0, ga = s0, be = s0;
  if (MODE == 0 && cvalid) mu = *reinterpret_cast<const float4*>(a.x + ch);      // pivot = row 0: sums of (x - pivot) do not cancel in E[x^2] - E[x]^2
  if (MODE == 1 && cvalid) {
    mu = *reinterpret_cast<const float4*>(a.stat + ch);
    rs = *reinterpret_cast<const float4*>(a.stat + a.c + ch);
    ga = *reinterpret_cast<const float4*>(a.gamma + ch);
    be = *reinterpret_cast<const float4*>(a.beta + ch);
  }
  // four rows in flight per thread (one 16-byte load per row and operand): a single dependent load per iteration left the pass
  // latency-bound at ~1.5 TB/s (profiles/r05_a2_lfae_census.txt)
  auto accumulate = [&](const float4& v, float4 g) {
    if (MODE == 0) {
      const float dx_ = v.x - mu.x, dy_ = v.y - mu.y, dz_ = v.z - mu.z, dw_ = v.w - mu.w;
      s0.x += dx_; s0.y += dy_; s0.z += dz_; s0.w += dw_;
      s1.x = fmaf(dx_, dx_, s1.x); s1.y = fmaf(dy_, dy_, s1.y); s1.z = fmaf(dz_, dz_, s1.z); s1.w = fmaf(dw_, dw_, s1.w);
    } else {
      const float hx = (v.x - mu.x) * rs.x, hy = (v.y - mu.y) * rs.y, hz = (v.z - mu.z) * rs.z, hw = (v.w - mu.w) * rs.w;
      if (a.relu) {
        if (!(fmaf(hx, ga.x, be.x) > 0.f)) g.x = 0.f;
        if (!(fmaf(hy, ga.y, be.y) > 0.f)) g.y = 0.f;
        if (!(fmaf(hz, ga.z, be.z) > 0.f)) g.z = 0.f;
        if (!(fmaf(hw, ga.w, be.w) > 0.f)) g.w = 0.f;
      }
      s0.x += g.x; s0.y += g.y; s0.z += g.z; s0.w += g.w;
      s1.x = fmaf(g.x, hx, s1.x); s1.y = fmaf(g.y, hy, s1.y); s1.z = fmaf(g.z, hz, s1.z); s1.w = fmaf(g.w, hw, s1.w);
    }
  };
  if (cvalid) {
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t r = r0 + rl;
    for (; r + 3 * lanes < r1; r += 4 * lanes) {
      float4 v[4], g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = *reinterpret_cast<const float4*>(a.x + (r + u * lanes) * a.ldx + ch);
        g[u] = MODE == 1 ? *reinterpret_cast<const float4*>(a.dy + (r + u * lanes) * a.lddy + ch) : zero4;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) accumulate(v[u], g[u]);
    }
    for (; r < r1; r += lanes) {
      const float4 v = *reinterpret_cast<const float4*>(a.x + r * a.ldx + ch);
      const float4 g = MODE == 1 ? *reinterpret_cast<const float4*>(a.dy + r * a.lddy + ch) : zero4;
      accumulate(v, g);
    }
  }
  float* m0 = &sm[0][tid * 4];
  float* m1 = &sm[1][tid * 4];
  m0[0] = s0.x; m0[1] = s0.y; m0[2] = s0.z; m0[3] = s0.w;
  m1[0] = s1.x; m1[1] = s1.y; m1[2] = s1.z; m1[3] = s1.w;
  __syncthreads();
  // The chunk's (sum, sum) pair of a channel travels as ONE 8-byte agent-scope word: written through to memory by its store, read past the
  // (per-XCD, mutually non-coherent) L2s by the finalizer's load - no cache-wide release / acquire around the ticket (lfdm_device.h,
  // guide section 6 guideline 16).  The release fence this replaces wrote back every dirty line of the XCD's L2 - the producing
  // convolution's output - from inside each of the 1 024+ workgroup tails: the pass ran at 0.6-1.5 TB/s (profiles/r05_p_bench_bn.txt).
  unsigned long long* p1 = reinterpret_cast<unsigned long long*>(a.part1) + ((int64_t)ct * a.nchunk + chunk) * tw;
  if (tid < tw) {
    const int qd = tid >> 2, comp = tid & 3;
    float acc0 = 0.f, acc1 = 0.f;
    for (int l = 0; l < lanes; ++l) {                     // fixed order
      acc0 += sm[0][(l * q + qd) * 4 + comp];
      acc1 += sm[1][(l * q + qd) * 4 + comp];
    }
    lfdm_agent_store_u64(p1 + tid, (unsigned long long)__float_as_uint(acc0) | ((unsigned long long)__float_as_uint(acc1) << 32));
  }
  // level-1 fold by the workgroup that completes its group of BN_GROUP chunks
  LFDM_DRAIN_STORES();
  __syncthreads();
  const int grp = chunk / BN_GROUP;
  const int c0 = grp * BN_GROUP, c1 = c0 + BN_GROUP < a.nchunk ? c0 + BN_GROUP : a.nchunk;
  if (tid == 0) {
    unsigned* cnt = a.tickets + ct * a.ngroups + grp;
    const bool last = lfdm_ticket_take(cnt) == (unsigned)(c1 - c0 - 1);
    if (last) lfdm_ticket_reset(cnt);
    s_last = last ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  if (tid < tw) {
    double acc0 = 0.0, acc1 = 0.0;
    const unsigned long long* base = reinterpret_cast<const unsigned long long*>(a.part1) + (int64_t)ct * a.nchunk * tw + tid;
    for (int k = c0; k < c1; k += 8) {                     // eight loads in flight, folded in index order
      unsigned long long w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = k + u < c1 ? lfdm_agent_load_u64(base + (int64_t)(k + u) * tw) : 0ull;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc0 += (double)__uint_as_float((unsigned)(w[u] & 0xffffffffull));
        acc1 += (double)__uint_as_float((unsigned)(w[u] >> 32));
      }
    }
    a.part2[(((int64_t)ct * a.ngroups + grp) * 2 + 0) * tw + tid] = acc0;
    a.part2[(((int64_t)ct * a.ngroups + grp) * 2 + 1) * tw + tid] = acc1;
  }
}

// grid (nchunk, ctiles).  MODE 0: y = relu(xhat * gamma + beta), statistics + running statistics written by chunk 0.
// MODE 1: dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)), dgamma / dbeta written by chunk 0.
template <int MODE>
__global__ __launch_bounds__(256) void bn_apply_kernel(BnArgs a) {
  __shared__ float s_mu[64], s_rs[64], s_ga[64], s_be[64], s_k1[64], s_k2[64];
  const int tid = threadIdx.x, q = a.q, lanes = 256 / q, tw = 4 * q;
  const int chunk = blockIdx.x, ct = blockIdx.y, seg = blockIdx.z;
  const float* x_all = a.x;
  const double* part2_all = a.part2;
  a.x += seg * a.rows * a.ldx;
  if (MODE == 1) a.dy += seg * a.rows * a.lddy;
  if (MODE == 1 && a.add) a.add += seg * a.rows * a.ldadd;
  a.out += seg * a.rows * a.ldo;
  a.stat += (int64_t)seg * 2 * a.c;
  a.part2 += seg * a.seg_p2;
  if (tid < tw) {
    const int channel = ct * tw + tid;
    if (channel < a.c) {
      // the (sum, sum) pair of one segment, level-2 fold in fixed order
      auto fold = [&](const double* p2, double& S0, double& S1) {
        S0 = 0.0; S1 = 0.0;
        for (int g = 0; g < a.ngroups; ++g) {
          S0 += p2[(((int64_t)ct * a.ngroups + g) * 2 + 0) * tw + tid];
          S1 += p2[(((int64_t)ct * a.ngroups + g) * 2 + 1) * tw + tid];
        }
      };
      double S0, S1;
      fold(a.part2, S0, S1);
      const double m = (double)a.rows;
      if (MODE == 0) {
        const double shifted = S0 / m;                 // mean of (x - pivot), pivot = row 0 (bn_reduce_kernel)
        const double mean = (double)a.x[channel] + shifted;
        double var = S1 / m - shifted * shifted;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
        s_mu[tid] = (float)mean;
        s_rs[tid] = rstd;
        if (chunk == 0) {
          a.stat[channel] = (float)mean;
          a.stat[a.c + channel] = rstd;
        }
        if (chunk == 0 && seg == 0 && a.running_mean) {
          // one momentum update per segment, in segment order (= the order of the separate module calls)
          float rm = a.running_mean[channel], rv = a.running_var[channel];
          for (int sg = 0; sg < a.segments; ++sg) {
            double T0, T1;
            fold(part2_all + sg * a.seg_p2, T0, T1);
            const double sh = T0 / m;
            const double mean_s = (double)x_all[sg * a.rows * a.ldx + channel] + sh;
            double var_s = T1 / m - sh * sh;
            if (var_s < 0.0) var_s = 0.0;
            const double unbiased = a.rows > 1 ? var_s * m / (m - 1.0) : var_s;
            rm = (float)((1.0 - a.momentum) * (double)rm + a.momentum * mean_s);
            rv = (float)((1.0 - a.momentum) * (double)rv + a.momentum * unbiased);
          }
          a.running_mean[channel] = rm;
          a.running_var[channel] = rv;
        }
      } else {
        s_mu[tid] = a.stat[channel];
        s_rs[tid] = a.stat[a.c + channel];
        s_k1[tid] = (float)(S0 / m);
        s_k2[tid] = (float)(S1 / m);
        if (chunk == 0 && seg == 0) {
          double D0 = S0, D1 = S1;
          for (int sg = 1; sg < a.segments; ++sg) {
            double T0, T1;
            fold(part2_all + sg * a.seg_p2, T0, T1);
            D0 += T0; D1 += T1;
          }
          if (a.dbeta) a.dbeta[channel] = (float)D0;
          if (a.dgamma) a.dgamma[channel] = (float)D1;
        }
      }
      s_ga[tid] = a.gamma[channel];
      s_be[tid] = a.beta[channel];
    }
  }
  __syncthreads();
  const int quad = tid & (q - 1), rl = tid / q;
  const int ch = ct * tw + quad * 4;
  if (ch >= a.c) return;
  const int64_t r0 = (int64_t)chunk * a.chunk_rows;
  const int64_t r1 = r0 + a.chunk_rows < a.rows ? r0 + a.chunk_rows : a.rows;
  float mu[4], rs[4], ga[4], be[4], k1[4], k2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    mu[i] = s_mu[quad * 4 + i]; rs[i] = s_rs[quad * 4 + i]; ga[i] = s_ga[quad * 4 + i]; be[i] = s_be[quad * 4 + i];
    k1[i] = MODE == 1 ? s_k1[quad * 4 + i] : 0.f;
    k2[i] = MODE == 1 ? s_k2[quad * 4 + i] : 0.f;
  }
  auto apply_row = [&](int64_t r, const float4& v4, const float4& g4) {
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
    float o[4];
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float z = fmaf((v[i] - mu[i]) * rs[i], ga[i], be[i]);
        o[i] = (a.relu && !(z > 0.f)) ? 0.f : z;
      }
    } else {
      const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float h = (v[i] - mu[i]) * rs[i];
        float gi = g[i];
        if (a.relu && !(fmaf(h, ga[i], be[i]) > 0.f)) gi = 0.f;
        o[i] = ga[i] * rs[i] * (gi - k1[i] - h * k2[i]);
      }
      if (a.add) {
        const float4 e = *reinterpret_cast<const float4*>(a.add + r * a.ldadd + ch);
        o[0] += e.x; o[1] += e.y; o[2] += e.z; o[3] += e.w;
      }
    }
    *reinterpret_cast<float4*>(a.out + r * a.ldo + ch) = make_float4(o[0], o[1], o[2], o[3]);
  };
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t r = r0 + rl;
  for (; r + 3 * lanes < r1; r += 4 * lanes) {
    float4 v[4], g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = *reinterpret_cast<const float4*>(a.x + (r + u * lanes) * a.ldx + ch);
      g[u] = MODE == 1 ? *reinterpret_cast<const float4*>(a.dy + (r + u * lanes) * a.lddy + ch) : zero4;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) apply_row(r + u * lanes, v[u], g[u]);
  }
  for (; r < r1; r += lanes) {
    const float4 v4 = *reinterpret_cast<const float4*>(a.x + r * a.ldx + ch);
    const float4 g4 = MODE == 1 ? *reinterpret_cast<const float4*>(a.dy + r * a.lddy + ch) : zero4;
    apply_row(r, v4, g4);
  }
}

// =====================================================================================================================================
// depth-wise blur + subsample, any strides
// =====================================================================================================================================
constexpr int BLUR_MAX_TAPS = 32 * 32;

struct BlurArgs {
  const float* x;           // forward input / backward: unused
  const float* wgt;         // (C, k, k)
  float* out;               // forward output
  const float* dy;          // backward: gradient of the forward output (out strides)
  float* dx;                // backward: gradient of the input (x strides)
  int64_t xs_n, xs_c, xs_h, xs_w, os_n, os_c, os_h, os_w;
  int channels, c_store, h, w, k, pad, stride, ho, wo;
  const float* scale;       // per channel, or null (1)
  const float* bias;        // per channel, or null (0)
};

// grid (ceil(ho*wo/256), c_store, N): out = scale[c] * (blur(x))[every stride-th] + bias[c]; channels >= `channels` store zeros
__global__ __launch_bounds__(256) void blur_down_kernel(BlurArgs a) {
  __shared__ float s_w[BLUR_MAX_TAPS];
  const int c = blockIdx.y, n = blockIdx.z, k = a.k;
  const bool real = c < a.channels;
  if (real)
    for (int i = threadIdx.x; i < k * k; i += 256) s_w[i] = a.wgt[(int64_t)c * k * k + i];
  __syncthreads();
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= a.ho * a.wo) return;
  const int oy = o / a.wo, ox = o - oy * a.wo;
  float acc = 0.f;
  if (real) {
    const float* plane = a.x + n * a.xs_n + c * a.xs_c;
    for (int ky = 0; ky < k; ++ky) {
      const int iy = oy * a.stride + ky - a.pad;
      if (iy < 0 || iy >= a.h) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int ix = ox * a.stride + kx - a.pad;
        if (ix >= 0 && ix < a.w) acc = fmaf(plane[iy * a.xs_h + ix * a.xs_w], s_w[ky * k + kx], acc);
      }
    }
    if (a.scale) acc *= a.scale[c];
    if (a.bias) acc += a.bias[c];
  }
  a.out[n * a.os_n + c * a.os_c + oy * a.os_h + ox * a.os_w] = acc;
}

// grid (ceil(h*w/256), channels, N): dx[y, x] = scale[c] * sum over the outputs whose window covers (y, x) of w[ky][kx] * dy[oy, ox]
__global__ __launch_bounds__(256) void blur_down_bwd_kernel(BlurArgs a) {
  __shared__ float s_w[BLUR_MAX_TAPS];
  const int c = blockIdx.y, n = blockIdx.z, k = a.k, s = a.stride;
  for (int i = threadIdx.x; i < k * k; i += 256) s_w[i] = a.wgt[(int64_t)c * k * k + i];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.h * a.w) return;
  const int y = i / a.w, x = i - y * a.w;
  // ky = y + pad - oy*s in [0, k)  <=>  oy in [ceil((y + pad - k + 1) / s), floor((y + pad) / s)]
  const int ty = y + a.pad, tx = x + a.pad;
  int oy0 = ty - k + 1 <= 0 ? 0 : (ty - k + 1 + s - 1) / s, oy1 = ty / s;
  int ox0 = tx - k + 1 <= 0 ? 0 : (tx - k + 1 + s - 1) / s, ox1 = tx / s;
  if (oy1 > a.ho - 1) oy1 = a.ho - 1;
  if (ox1 > a.wo - 1) ox1 = a.wo - 1;
  const float* plane = a.dy + n * a.os_n + c * a.os_c;
  float acc = 0.f;
  for (int oy = oy0; oy <= oy1; ++oy)
    for (int ox = ox0; ox <= ox1; ++ox)
      acc = fmaf(plane[oy * a.os_h + ox * a.os_w], s_w[(ty - oy * s) * k + (tx - ox * s)], acc);
  if (a.scale) acc *= a.scale[c];
  a.dx[n * a.xs_n + c * a.xs_c + y * a.xs_h + x * a.xs_w] = acc;
}

// =====================================================================================================================================
// grid_sample / warp backward
// =====================================================================================================================================
// ATen upsample_bilinear2d (align_corners=False) source index - the same arithmetic as warp.hip's forward
__device__ __forceinline__ void resize_src(int dst, int n_in, int n_out, int& i0, int& i1, float& l1) {
  const float scale = (float)n_in / (float)n_out;
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

struct WTaps {
  int x0, y0;
  float wx0, wx1, wy0, wy1;
  float occ;
};

__device__ __forceinline__ float bilerp4(const float* m, int fw, int y0, int y1, int x0, int x1, float ly, float lx) {
  const float v00 = m[y0 * fw + x0], v01 = m[y0 * fw + x1];
  const float v10 = m[y1 * fw + x0], v11 = m[y1 * fw + x1];
  return (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

// the forward's per-pixel set-up (warp.hip: pixel_taps), frames == 1 per sample
__device__ __forceinline__ WTaps warp_taps(const lfdm_warp_bwd_params& p, int n, int oy, int ox) {
  const int64_t moff = (int64_t)n * p.fsn;
  float gx, gy, o = 1.f;
  if (p.fh == p.h && p.fw == p.w) {
    gx = p.flow_x[moff + oy * p.fw + ox];
    gy = p.flow_y[moff + oy * p.fw + ox];
    if (p.occ) o = p.occ[moff + oy * p.fw + ox];
  } else {
    int y0, y1, x0, x1;
    float ly, lx;
    resize_src(oy, p.fh, p.h, y0, y1, ly);
    resize_src(ox, p.fw, p.w, x0, x1, lx);
    gx = bilerp4(p.flow_x + moff, p.fw, y0, y1, x0, x1, ly, lx);
    gy = bilerp4(p.flow_y + moff, p.fw, y0, y1, x0, x1, ly, lx);
    if (p.occ) o = bilerp4(p.occ + moff, p.fw, y0, y1, x0, x1, ly, lx);
  }
  float ix = ((gx + 1.f) * (float)p.w - 1.f) * 0.5f;
  float iy = ((gy + 1.f) * (float)p.h - 1.f) * 0.5f;
  ix = fminf(fmaxf(ix, -1.0e6f), 1.0e6f);
  iy = fminf(fmaxf(iy, -1.0e6f), 1.0e6f);
  const float fx = floorf(ix), fy = floorf(iy);
  WTaps r;
  r.x0 = (int)fx;
  r.y0 = (int)fy;
  r.wx1 = ix - fx;
  r.wx0 = (fx + 1.f) - ix;
  r.wy1 = iy - fy;
  r.wy0 = (fy + 1.f) - iy;
  r.occ = o;
  return r;
}

// exponent k of the fixed-point scale 2^k for a scatter whose terms are bounded by `amax` and whose per-address count is bounded by
// `count`: count * amax * 2^k < 2^62
__device__ __forceinline__ int fix_exponent(unsigned amax_bits, int64_t count) {
  const float amax = __uint_as_float(amax_bits);
  if (!(amax > 0.f)) return 0;
  int e;
  frexpf(amax, &e);                     // amax = m * 2^e, m in [0.5, 1)  ->  amax < 2^e
  int lc = 0;
  while (((int64_t)1 << lc) < count) ++lc;
  int k = 62 - lc - e;
  if (k > 120) k = 120;
  if (k < -120) k = -120;
  return k;
}

__device__ __forceinline__ void fix_add(long long* acc, float v, int k) {
  if (v == 0.f) return;
  const long long q = (long long)rintf(ldexpf(v, k));
#if defined(LFDM_EMU_BUILD)
  __atomic_fetch_add(acc, q, __ATOMIC_SEQ_CST);
#else
  __hip_atomic_fetch_add(acc, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// max |x| over a strided (rows, c) matrix as the bit pattern of a non-negative float (integer max: order independent)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t rows, int c, int64_t ld, unsigned* out) {
  __shared__ float s_m[4];
  float m = 0.f;
  // NaN never wins a comparison: a NaN gradient stays a NaN through the float paths of the caller
  if ((c & 3) == 0 && (ld & 3) == 0 && ((uintptr_t)x & 15) == 0) {       // 16-byte loads, two in flight (a scalar loop read 2.5 TB/s)
    const int c4 = c >> 2;
    const int64_t total = rows * c4, step = (int64_t)gridDim.x * 256;
    auto at = [&](int64_t i) {
      const int64_t r = i / c4;
      return *reinterpret_cast<const float4*>(x + r * ld + 4 * (i - r * c4));
    };
    auto fold = [&](const float4& v) {
      const float a = fmaxf(fabsf(v.x), fabsf(v.y)), b = fmaxf(fabsf(v.z), fabsf(v.w));
      if (a > m) m = a;
      if (b > m) m = b;
    };
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + step < total; i += 2 * step) {
      const float4 v0 = at(i), v1 = at(i + step);
      fold(v0);
      fold(v1);
    }
    if (i < total) fold(at(i));
  } else {
    const int64_t total = rows * c;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const int64_t r = i / c;
      const float v = fabsf(x[r * ld + (i - r * c)]);
      if (v > m) m = v;
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    atomicMax(out, __float_as_uint(m));
  }
}

// fixed-point accumulator -> float (+= nothing: plain store), accumulator re-zeroed for the next call.  The LAST workgroup also clears
// the max word (ticket), so the workspace is ready for the next backward without a memset launch.
__global__ __launch_bounds__(256) void fix_finalize_kernel(long long* __restrict__ acc, float* __restrict__ out, int64_t rows, int c, int64_t ld,
                                                           unsigned* amax_bits, int64_t count, unsigned* ticket) {
  const int k = fix_exponent(*amax_bits, count);
  const int64_t total = rows * c;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const long long q = acc[i];
    acc[i] = 0;
    const int64_t r = i / c;
    out[r * ld + (i - r * c)] = (float)ldexp((double)q, -k);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (lfdm_ticket_take(ticket) == gridDim.x - 1) {
      lfdm_ticket_reset(ticket);
      *amax_bits = 0u;
    }
  }
}

// Channels-last backward of out = warp(src; flow) * occ + prev * (1 - occ).  A pixel is served by G = C/4 adjacent lanes (a power of two
// <= 64); lane l owns the channels {l, l + G, l + 2G, l + 3G}, so that every load, store and - what matters - every 64-bit atomic
// instruction of the pixel's lanes covers ONE contiguous run of G elements (with four consecutive channels per lane an atomic instruction
// touched 8 of every 32 bytes over a 4x larger span: four times the cache-line operations at the L2's atomic units).  Per pixel:
// d_prev = dout * (1 - occ); d_src[tap] += w_tap * occ * dout (fixed point); the three map gradients (d flow_x, d flow_y, d occ) at OUTPUT
// resolution into dmaps (N, 3, H, W) - lfdm_resize_adjoint folds them to the map resolution.
// grid (blocks over h*w*G lane items, N)
__global__ __launch_bounds__(256) void warp_bwd_cl_kernel(lfdm_warp_bwd_params p) {
  const int g = p.c >> 2;
  const int gshift = __builtin_ctz(g);
  const int hw = p.h * p.w;
  const int per_img = hw * g;
  const int n = blockIdx.y;
  const int kfix = p.dsrc_fix ? fix_exponent(*p.amax_bits, (int64_t)4 * hw) : 0;
  for (int idx0 = blockIdx.x * 256; idx0 < per_img; idx0 += gridDim.x * 256) {
    const int idx = idx0 + threadIdx.x;
    const bool live = idx < per_img;
    const int pix = live ? (idx >> gshift) : 0;
    const int c = idx - ((idx >> gshift) << gshift);                 // first channel of this lane; the others follow at stride g
    const int oy = pix / p.w, ox = pix - oy * p.w;
    const int64_t gp = (int64_t)n * hw + pix;
    float dfx = 0.f, dfy = 0.f, dob = 0.f;
    if (live) {
      const WTaps tp = warp_taps(p, n, oy, ox);
      const float o = tp.occ, om = 1.f - o;
      float d[4], pv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) d[i] = p.dout[gp * p.ld_dout + c + i * g];
      if (p.prev) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pv[i] = p.prev[gp * p.ld_prev + c + i * g];
      }
      if (p.dprev) {
#pragma unroll
        for (int i = 0; i < 4; ++i) p.dprev[gp * p.ld_dprev + c + i * g] = d[i] * om;
      }
      float v[4][4];
      bool in[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int yy = tp.y0 + (k >> 1), xx = tp.x0 + (k & 1);
        in[k] = yy >= 0 && yy < p.h && xx >= 0 && xx < p.w;
        const float* sp = p.src + ((int64_t)n * hw + (in[k] ? yy * p.w + xx : 0)) * p.ld_src + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[k][i] = in[k] ? sp[i * g] : 0.f;
      }
      const float wk[4] = {tp.wx0 * tp.wy0, tp.wx1 * tp.wy0, tp.wx0 * tp.wy1, tp.wx1 * tp.wy1};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dw = d[i] * o;                                   // gradient of the warped value
        // warped value in the forward's order: taps (y0,x0) (y0,x1) (y1,x0) (y1,x1)
        float wv = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) wv = fmaf(v[k][i], in[k] ? wk[k] : 0.f, wv);
        dob += d[i] * (wv - pv[i]);
        // d warped / d ix, d iy (grid_sampler_2d_backward: out-of-range taps contribute zero)
        const float gx_ = (v[1][i] - v[0][i]) * tp.wy0 + (v[3][i] - v[2][i]) * tp.wy1;
        const float gy_ = (v[2][i] - v[0][i]) * tp.wx0 + (v[3][i] - v[1][i]) * tp.wx1;
        dfx = fmaf(dw, gx_, dfx);
        dfy = fmaf(dw, gy_, dfy);
      }
      if (p.dsrc_fix) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (!in[k]) continue;
          const int yy = tp.y0 + (k >> 1), xx = tp.x0 + (k & 1);
          long long* dst = p.dsrc_fix + ((int64_t)n * hw + yy * p.w + xx) * p.c + c;
          const float wo = wk[k] * o;
#pragma unroll
          for (int i = 0; i < 4; ++i) fix_add(dst + i * g, d[i] * wo, kfix);
        }
      }
    }
    // sum over the pixel's G lanes (adjacent lanes of one wavefront; G <= 64 is a power of two)
    for (int m = 1; m < g; m <<= 1) {
      dfx += __shfl_xor(dfx, m);
      dfy += __shfl_xor(dfy, m);
      dob += __shfl_xor(dob, m);
    }
    if (live && c == 0 && p.dmaps) {
      float* dm = p.dmaps + (int64_t)n * 3 * hw + pix;
      dm[0] = dfx * (0.5f * (float)p.w);
      dm[hw] = dfy * (0.5f * (float)p.h);
      dm[2 * hw] = dob;
    }
  }
}

// The same backward for tensors with few channels and any strides (the RGB image of generator.py:126-128; the deformed sources of
// pixelwise_flow_predictor.py:95-102 with src shared by `n_div` consecutive samples): one thread per output pixel loops over the channels.
// d_src is not produced (the sampled tensor is an input image).  grid (ceil(h*w/256), N)
__global__ __launch_bounds__(256) void warp_bwd_px_kernel(lfdm_warp_bwd_params p) {
  const int hw = p.h * p.w;
  const int n = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= hw) return;
  const int oy = pix / p.w, ox = pix - oy * p.w;
  const WTaps tp = warp_taps(p, n, oy, ox);
  const float o = tp.occ, om = 1.f - o;
  const float wk[4] = {tp.wx0 * tp.wy0, tp.wx1 * tp.wy0, tp.wx0 * tp.wy1, tp.wx1 * tp.wy1};
  bool in[4];
  int64_t off[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int yy = tp.y0 + (k >> 1), xx = tp.x0 + (k & 1);
    in[k] = yy >= 0 && yy < p.h && xx >= 0 && xx < p.w;
    off[k] = in[k] ? yy * p.ss_h + xx * p.ss_w : 0;
  }
  const float* sb = p.src + (int64_t)(n / p.n_div) * p.ss_n;
  float dfx = 0.f, dfy = 0.f, dob = 0.f;
  for (int c = 0; c < p.c; ++c) {
    const float d = p.dout[n * p.ds_n + c * p.ds_c + oy * p.ds_h + ox * p.ds_w];
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = in[k] ? sb[c * p.ss_c + off[k]] : 0.f;
    float wv = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) wv = fmaf(v[k], in[k] ? wk[k] : 0.f, wv);
    float pv = 0.f;
    if (p.prev) pv = p.prev[n * p.ps_n + c * p.ps_c + oy * p.ps_h + ox * p.ps_w];
    if (p.dprev) p.dprev[n * p.dps_n + c * p.dps_c + oy * p.dps_h + ox * p.dps_w] = d * om;
    dob += d * (wv - pv);
    const float dw = d * o;
    dfx = fmaf(dw, (v[1] - v[0]) * tp.wy0 + (v[3] - v[2]) * tp.wy1, dfx);
    dfy = fmaf(dw, (v[2] - v[0]) * tp.wx0 + (v[3] - v[1]) * tp.wx1, dfy);
  }
  float* dm = p.dmaps + (int64_t)n * 3 * hw + pix;
  dm[0] = dfx * (0.5f * (float)p.w);
  dm[hw] = dfy * (0.5f * (float)p.h);
  dm[2 * hw] = dob;
}

// Adjoint of the bilinear resize (fh, fw) -> (h, w) that the warp applies to its three maps: dlow[m, y, x] = sum over the output pixels
// whose two source rows / columns include (y, x) of their weight * dhigh.  Gather form (each low-resolution pixel scans the few
// output rows / columns that can read it: run-to-run identical, no atomics).  grid (ceil(fh*fw/256), planes = N*3)
__global__ __launch_bounds__(256) void resize_adjoint_kernel(const float* __restrict__ dhigh, float* __restrict__ dlow, int h, int w,
                                                             int fh, int fw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= fh * fw) return;
  const int y = i / fw, x = i - y * fw;
  const float* src = dhigh + (int64_t)blockIdx.y * h * w;
  // output rows that can touch low row y: src = s*(oy+0.5)-0.5 in (y-1, y+1)  ->  oy in ((y-0.5)/s - 0.5, (y+1.5)/s - 0.5); widen by one
  const float sy = (float)fh / (float)h, sx = (float)fw / (float)w;
  int oy_lo = (int)floorf(((float)y - 0.5f) / sy - 0.5f) - 1, oy_hi = (int)ceilf(((float)y + 1.5f) / sy - 0.5f) + 1;
  int ox_lo = (int)floorf(((float)x - 0.5f) / sx - 0.5f) - 1, ox_hi = (int)ceilf(((float)x + 1.5f) / sx - 0.5f) + 1;
  if (oy_lo < 0) oy_lo = 0;
  if (ox_lo < 0) ox_lo = 0;
  if (oy_hi > h - 1) oy_hi = h - 1;
  if (ox_hi > w - 1) ox_hi = w - 1;
  float acc = 0.f;
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    int y0, y1;
    float ly;
    resize_src(oy, fh, h, y0, y1, ly);
    const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
    if (wy == 0.f) continue;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      int x0, x1;
      float lx;
      resize_src(ox, fw, w, x0, x1, lx);
      const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
      if (wx != 0.f) acc = fmaf(src[oy * w + ox], wy * wx, acc);
    }
  }
  dlow[(int64_t)blockIdx.y * fh * fw + i] = acc;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// F.grid_sample (bilinear, align_corners=False) with an explicit grid (N, ho, wo, 2) on strided small-channel tensors; padding zeros (0)
// or reflection (1).  grid (ceil(ho*wo/256), N)
__device__ __forceinline__ float reflect_coord(float v, int size) {
  // ATen reflect_coordinates(in, -1, 2*size - 1) for align_corners=False, then clip_coordinates
  const float twice_low = -1.f, twice_high = 2.f * (float)size - 1.f;
  const float mn = twice_low * 0.5f, span = (twice_high - twice_low) * 0.5f;
  float in = fabsf(v - mn);
  const float extra = fmodf(in, span);
  const int flips = (int)floorf(in / span);
  float r = (flips % 2 == 0) ? extra + mn : span - extra + mn;
  return fminf((float)(size - 1), fmaxf(r, 0.f));
}

struct GsArgs {
  const float* x;
  const float* grid;
  float* out;
  const float* dout;
  float* dgrid;
  int64_t xs_n, xs_c, xs_h, xs_w, os_n, os_c, os_h, os_w;
  int c, h, w, ho, wo, n_div, pad_mode;
};

template <bool BWD>
__global__ __launch_bounds__(256) void grid_sample_kernel(GsArgs a) {
  const int n = blockIdx.y;
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= a.ho * a.wo) return;
  const int oy = o / a.wo, ox = o - oy * a.wo;
  const float* gp = a.grid + ((int64_t)n * a.ho * a.wo + o) * 2;
  float ix = ((gp[0] + 1.f) * (float)a.w - 1.f) * 0.5f;
  float iy = ((gp[1] + 1.f) * (float)a.h - 1.f) * 0.5f;
  if (a.pad_mode == 1) {
    ix = reflect_coord(ix, a.w);
    iy = reflect_coord(iy, a.h);
  }
  ix = fminf(fmaxf(ix, -1.0e6f), 1.0e6f);
  iy = fminf(fmaxf(iy, -1.0e6f), 1.0e6f);
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix, wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
  const float wk[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
  bool in[4];
  int64_t off[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
    in[k] = yy >= 0 && yy < a.h && xx >= 0 && xx < a.w;
    off[k] = in[k] ? yy * a.xs_h + xx * a.xs_w : 0;
  }
  const float* sb = a.x + (int64_t)(n / a.n_div) * a.xs_n;
  float dfx = 0.f, dfy = 0.f;
  for (int c = 0; c < a.c; ++c) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = in[k] ? sb[c * a.xs_c + off[k]] : 0.f;
    if (!BWD) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) acc = fmaf(v[k], in[k] ? wk[k] : 0.f, acc);
      a.out[n * a.os_n + c * a.os_c + oy * a.os_h + ox * a.os_w] = acc;
    } else {
      const float d = a.dout[n * a.os_n + c * a.os_c + oy * a.os_h + ox * a.os_w];
      dfx = fmaf(d, (v[1] - v[0]) * wy0 + (v[3] - v[2]) * wy1, dfx);
      dfy = fmaf(d, (v[2] - v[0]) * wx0 + (v[3] - v[1]) * wx1, dfy);
    }
  }
  if (BWD) {
    float* dg = a.dgrid + ((int64_t)n * a.ho * a.wo + o) * 2;
    dg[0] = dfx * (0.5f * (float)a.w);
    dg[1] = dfy * (0.5f * (float)a.h);
  }
}

// =====================================================================================================================================
// 2x2 symmetric SVD backward:  gA = U [[gs0, k s1 / (s1^2 - s0^2)], [k s0 / (s1^2 - s0^2), gs1]] U^T,  k = (U^T gU)_01 - (U^T gU)_10
// (torch's svd_backward with V = U - a positive semi-definite input - and no gradient through V)
// =====================================================================================================================================
__global__ __launch_bounds__(256) void svd2x2_sym_bwd_kernel(const float* __restrict__ u, const float* __restrict__ s, const float* __restrict__ gu,
                                                            const float* __restrict__ gs, float* __restrict__ ga, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float u00 = u[4 * i], u01 = u[4 * i + 1], u10 = u[4 * i + 2], u11 = u[4 * i + 3];
  const float s0 = s[2 * i], s1 = s[2 * i + 1];
  float m00 = 0.f, m11 = 0.f, m01 = 0.f, m10 = 0.f;
  if (gs) { m00 = gs[2 * i]; m11 = gs[2 * i + 1]; }
  if (gu) {
    const float g00 = gu[4 * i], g01 = gu[4 * i + 1], g10 = gu[4 * i + 2], g11 = gu[4 * i + 3];
    const float k = (u00 * g01 + u10 * g11) - (u01 * g00 + u11 * g10);          // (U^T gU)_01 - (U^T gU)_10
    const float den = s1 * s1 - s0 * s0;
    m01 = k / den * s1;
    m10 = k / den * s0;
  }
  // T = U * M, gA = T * U^T
  const float t00 = u00 * m00 + u01 * m10, t01 = u00 * m01 + u01 * m11;
  const float t10 = u10 * m00 + u11 * m10, t11 = u10 * m01 + u11 * m11;
  ga[4 * i] = t00 * u00 + t01 * u01;
  ga[4 * i + 1] = t00 * u10 + t01 * u11;
  ga[4 * i + 2] = t10 * u00 + t11 * u01;
  ga[4 * i + 3] = t10 * u10 + t11 * u11;
}

// =====================================================================================================================================
// 2x2 pooling family on channels-last rows, element-wise ReLU mask, mean absolute difference
// =====================================================================================================================================
// (h, w) is always the FINE resolution (even).  mode 0: average 2x2 (DownBlock2d, util.py:133) | 1: sum 2x2 (backward of the nearest x2
// up-sampling) | 2: nearest x2 (UpBlock2d, util.py:109) | 3: nearest x2 times 0.25 (backward of the average) | 4: max 2x2 (VGG-19's
// MaxPool2d) | 5: backward of the max: aux = dy at the coarse resolution, x = the forward's input, the gradient goes to the FIRST maximum of
// the window in row-major order (ATen's max_pool2d_with_indices keeps the first).
template <int MODE>
__global__ __launch_bounds__(256) void pool2_kernel(const float* __restrict__ x, const float* __restrict__ aux, float* __restrict__ out,
                                                    int n_img, int h, int w, int channels) {
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
    const int64_t fine = (((int64_t)n * h + 2 * oy) * w + 2 * ox) * channels + c;
    const int64_t coarse = (((int64_t)n * ho + oy) * wo + ox) * channels + c;
    const int64_t dxs = channels, dys = (int64_t)w * channels;
    if (MODE == 2 || MODE == 3) {
      float4 v = *reinterpret_cast<const float4*>(x + coarse);
      if (MODE == 3) { v.x *= 0.25f; v.y *= 0.25f; v.z *= 0.25f; v.w *= 0.25f; }
      *reinterpret_cast<float4*>(out + fine) = v;
      *reinterpret_cast<float4*>(out + fine + dxs) = v;
      *reinterpret_cast<float4*>(out + fine + dys) = v;
      *reinterpret_cast<float4*>(out + fine + dys + dxs) = v;
      continue;
    }
    const float4 v00 = *reinterpret_cast<const float4*>(x + fine);
    const float4 v01 = *reinterpret_cast<const float4*>(x + fine + dxs);
    const float4 v10 = *reinterpret_cast<const float4*>(x + fine + dys);
    const float4 v11 = *reinterpret_cast<const float4*>(x + fine + dys + dxs);
    if (MODE == 0 || MODE == 1) {
      const float k = MODE == 0 ? 0.25f : 1.f;
      float4 y;
      y.x = ((v00.x + v01.x) + (v10.x + v11.x)) * k;
      y.y = ((v00.y + v01.y) + (v10.y + v11.y)) * k;
      y.z = ((v00.z + v01.z) + (v10.z + v11.z)) * k;
      y.w = ((v00.w + v01.w) + (v10.w + v11.w)) * k;
      *reinterpret_cast<float4*>(out + coarse) = y;
    } else if (MODE == 4) {
      float4 y;
      y.x = fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x));
      y.y = fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y));
      y.z = fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z));
      y.w = fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w));
      *reinterpret_cast<float4*>(out + coarse) = y;
    } else {
      const float4 g = *reinterpret_cast<const float4*>(aux + coarse);
      const float a[4][4] = {{v00.x, v01.x, v10.x, v11.x}, {v00.y, v01.y, v10.y, v11.y}, {v00.z, v01.z, v10.z, v11.z}, {v00.w, v01.w, v10.w, v11.w}};
      const float gg[4] = {g.x, g.y, g.z, g.w};
      float o[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int best = 0;
#pragma unroll
        for (int t = 1; t < 4; ++t)
          if (a[j][t] > a[j][best]) best = t;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t][j] = t == best ? gg[j] : 0.f;
      }
      *reinterpret_cast<float4*>(out + fine) = make_float4(o[0][0], o[0][1], o[0][2], o[0][3]);
      *reinterpret_cast<float4*>(out + fine + dxs) = make_float4(o[1][0], o[1][1], o[1][2], o[1][3]);
      *reinterpret_cast<float4*>(out + fine + dys) = make_float4(o[2][0], o[2][1], o[2][2], o[2][3]);
      *reinterpret_cast<float4*>(out + fine + dys + dxs) = make_float4(o[3][0], o[3][1], o[3][2], o[3][3]);
    }
  }
}

// out = y > 0 ? dy : 0 (the backward of a ReLU fused into the producing convolution's epilogue); n4 float4 items
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float4* __restrict__ y, const float4* __restrict__ dy, float4* __restrict__ out, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = y[i];
    float4 g = dy[i];
    if (!(v.x > 0.f)) g.x = 0.f;
    if (!(v.y > 0.f)) g.y = 0.f;
    if (!(v.z > 0.f)) g.z = 0.f;
    if (!(v.w > 0.f)) g.w = 0.f;
    out[i] = g;
  }
}

constexpr int L1_BLOCKS = 1024;
// forward: *out = weight / n * sum |x - y| (per-workgroup partials, folded by the last workgroup in index order: run-to-run identical);
// backward: dx = sign(x - y) * (*gout) * weight / n
__global__ __launch_bounds__(256) void l1_mean_fwd_kernel(const float4* __restrict__ x, const float4* __restrict__ y, int64_t n4, float scale,
                                                          float* __restrict__ partial, unsigned* ticket, float* __restrict__ out) {
  __shared__ float s_p[4];
  __shared__ int s_last;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 a = x[i], b = y[i];
    acc += (fabsf(a.x - b.x) + fabsf(a.y - b.y)) + (fabsf(a.z - b.z) + fabsf(a.w - b.w));
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = acc;
  __syncthreads();
  // (partials as 8-byte agent-scope words: written through / read past the non-coherent L2s - no cache-wide fence, see bn_reduce_kernel)
  unsigned long long* part = reinterpret_cast<unsigned long long*>(partial);
  if (threadIdx.x == 0) {
    lfdm_agent_store_u64(part + blockIdx.x, (unsigned long long)__float_as_uint((s_p[0] + s_p[1]) + (s_p[2] + s_p[3])));
    LFDM_DRAIN_STORES();
    const bool last = lfdm_ticket_take(ticket) == gridDim.x - 1;
    if (last) lfdm_ticket_reset(ticket);
    s_last = last ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  double t = 0.0;
  for (unsigned i = threadIdx.x; i < gridDim.x; i += 256) t += (double)__uint_as_float((unsigned)lfdm_agent_load_u64(part + i));
  // fixed order: lane-strided partial sums folded through a wavefront reduction of doubles in LDS
  __shared__ double s_d[256];
  s_d[threadIdx.x] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < 256; ++i) tot += s_d[i];
    *out = (float)(tot * (double)scale);
  }
}

__global__ __launch_bounds__(256) void l1_mean_bwd_kernel(const float4* __restrict__ x, const float4* __restrict__ y, int64_t n4, float scale,
                                                          const float* __restrict__ gout, float4* __restrict__ dx) {
  const float g = *gout * scale;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 a = x[i], b = y[i];
    float4 o;
    o.x = a.x > b.x ? g : (a.x < b.x ? -g : 0.f);
    o.y = a.y > b.y ? g : (a.y < b.y ? -g : 0.f);
    o.z = a.z > b.z ? g : (a.z < b.z ? -g : 0.f);
    o.w = a.w > b.w ? g : (a.w < b.w ? -g : 0.f);
    dx[i] = o;
  }
}

// im2col of channels-last rows with FEW channels (c % 4 == 0, c <= 16): out[(n, qy, qx)][tap * c + ch] = x[n, qy + ky - pad, qx + kx - pad][ch]
// (zero outside), stride 1.  Turns the weight gradient of a k x k convolution whose input (or, with the roles swapped, output) has a
// handful of channels into ONE 1x1 weight-gradient GEMM over k*k*c columns: the per-tap kernel pads 4 channels to its 64-wide tile
// (16x the work for the 7x7 RGB convolutions of the generator, LFAE/modules/generator.py:37,56).  One thread per (output row, tap, float4).
__global__ __launch_bounds__(256) void im2col_cl_kernel(const float* __restrict__ x, float* __restrict__ out, int n_img, int h, int w, int c,
                                                        int ldx, int k, int pad, int hq, int wq) {
  const int c4n = c >> 2, per_row = k * k * c4n;
  const int64_t total = (int64_t)n_img * hq * wq * per_row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int e = (int)(i % per_row);
    int64_t r = i / per_row;
    const int tap = e / c4n, c4 = e - tap * c4n;
    const int qx = (int)(r % wq);
    int64_t t = r / wq;
    const int qy = (int)(t % hq);
    const int n = (int)(t / hq);
    const int iy = qy + tap / k - pad, ix = qx + tap % k - pad;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy >= 0 && iy < h && ix >= 0 && ix < w) v = *reinterpret_cast<const float4*>(x + (((int64_t)n * h + iy) * w + ix) * ldx + 4 * c4);
    *reinterpret_cast<float4*>(out + r * ((int64_t)k * k * c) + tap * c + 4 * c4) = v;
  }
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

// =====================================================================================================================================
// C ABI
// =====================================================================================================================================
extern "C" size_t lfdm_batchnorm_train_ws_bytes(int64_t rows, int channels, int segments) {
  if (rows <= 0 || channels <= 0 || segments <= 0) return 0;
  const BnGeom g = bn_geom(rows, channels);
  return (size_t)segments * ((size_t)g.ctiles * g.nchunk * 2 * g.tw * sizeof(float) + (size_t)g.ctiles * g.ngroups * 2 * g.tw * sizeof(double)) +
         16;
}

namespace {
int bn_fill(BnArgs& a, const float* x, int64_t rows, int channels, int segments, int ldx, const float* gamma, const float* beta, void* ws,
            size_t ws_bytes, unsigned* tickets, const char* who) {
  if (!x || !gamma || !beta || !ws || !tickets || rows <= 0 || channels <= 0 || channels % 4 != 0 || channels > 64 * 32 || ldx % 4 != 0 ||
      segments <= 0 || segments > 64 || ldx < channels || !aligned16(x) || !aligned16(gamma) || !aligned16(beta) ||
      ws_bytes < lfdm_batchnorm_train_ws_bytes(rows, channels, segments)) {
    (void)who;
    lfdm_set_error("batchnorm_train: rows > 0 (per segment), 1..64 segments, channels % 4 == 0 (<= 2048), 16-byte aligned x / gamma / beta, "
                   "workspace of lfdm_batchnorm_train_ws_bytes and a zeroed ticket array of LFDM_BN_TICKETS words");
    return LFDM_EINVAL;
  }
  const BnGeom g = bn_geom(rows, channels);
  if ((int64_t)segments * g.ctiles * g.ngroups > LFDM_BN_TICKETS) {
    lfdm_set_error("batchnorm_train: segments x channel tiles x chunk groups exceeds LFDM_BN_TICKETS");
    return LFDM_EINVAL;
  }
  a.segments = segments;
  a.seg_p1 = (int64_t)g.ctiles * g.nchunk * 2 * g.tw;
  a.seg_p2 = (int64_t)g.ctiles * g.ngroups * 2 * g.tw;
  a.seg_tk = g.ctiles * g.ngroups;
  a.x = x; a.rows = rows; a.c = channels; a.ldx = ldx; a.gamma = gamma; a.beta = beta;
  a.q = g.q; a.chunk_rows = g.chunk_rows; a.nchunk = g.nchunk; a.ngroups = g.ngroups;
  // doubles first (alignment), then the float partials
  a.part2 = reinterpret_cast<double*>((((uintptr_t)ws) + 15) & ~(uintptr_t)15);
  a.part1 = reinterpret_cast<float*>(a.part2 + (size_t)segments * a.seg_p2);
  a.tickets = tickets;
  return LFDM_OK;
}
}  // namespace

extern "C" int lfdm_batchnorm_train_fwd_cl_f32(const float* x, float* y, int64_t rows, int channels, int segments, int ldx, int ldy,
                                               const float* gamma,
                                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                               int relu, float* stat, void* ws, size_t ws_bytes, unsigned* tickets, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BnArgs a = {};
  const int rc = bn_fill(a, x, rows, channels, segments, ldx, gamma, beta, ws, ws_bytes, tickets, "batchnorm_train_fwd");
  if (rc) return rc;
  if (!y || !stat || ldy % 4 != 0 || ldy < channels || !aligned16(y) || ((running_mean == nullptr) != (running_var == nullptr))) {
    lfdm_set_error("batchnorm_train_fwd: y (16-byte aligned, ldy % 4 == 0) and stat (segments * 2 * channels floats) are required; running_mean and "
                   "running_var come together");
    return LFDM_EINVAL;
  }
  a.out = y; a.ldo = ldy; a.stat = stat; a.running_mean = running_mean; a.running_var = running_var; a.momentum = momentum; a.eps = eps;
  a.relu = relu;
  const BnGeom g = bn_geom(rows, channels);
  const dim3 grid((unsigned)g.nchunk, (unsigned)g.ctiles, (unsigned)segments);
  LFDM_LAUNCH((bn_reduce_kernel<0>), grid, dim3(256), 0, stream, a);
  LFDM_LAUNCH((bn_apply_kernel<0>), grid, dim3(256), 0, stream, a);
  return lfdm_check_launch("batchnorm_train_fwd");
}

extern "C" int lfdm_batchnorm_train_bwd_cl_f32(const float* x, const float* dy, float* dx, int64_t rows, int channels, int segments, int ldx,
                                               int lddy, int lddx, const float* dx_add, int ldadd, const float* gamma, const float* beta, const float* stat, int relu, float* dgamma,
                                               float* dbeta, void* ws, size_t ws_bytes, unsigned* tickets, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BnArgs a = {};
  const int rc = bn_fill(a, x, rows, channels, segments, ldx, gamma, beta, ws, ws_bytes, tickets, "batchnorm_train_bwd");
  if (rc) return rc;
  if (!dy || !dx || !stat || lddy % 4 != 0 || lddx % 4 != 0 || lddy < channels || lddx < channels || !aligned16(dy) || !aligned16(dx) ||
      !aligned16(stat)) {
    lfdm_set_error("batchnorm_train_bwd: dy, dx (16-byte aligned, leading dimensions % 4 == 0) and the forward's stat are required");
    return LFDM_EINVAL;
  }
  if (dx_add && (ldadd % 4 != 0 || ldadd < channels || !aligned16(dx_add))) {
    lfdm_set_error("batchnorm_train_bwd: dx_add must be 16-byte aligned with a leading dimension % 4 == 0");
    return LFDM_EINVAL;
  }
  a.add = dx_add; a.ldadd = ldadd;
  a.dy = dy; a.lddy = lddy; a.out = dx; a.ldo = lddx; a.stat = const_cast<float*>(stat); a.relu = relu; a.dgamma = dgamma; a.dbeta = dbeta;
  const BnGeom g = bn_geom(rows, channels);
  const dim3 grid((unsigned)g.nchunk, (unsigned)g.ctiles, (unsigned)segments);
  LFDM_LAUNCH((bn_reduce_kernel<1>), grid, dim3(256), 0, stream, a);
  LFDM_LAUNCH((bn_apply_kernel<1>), grid, dim3(256), 0, stream, a);
  return lfdm_check_launch("batchnorm_train_bwd");
}

namespace {
int blur_fill(BlurArgs& a, const lfdm_blur_params* p, bool bwd) {
  if (!p || !p->wgt || p->n_img <= 0 || p->n_img > 65535 || p->channels <= 0 || p->c_store < p->channels || p->c_store > 65535 || p->h <= 0 ||
      p->w <= 0 || p->k <= 0 || p->k * p->k > BLUR_MAX_TAPS || p->stride <= 0 || p->pad_lo < 0 || p->pad_hi < 0 ||
      (bwd ? (!p->dy || !p->dx) : (!p->x || !p->out))) {
    lfdm_set_error("blur_down: bad arguments (n_img, c_store <= 65535; k*k <= 1024)");
    return LFDM_EINVAL;
  }
  const int hf = p->h + p->pad_lo + p->pad_hi - p->k + 1, wf = p->w + p->pad_lo + p->pad_hi - p->k + 1;
  if (hf <= 0 || wf <= 0) { lfdm_set_error("blur_down: kernel larger than the padded input"); return LFDM_EINVAL; }
  a.x = p->x; a.wgt = p->wgt; a.out = p->out; a.dy = p->dy; a.dx = p->dx;
  a.xs_n = p->xs_n; a.xs_c = p->xs_c; a.xs_h = p->xs_h; a.xs_w = p->xs_w;
  a.os_n = p->os_n; a.os_c = p->os_c; a.os_h = p->os_h; a.os_w = p->os_w;
  a.channels = p->channels; a.c_store = p->c_store; a.h = p->h; a.w = p->w; a.k = p->k; a.pad = p->pad_lo; a.stride = p->stride;
  a.ho = (hf + p->stride - 1) / p->stride; a.wo = (wf + p->stride - 1) / p->stride;
  a.scale = p->scale; a.bias = p->bias;
  return LFDM_OK;
}
}  // namespace

extern "C" int lfdm_blur_down_fwd_f32(const lfdm_blur_params* p, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BlurArgs a = {};
  const int rc = blur_fill(a, p, false);
  if (rc) return rc;
  LFDM_LAUNCH(blur_down_kernel, dim3((unsigned)((a.ho * a.wo + 255) / 256), (unsigned)a.c_store, (unsigned)p->n_img), dim3(256), 0, stream, a);
  return lfdm_check_launch("blur_down_fwd");
}

extern "C" int lfdm_blur_down_bwd_f32(const lfdm_blur_params* p, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BlurArgs a = {};
  const int rc = blur_fill(a, p, true);
  if (rc) return rc;
  LFDM_LAUNCH(blur_down_bwd_kernel, dim3((unsigned)((a.h * a.w + 255) / 256), (unsigned)a.channels, (unsigned)p->n_img), dim3(256), 0, stream, a);
  return lfdm_check_launch("blur_down_bwd");
}

extern "C" int lfdm_absmax_f32(const float* x, int64_t rows, int channels, int64_t ld, unsigned* out_bits, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !out_bits || rows <= 0 || channels <= 0 || ld < channels) { lfdm_set_error("absmax: bad arguments"); return LFDM_EINVAL; }
  int64_t blocks = (rows * channels + 256 * 16 - 1) / (256 * 16);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  LFDM_LAUNCH(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, rows, channels, ld, out_bits);
  return lfdm_check_launch("absmax");
}

extern "C" int lfdm_warp_bwd_f32(const lfdm_warp_bwd_params* pp, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!pp) { lfdm_set_error("warp_bwd: null params"); return LFDM_EINVAL; }
  const lfdm_warp_bwd_params& p = *pp;
  if (!p.src || !p.dout || !p.flow_x || !p.flow_y || !p.dmaps || p.n_img <= 0 || p.n_img > 65535 || p.h <= 0 || p.w <= 0 || p.c <= 0 || p.fh <= 0 ||
      p.fw <= 0 || p.n_div <= 0) {
    lfdm_set_error("warp_bwd: src, dout, flow_x, flow_y, dmaps and positive sizes are required (n_img <= 65535)");
    return LFDM_EINVAL;
  }
  const int hw = p.h * p.w;
  if (p.layout_cl) {
    const int g = p.c / 4;
    if (p.c % 4 != 0 || g > 64 || (g & (g - 1)) != 0 || (256 % g) != 0 || p.n_div != 1 || p.ld_src % 4 != 0 || p.ld_dout % 4 != 0 ||
        (p.prev && p.ld_prev % 4 != 0) || (p.dprev && p.ld_dprev % 4 != 0) || !aligned16(p.src) || !aligned16(p.dout) ||
        (p.prev && !aligned16(p.prev)) || (p.dprev && !aligned16(p.dprev)) || (p.dsrc_fix && (!p.amax_bits || (((uintptr_t)p.dsrc_fix) & 7))) ||
        (int64_t)hw * g > 0x7fffffff) {
      lfdm_set_error("warp_bwd (channels-last): C / 4 a power of two <= 64, 16-byte aligned rows, n_div == 1, dsrc_fix needs amax_bits");
      return LFDM_EINVAL;
    }
    int64_t blocks = ((int64_t)hw * g + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    LFDM_LAUNCH(warp_bwd_cl_kernel, dim3((unsigned)blocks, (unsigned)p.n_img), dim3(256), 0, stream, p);
  } else {
    if (p.dsrc_fix) { lfdm_set_error("warp_bwd (strided): the gradient of the sampled tensor is only produced by the channels-last form"); return LFDM_EINVAL; }
    LFDM_LAUNCH(warp_bwd_px_kernel, dim3((unsigned)((hw + 255) / 256), (unsigned)p.n_img), dim3(256), 0, stream, p);
  }
  return lfdm_check_launch("warp_bwd");
}

extern "C" int lfdm_fix_finalize_f32(long long* acc, float* out, int64_t rows, int channels, int64_t ld, unsigned* amax_bits, int64_t count,
                                     unsigned* ticket, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!acc || !out || !amax_bits || !ticket || rows <= 0 || channels <= 0 || ld < channels || count <= 0) {
    lfdm_set_error("fix_finalize: bad arguments");
    return LFDM_EINVAL;
  }
  int64_t blocks = (rows * channels + 256 * 8 - 1) / (256 * 8);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  LFDM_LAUNCH(fix_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, acc, out, rows, channels, ld, amax_bits, count, ticket);
  return lfdm_check_launch("fix_finalize");
}

extern "C" int lfdm_resize_adjoint_f32(const float* dhigh, float* dlow, int planes, int h, int w, int fh, int fw, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!dhigh || !dlow || planes <= 0 || planes > 65535 || h <= 0 || w <= 0 || fh <= 0 || fw <= 0) {
    lfdm_set_error("resize_adjoint: bad arguments (planes <= 65535)");
    return LFDM_EINVAL;
  }
  LFDM_LAUNCH(resize_adjoint_kernel, dim3((unsigned)((fh * fw + 255) / 256), (unsigned)planes), dim3(256), 0, stream, dhigh, dlow, h, w, fh, fw);
  return lfdm_check_launch("resize_adjoint");
}

namespace {
int gs_fill(GsArgs& a, const lfdm_grid_sample_params* p, bool bwd) {
  if (!p || !p->x || !p->grid || p->n_img <= 0 || p->n_img > 65535 || p->channels <= 0 || p->h <= 0 || p->w <= 0 || p->ho <= 0 || p->wo <= 0 ||
      p->n_div <= 0 || p->pad_mode < 0 || p->pad_mode > 1 || (bwd ? (!p->dout || !p->dgrid) : !p->out)) {
    lfdm_set_error("grid_sample: bad arguments (n_img <= 65535, pad_mode 0 zeros / 1 reflection)");
    return LFDM_EINVAL;
  }
  a.x = p->x; a.grid = p->grid; a.out = p->out; a.dout = p->dout; a.dgrid = p->dgrid;
  a.xs_n = p->xs_n; a.xs_c = p->xs_c; a.xs_h = p->xs_h; a.xs_w = p->xs_w;
  a.os_n = p->os_n; a.os_c = p->os_c; a.os_h = p->os_h; a.os_w = p->os_w;
  a.c = p->channels; a.h = p->h; a.w = p->w; a.ho = p->ho; a.wo = p->wo; a.n_div = p->n_div; a.pad_mode = p->pad_mode;
  return LFDM_OK;
}
}  // namespace

extern "C" int lfdm_grid_sample_fwd_f32(const lfdm_grid_sample_params* p, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GsArgs a = {};
  const int rc = gs_fill(a, p, false);
  if (rc) return rc;
  LFDM_LAUNCH((grid_sample_kernel<false>), dim3((unsigned)((a.ho * a.wo + 255) / 256), (unsigned)p->n_img), dim3(256), 0, stream, a);
  return lfdm_check_launch("grid_sample_fwd");
}

extern "C" int lfdm_grid_sample_bwd_f32(const lfdm_grid_sample_params* p, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GsArgs a = {};
  const int rc = gs_fill(a, p, true);
  if (rc) return rc;
  if (p->pad_mode != 0) { lfdm_set_error("grid_sample_bwd: zeros padding only (the reflection warp of model.py:118-122 has no gradient)"); return LFDM_EINVAL; }
  LFDM_LAUNCH((grid_sample_kernel<true>), dim3((unsigned)((a.ho * a.wo + 255) / 256), (unsigned)p->n_img), dim3(256), 0, stream, a);
  return lfdm_check_launch("grid_sample_bwd");
}

extern "C" int lfdm_svd2x2_sym_bwd_f32(const float* u, const float* s, const float* gu, const float* gs, float* ga, int64_t n,
                                       lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!u || !s || !ga || n <= 0 || (!gu && !gs)) { lfdm_set_error("svd2x2_sym_bwd: u, s, ga and at least one of gu / gs are required"); return LFDM_EINVAL; }
  LFDM_LAUNCH(svd2x2_sym_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, u, s, gu, gs, ga, n);
  return lfdm_check_launch("svd2x2_sym_bwd");
}

extern "C" int lfdm_pool2_cl_f32(const float* x, const float* aux, float* out, int n_img, int h, int w, int channels, int mode,
                                 lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !out || n_img <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 1) || channels <= 0 || channels % 4 != 0 || mode < 0 || mode > 5 ||
      (mode == 5 && !aux) || !aligned16(x) || !aligned16(out) || (aux && !aligned16(aux))) {
    lfdm_set_error("pool2: even fine resolution (h, w), channels % 4 == 0, mode 0..5 (5 needs aux = dy), 16-byte aligned rows");
    return LFDM_EINVAL;
  }
  const int64_t total = (int64_t)n_img * (h / 2) * (w / 2) * (channels / 4);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 65535 * 4) blocks = 65535 * 4;
  const dim3 grid((unsigned)blocks);
  switch (mode) {
    case 0: LFDM_LAUNCH((pool2_kernel<0>), grid, dim3(256), 0, stream, x, aux, out, n_img, h, w, channels); break;
    case 1: LFDM_LAUNCH((pool2_kernel<1>), grid, dim3(256), 0, stream, x, aux, out, n_img, h, w, channels); break;
    case 2: LFDM_LAUNCH((pool2_kernel<2>), grid, dim3(256), 0, stream, x, aux, out, n_img, h, w, channels); break;
    case 3: LFDM_LAUNCH((pool2_kernel<3>), grid, dim3(256), 0, stream, x, aux, out, n_img, h, w, channels); break;
    case 4: LFDM_LAUNCH((pool2_kernel<4>), grid, dim3(256), 0, stream, x, aux, out, n_img, h, w, channels); break;
    default: LFDM_LAUNCH((pool2_kernel<5>), grid, dim3(256), 0, stream, x, aux, out, n_img, h, w, channels); break;
  }
  return lfdm_check_launch("pool2");
}

extern "C" int lfdm_relu_bwd_f32(const float* y, const float* dy, float* out, int64_t n, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!y || !dy || !out || n <= 0 || n % 4 != 0 || !aligned16(y) || !aligned16(dy) || !aligned16(out)) {
    lfdm_set_error("relu_bwd: n % 4 == 0 and 16-byte aligned tensors");
    return LFDM_EINVAL;
  }
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 65535 * 4) blocks = 65535 * 4;
  LFDM_LAUNCH(relu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const float4*>(y), reinterpret_cast<const float4*>(dy),
              reinterpret_cast<float4*>(out), n / 4);
  return lfdm_check_launch("relu_bwd");
}

extern "C" int lfdm_l1_mean_fwd_f32(const float* x, const float* y, int64_t n, float weight, float* out, void* ws, size_t ws_bytes,
                                    unsigned* ticket, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !y || !out || !ws || !ticket || n <= 0 || n % 4 != 0 || ws_bytes < L1_BLOCKS * sizeof(unsigned long long) || (((uintptr_t)ws) & 7) || !aligned16(x) || !aligned16(y)) {
    lfdm_set_error("l1_mean_fwd: n % 4 == 0, 16-byte aligned x / y, an 8 KB workspace (8-byte aligned) and one zeroed ticket word");
    return LFDM_EINVAL;
  }
  int64_t blocks = (n / 4 + 256 * 8 - 1) / (256 * 8);
  if (blocks > L1_BLOCKS) blocks = L1_BLOCKS;
  if (blocks < 1) blocks = 1;
  LFDM_LAUNCH(l1_mean_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(y),
              n / 4, weight / (float)n, reinterpret_cast<float*>(ws), ticket, out);
  return lfdm_check_launch("l1_mean_fwd");
}

extern "C" int lfdm_l1_mean_bwd_f32(const float* x, const float* y, int64_t n, float weight, const float* gout, float* dx, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !y || !gout || !dx || n <= 0 || n % 4 != 0 || !aligned16(x) || !aligned16(y) || !aligned16(dx)) {
    lfdm_set_error("l1_mean_bwd: n % 4 == 0 and 16-byte aligned tensors");
    return LFDM_EINVAL;
  }
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 65535 * 4) blocks = 65535 * 4;
  LFDM_LAUNCH(l1_mean_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(y),
              n / 4, weight / (float)n, gout, reinterpret_cast<float4*>(dx));
  return lfdm_check_launch("l1_mean_bwd");
}

extern "C" int lfdm_im2col_cl_f32(const float* x, float* out, int n_img, int h, int w, int channels, int ldx, int k, int pad, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int hq = h + 2 * pad - k + 1, wq = w + 2 * pad - k + 1;
  if (!x || !out || n_img <= 0 || h <= 0 || w <= 0 || channels <= 0 || channels % 4 != 0 || channels > 16 || ldx % 4 != 0 || ldx < channels ||
      k <= 0 || pad < 0 || hq <= 0 || wq <= 0 || !aligned16(x) || !aligned16(out)) {
    lfdm_set_error("im2col_cl: channels % 4 == 0 and <= 16, 16-byte aligned rows, stride 1");
    return LFDM_EINVAL;
  }
  const int64_t total = (int64_t)n_img * hq * wq * k * k * (channels / 4);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 65535 * 8) blocks = 65535 * 8;
  LFDM_LAUNCH(im2col_cl_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, out, n_img, h, w, channels, ldx, k, pad, hq, wq);
  return lfdm_check_launch("im2col_cl");
}
