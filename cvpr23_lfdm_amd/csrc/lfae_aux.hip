// Small LFAE helpers of the training pseudo-ground-truth pass (include/lfdm_hip.h).
//
// lfdm_depthwise_down_planar_f32 - AntiAliasInterpolation2d (LFAE/modules/util.py:217-264): zero-pad (ka, kb),
// depthwise k x k Gaussian, keep every `stride`-th pixel.  The reference convolves at full resolution and
// then subsamples ([::4]); only the kept outputs are computed here (1/16 of the work at scale 0.25), one
// thread per output pixel, filter in LDS.  (Through ATen this op became 875 im2col + tiny-GEMM launch pairs
// per training step in MIOpen - 23 ms for 0.17 GFLOP.)
#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

constexpr int DW_MAX_TAPS = 32 * 32;

// grid (ceil(ho*wo/256), C, N)
__global__ __launch_bounds__(256) void depthwise_down_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                             float* __restrict__ out, int channels, int h, int w, int k,
                                                             int pad, int stride, int ho, int wo) {
  __shared__ float s_w[DW_MAX_TAPS];
  const int c = blockIdx.y, n = blockIdx.z;
  for (int i = threadIdx.x; i < k * k; i += 256) s_w[i] = wgt[(int64_t)c * k * k + i];
  __syncthreads();
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= ho * wo) return;
  const int oy = o / wo, ox = o - oy * wo;
  const float* plane = x + ((int64_t)n * channels + c) * h * w;
  float acc = 0.f;
  for (int ky = 0; ky < k; ++ky) {
    const int iy = oy * stride + ky - pad;
    if (iy < 0 || iy >= h) continue;
    for (int kx = 0; kx < k; ++kx) {
      const int ix = ox * stride + kx - pad;
      if (ix >= 0 && ix < w) acc = fmaf(plane[iy * w + ix], s_w[ky * k + kx], acc);
    }
  }
  out[((int64_t)n * channels + c) * ho * wo + o] = acc;
}


// ---------------------------------------------------------------------------------------------------------------------
// PixelwiseFlowPredictor head and tail (LFAE/modules/pixelwise_flow_predictor.py:48-128) as two kernels instead of ~120
// element-wise ATen launches over (N, K+1, h, w[, 2]) tensors (N = B*T = 320 frames per training step):
//  A. lfdm_lfae_motion_inputs_f32: heat-map representation (:48-65), sparse motions incl. the background homography
//     (:67-93) and the K+1 warped copies of the source image (:95-102), written straight into the hourglass' channels-last
//     input rows (channel kk*(1+c) + {heat, image channels}) and as `sparse` (N, K+1, h, w, 2) for the tail;
//  B. lfdm_lfae_motion_combine_f32: softmax over the K+1 mask logits, flow = sum_k mask_k * sparse_k, sigmoid of the occlusion
//     logit, reading the head convolution's channels-last rows in place.
constexpr int MOT_MAX_K = 32;

// grid (ceil(h*w/256), N).  c == 3 image channels (one float4 per region and pixel).
__global__ __launch_bounds__(256) void lfae_motion_inputs_kernel(
    const float* __restrict__ src_img, const float* __restrict__ drv_shift, const float* __restrict__ drv_covar,
    const float* __restrict__ drv_affine, const float* __restrict__ src_shift, const float* __restrict__ src_covar,
    const float* __restrict__ src_affine, const float* __restrict__ bg, float region_var, int revert_axis_swap, int frames,
    int K, int h, int w, float* __restrict__ rows, int ld, float* __restrict__ sparse) {
  __shared__ float s_par[MOT_MAX_K][16];      // per region: ds(2) ss(2) inv_d(4) inv_s(4) A(4)
  __shared__ float s_bg[9];
  const int n = blockIdx.y, b = n / frames;
  if ((int)threadIdx.x < K) {
    const int k = threadIdx.x;
    float* q = s_par[k];
    const float* dsh = drv_shift + ((int64_t)n * K + k) * 2;
    const float* ssh = src_shift + ((int64_t)b * K + k) * 2;
    q[0] = dsh[0]; q[1] = dsh[1]; q[2] = ssh[0]; q[3] = ssh[1];
    if (drv_covar) {
      const float* cd = drv_covar + ((int64_t)n * K + k) * 4;
      const float* cs = src_covar + ((int64_t)b * K + k) * 4;
      const float dd = cd[0] * cd[3] - cd[1] * cd[2], dsv = cs[0] * cs[3] - cs[1] * cs[2];
      q[4] = cd[3] / dd; q[5] = -cd[1] / dd; q[6] = -cd[2] / dd; q[7] = cd[0] / dd;
      q[8] = cs[3] / dsv; q[9] = -cs[1] / dsv; q[10] = -cs[2] / dsv; q[11] = cs[0] / dsv;
    }
    q[12] = 1.f; q[13] = 0.f; q[14] = 0.f; q[15] = 1.f;
    if (drv_affine) {
      const float* ad = drv_affine + ((int64_t)n * K + k) * 4;
      const float* as = src_affine + ((int64_t)b * K + k) * 4;
      const float det = ad[0] * ad[3] - ad[1] * ad[2];
      const float i00 = ad[3] / det, i01 = -ad[1] / det, i10 = -ad[2] / det, i11 = ad[0] / det;
      float a00 = as[0] * i00 + as[1] * i10, a01 = as[0] * i01 + as[1] * i11;
      float a10 = as[2] * i00 + as[3] * i10, a11 = as[2] * i01 + as[3] * i11;
      if (revert_axis_swap) {
        const float sg = a00 > 0.f ? 1.f : (a00 < 0.f ? -1.f : 0.f);
        a00 *= sg; a01 *= sg; a10 *= sg; a11 *= sg;
      }
      q[12] = a00; q[13] = a01; q[14] = a10; q[15] = a11;
    }
  }
  if (bg && threadIdx.x >= 64 && threadIdx.x < 73) s_bg[threadIdx.x - 64] = bg[(int64_t)n * 9 + threadIdx.x - 64];
  __syncthreads();
  const int hw = h * w;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= hw) return;
  const int y = pix / w, x = pix - y * w;
  const float gx = 2.f * ((float)x / (float)(w - 1)) - 1.f, gy = 2.f * ((float)y / (float)(h - 1)) - 1.f;
  const float* img = src_img + (int64_t)b * 3 * hw;
  float4* orow = reinterpret_cast<float4*>(rows + ((int64_t)n * hw + pix) * ld);
  for (int kk = 0; kk <= K; ++kk) {
    float heat = 0.f, sx, sy;
    if (kk == 0) {
      sx = gx; sy = gy;
      if (bg) {
        const float hx = s_bg[0] * gx + s_bg[1] * gy + s_bg[2];
        const float hy = s_bg[3] * gx + s_bg[4] * gy + s_bg[5];
        const float hz = s_bg[6] * gx + s_bg[7] * gy + s_bg[8];
        sx = hx / hz; sy = hy / hz;
      }
    } else {
      const float* q = s_par[kk - 1];
      const float dx = gx - q[0], dy = gy - q[1], ex = gx - q[2], ey = gy - q[3];
      float ud, us;
      if (drv_covar) {
        ud = (dx * q[4] + dy * q[6]) * dx + (dx * q[5] + dy * q[7]) * dy;
        us = (ex * q[8] + ey * q[10]) * ex + (ex * q[9] + ey * q[11]) * ey;
      } else {
        ud = (dx * dx + dy * dy) / region_var;
        us = (ex * ex + ey * ey) / region_var;
      }
      heat = expf(-0.5f * ud) - expf(-0.5f * us);
      sx = q[12] * dx + q[13] * dy + q[2];
      sy = q[14] * dx + q[15] * dy + q[3];
    }
    *reinterpret_cast<float2*>(sparse + ((((int64_t)n * (K + 1) + kk) * hw + pix) * 2)) = make_float2(sx, sy);
    // F.grid_sample(bilinear, zeros, align_corners=False)
    const float fx = ((sx + 1.f) * (float)w - 1.f) * 0.5f, fy = ((sy + 1.f) * (float)h - 1.f) * 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float wx1 = fx - x0f, wy1 = fy - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    float v[3] = {0.f, 0.f, 0.f};
    // (coordinates far outside the image: every tap is out of range; the int conversion is only used when in range)
    const bool finite_ok = x0f > -4.f && x0f < (float)(w + 4) && y0f > -4.f && y0f < (float)(h + 4);
    if (finite_ok) {
      const int x0 = (int)x0f, y0 = (int)y0f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
          const float wt = ((t & 1) ? wx1 : wx0) * ((t >> 1) ? wy1 : wy0);
          const int o = yy * w + xx;
          v[0] += img[o] * wt; v[1] += img[hw + o] * wt; v[2] += img[2 * hw + o] * wt;
        }
      }
    }
    orow[kk] = make_float4(heat, v[0], v[1], v[2]);
  }
  for (int q4 = K + 1; q4 < ld / 4; ++q4) orow[q4] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// one thread per (n, pixel); heads: channels-last rows (N*h*w, ldh): columns [0, K] mask logits, column K+1 occlusion logit
__global__ __launch_bounds__(256) void lfae_motion_combine_kernel(const float* __restrict__ heads, int ldh,
                                                                  const float* __restrict__ sparse, int K, int hw, int64_t total,
                                                                  float* __restrict__ flow, float* __restrict__ occ) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t n = i / hw;
  const int pix = (int)(i - n * hw);
  const float* row = heads + i * ldh;
  float m = row[0];
  for (int k = 1; k <= K; ++k) m = fmaxf(m, row[k]);
  float sum = 0.f, fx = 0.f, fy = 0.f;
  for (int k = 0; k <= K; ++k) sum += expf(row[k] - m);
  for (int k = 0; k <= K; ++k) {
    const float pk = expf(row[k] - m) / sum;
    const float2 sp = *reinterpret_cast<const float2*>(sparse + (((n * (K + 1) + k) * hw + pix) * 2));
    fx += sp.x * pk;
    fy += sp.y * pk;
  }
  *reinterpret_cast<float2*>(flow + i * 2) = make_float2(fx, fy);
  if (occ) occ[i] = 1.0f / (1.0f + expf(-row[K + 1]));
}


// ---------------------------------------------------------------------------------------------------------------------
// RegionPredictor tail (LFAE/modules/region_predictor.py:16-25,60-96): spatial softmax of the K region logits, centre,
// covariance, and U sqrt(S) of the covariance's SVD - with torch.svd's (LAPACK xGESDD) sign convention, which the
// reference's `affine` inherits - for all N frames in one launch.  The SVD is the closed 2x2 path of LAPACK (xGEBRD's one
// Householder reflection, xBDSQR's split test, xLASV2, the selection-sort swap), written branch-free exactly like the
// element-wise torch formulation it replaces (~200 tiny launches per call) so that both select the same branches.
struct Svd2 {
  float u00, u01, u10, u11, s1, s2;
};

__device__ inline float sign1(float x) { return x < 0.f ? -1.f : 1.f; }      // Fortran SIGN(1, x)

#pragma clang fp contract(off)
__device__ void lasv2(float f, float g, float h, float& ssmin_o, float& ssmax_o, float& snl_o, float& csl_o) {
  float fa = fabsf(f), ha = fabsf(h);
  const bool swap = ha > fa;
  const float ft = swap ? h : f, ht = swap ? f : h;
  { const float t0 = fa; fa = swap ? ha : fa; ha = swap ? t0 : ha; }
  const float gt = g, ga = fabsf(g);
  const bool diag = ga == 0.f;
  const float eps = 1.1920928955078125e-07f / 2.f;          // xLAMCH('E')
  const bool gbig = !diag && ga > fa;
  const float ft_ = ft == 0.f ? 1.f : ft;
  const float ga_ = diag ? 1.f : ga;
  const float gt_ = diag ? 1.f : gt;
  const bool large = gbig && (fa / ga_) < eps;
  const float d = fa - ha;
  const float l = d == fa ? 1.f : d / (fa == 0.f ? 1.f : fa);
  const float m = gt / ft_;
  const float t = 2.f - l;
  const float mm = m * m, tt = t * t;
  const float sq = sqrtf(tt + mm);
  const float r = l == 0.f ? fabsf(m) : sqrtf(l * l + mm);
  const float a = 0.5f * (sq + r);
  float ssmin = ha / a, ssmax = fa * a;
  const float d_ = d == 0.f ? 1.f : d;
  const float t_tiny = l == 0.f ? 2.f * sign1(ft) * sign1(gt) : gt / (fabsf(d_) * sign1(ft)) + m / t;
  const float rl = (r + l) == 0.f ? 1.f : r + l;
  const float t_norm = (m / (sq + t) + m / rl) * (1.f + a);
  const float t2 = mm == 0.f ? t_tiny : t_norm;
  const float l2 = sqrtf(t2 * t2 + 4.f);
  float crt = 2.f / l2, srt = t2 / l2;
  float clt = (crt + srt * m) / a;
  float slt = (ht / ft_) * srt / a;
  if (large) {
    clt = 1.f; slt = ht / gt_; srt = 1.f; crt = ft / gt_;
    ssmax = ga;
    ssmin = ha > 1.f ? fa / (ga_ / (ha == 0.f ? 1.f : ha)) : (fa / ga_) * ha;
  }
  if (diag) { clt = 1.f; crt = 1.f; slt = 0.f; srt = 0.f; ssmax = fa; ssmin = ha; }
  csl_o = swap ? srt : clt;
  snl_o = swap ? crt : slt;
  ssmin_o = fabsf(ssmin);
  ssmax_o = fabsf(ssmax);
}

__device__ Svd2 svd2x2_sym_lapack(float a, float b, float c) {
  const float r = sqrtf(a * a + b * b);
  const float beta = -sign1(a) * r;
  const bool noref = b == 0.f;
  const float safe = noref ? 1.f : beta;
  const float h00 = noref ? 1.f : a / safe, h01 = noref ? 0.f : b / safe, h11 = noref ? 1.f : -a / safe;
  const float d1 = noref ? a : beta;
  const float e1 = h00 * b + h01 * c;
  const float d2 = h01 * b + h11 * c;
  float ssmin, ssmax, snl, csl;
  lasv2(d1, e1, d2, ssmin, ssmax, snl, csl);
  const float eps = 1.1920928955078125e-07f / 2.f;
  // tol = max(10, min(100, eps^(-1/8))) * eps, evaluated in double like the host expression it mirrors, then used as fp32
  const double epsd = (double)eps;
  double tol_d = pow(epsd, -0.125);
  tol_d = (tol_d < 100.0 ? tol_d : 100.0);
  tol_d = (tol_d > 10.0 ? tol_d : 10.0) * epsd;
  const float tol = (float)tol_d;
  const float ad1 = fabsf(d1), ad2 = fabsf(d2), ae = fabsf(e1);
  const float mu = ad2 * (ad1 / ((ad1 + ae) == 0.f ? 1.f : ad1 + ae));
  const float sminoa = (ad1 == 0.f ? 0.f : fminf(ad1, mu)) / 1.41421356237309514547f;
  const float tiny24 = 24.f * 1.17549435082228750797e-38f;
  const float thr = tol * sminoa;
  const bool split = ae <= (thr > tiny24 ? thr : tiny24);
  if (split) { csl = 1.f; snl = 0.f; }
  const float s1 = split ? ad1 : ssmax, s2 = split ? ad2 : ssmin;
  const float u00 = h00 * csl + h01 * snl, u01 = -h00 * snl + h01 * csl;
  const float u10 = h01 * csl + h11 * snl, u11 = -h01 * snl + h11 * csl;
  const bool sw = s2 > s1;
  Svd2 o;
  o.u00 = sw ? u01 : u00; o.u01 = sw ? u00 : u01;
  o.u10 = sw ? u11 : u10; o.u11 = sw ? u10 : u11;
  o.s1 = sw ? s2 : s1; o.s2 = sw ? s1 : s2;
  return o;
}
#pragma clang fp contract(fast)

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* s_red /*[4]*/) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const float o = __shfl_xor(v, m);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  __syncthreads();                                   // s_red free again
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float a = s_red[0], b = s_red[1], c = s_red[2], d = s_red[3];
  return is_max ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
}

__global__ __launch_bounds__(256) void svd2x2_sym_kernel(const float* __restrict__ abc, int64_t n, float* __restrict__ u,
                                                        float* __restrict__ sv) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Svd2 o = svd2x2_sym_lapack(abc[3 * i], abc[3 * i + 1], abc[3 * i + 2]);
  u[4 * i] = o.u00; u[4 * i + 1] = o.u01; u[4 * i + 2] = o.u10; u[4 * i + 3] = o.u11;
  sv[2 * i] = o.s1; sv[2 * i + 1] = o.s2;
}

// grid (K, N), 256 threads; logits: channels-last rows (N*h*w, ldh), column k
constexpr int RS_PER = 16;       // pixels per thread: h*w <= 4096
__global__ __launch_bounds__(256) void lfae_region_stats_kernel(const float* __restrict__ logits, int ldh, int K, int h, int w,
                                                                float temperature, float* __restrict__ heatmap,
                                                                float* __restrict__ shift, float* __restrict__ covar,
                                                                float* __restrict__ affine, float* __restrict__ u_out,
                                                                float* __restrict__ d_out) {
  __shared__ float s_red[4];
  const int k = blockIdx.x, n = blockIdx.y, hw = h * w;
  const float* src = logits + (int64_t)n * hw * ldh + k;
  float v[RS_PER];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < RS_PER; ++i) {
    const int pix = threadIdx.x + 256 * i;
    v[i] = pix < hw ? src[(int64_t)pix * ldh] / temperature : -3.0e38f;
    mx = fmaxf(mx, v[i]);
  }
  mx = block_reduce(mx, true, s_red);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < RS_PER; ++i) {
    const int pix = threadIdx.x + 256 * i;
    v[i] = pix < hw ? expf(v[i] - mx) : 0.f;
    sum += v[i];
  }
  sum = block_reduce(sum, false, s_red);
  float ax = 0.f, ay = 0.f;
  float* hm = heatmap + ((int64_t)n * K + k) * hw;
#pragma unroll
  for (int i = 0; i < RS_PER; ++i) {
    const int pix = threadIdx.x + 256 * i;
    if (pix < hw) {
      v[i] = v[i] / sum;
      hm[pix] = v[i];
      const int y = pix / w, x = pix - y * w;
      const float gx = 2.f * ((float)x / (float)(w - 1)) - 1.f, gy = 2.f * ((float)y / (float)(h - 1)) - 1.f;
      ax += v[i] * gx;
      ay += v[i] * gy;
    }
  }
  ax = block_reduce(ax, false, s_red);
  ay = block_reduce(ay, false, s_red);
  float cxx = 0.f, cxy = 0.f, cyy = 0.f;
#pragma unroll
  for (int i = 0; i < RS_PER; ++i) {
    const int pix = threadIdx.x + 256 * i;
    if (pix < hw) {
      const int y = pix / w, x = pix - y * w;
      const float gx = 2.f * ((float)x / (float)(w - 1)) - 1.f, gy = 2.f * ((float)y / (float)(h - 1)) - 1.f;
      const float dx = gx - ax, dy = gy - ay;
      cxx += dx * dx * v[i];
      cxy += dx * dy * v[i];
      cyy += dy * dy * v[i];
    }
  }
  cxx = block_reduce(cxx, false, s_red);
  cxy = block_reduce(cxy, false, s_red);
  cyy = block_reduce(cyy, false, s_red);
  if (threadIdx.x == 0) {
    const int64_t i = (int64_t)n * K + k;
    shift[2 * i] = ax; shift[2 * i + 1] = ay;
    covar[4 * i] = cxx; covar[4 * i + 1] = cxy; covar[4 * i + 2] = cxy; covar[4 * i + 3] = cyy;
    const Svd2 sv = svd2x2_sym_lapack(cxx, cxy, cyy);
    const float r1 = sqrtf(sv.s1), r2 = sqrtf(sv.s2);
    u_out[4 * i] = sv.u00; u_out[4 * i + 1] = sv.u01; u_out[4 * i + 2] = sv.u10; u_out[4 * i + 3] = sv.u11;
    d_out[4 * i] = r1; d_out[4 * i + 1] = 0.f; d_out[4 * i + 2] = 0.f; d_out[4 * i + 3] = r2;
    affine[4 * i] = sv.u00 * r1; affine[4 * i + 1] = sv.u01 * r2; affine[4 * i + 2] = sv.u10 * r1; affine[4 * i + 3] = sv.u11 * r2;
  }
}

}  // namespace

extern "C" int lfdm_depthwise_down_planar_f32(const float* x, const float* wgt, float* out, int n_img, int channels,
                                              int h, int w, int k, int pad_lo, int pad_hi, int stride,
                                              lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !wgt || !out || n_img <= 0 || channels <= 0 || h <= 0 || w <= 0 || k <= 0 || k * k > DW_MAX_TAPS || stride <= 0 ||
      pad_lo < 0 || pad_hi < 0 || n_img > 65535 || channels > 65535) {
    lfdm_set_error("depthwise_down: bad arguments");
    return LFDM_EINVAL;
  }
  // full-resolution output of the padded convolution has (h + pad_lo + pad_hi - k + 1) rows; keep every stride-th
  const int hf = h + pad_lo + pad_hi - k + 1, wf = w + pad_lo + pad_hi - k + 1;
  if (hf <= 0 || wf <= 0) { lfdm_set_error("depthwise_down: kernel larger than the padded input"); return LFDM_EINVAL; }
  const int ho = (hf + stride - 1) / stride, wo = (wf + stride - 1) / stride;
  LFDM_LAUNCH(depthwise_down_kernel, dim3((unsigned)((ho * wo + 255) / 256), (unsigned)channels, (unsigned)n_img), dim3(256), 0,
              stream, x, wgt, out, channels, h, w, k, pad_lo, stride, ho, wo);
  return lfdm_check_launch("depthwise_down");
}

extern "C" int lfdm_lfae_motion_inputs_f32(const float* src_img, const float* drv_shift, const float* drv_covar,
                                           const float* drv_affine, const float* src_shift, const float* src_covar,
                                           const float* src_affine, const float* bg, float region_var, int revert_axis_swap,
                                           int batch, int frames, int regions, int h, int w, float* rows, int ld,
                                           float* sparse, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t n = (int64_t)batch * frames;
  if (!src_img || !drv_shift || !src_shift || !rows || !sparse || batch <= 0 || frames <= 0 || regions <= 0 ||
      regions > MOT_MAX_K || h < 2 || w < 2 || n > 65535 || ld % 4 != 0 || ld < 4 * (regions + 1) ||
      ((drv_covar == nullptr) != (src_covar == nullptr)) || ((drv_affine == nullptr) != (src_affine == nullptr)) ||
      (!drv_covar && !(region_var > 0.f)) || (((uintptr_t)rows) & 15) != 0 || (((uintptr_t)sparse) & 7) != 0) {
    lfdm_set_error("lfae_motion_inputs: bad arguments (3-channel source, <= 32 regions, ld % 4 == 0 and >= 4*(regions+1))");
    return LFDM_EINVAL;
  }
  LFDM_LAUNCH(lfae_motion_inputs_kernel, dim3((unsigned)((h * w + 255) / 256), (unsigned)n), dim3(256), 0, stream, src_img,
              drv_shift, drv_covar, drv_affine, src_shift, src_covar, src_affine, bg, region_var, revert_axis_swap, frames, regions,
              h, w, rows, ld, sparse);
  return lfdm_check_launch("lfae_motion_inputs");
}

extern "C" int lfdm_lfae_motion_combine_f32(const float* heads, int ldh, const float* sparse, int n_img, int regions, int hw,
                                            float* flow, float* occ, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!heads || !sparse || !flow || n_img <= 0 || regions <= 0 || hw <= 0 || ldh < regions + 1 + (occ ? 1 : 0) ||
      (((uintptr_t)sparse) & 7) != 0 || (((uintptr_t)flow) & 7) != 0) {
    lfdm_set_error("lfae_motion_combine: bad arguments");
    return LFDM_EINVAL;
  }
  const int64_t total = (int64_t)n_img * hw;
  LFDM_LAUNCH(lfae_motion_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, heads, ldh, sparse, regions,
              hw, total, flow, occ);
  return lfdm_check_launch("lfae_motion_combine");
}

extern "C" int lfdm_lfae_region_stats_f32(const float* logits, int ldh, int n_img, int regions, int h, int w, float temperature,
                                          float* heatmap, float* shift, float* covar, float* affine, float* u, float* d,
                                          lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!logits || !heatmap || !shift || !covar || !affine || !u || !d || n_img <= 0 || n_img > 65535 || regions <= 0 ||
      ldh < regions || h < 2 || w < 2 || h * w > 256 * RS_PER || !(temperature > 0.f)) {
    lfdm_set_error("lfae_region_stats: bad arguments (h*w <= 4096)");
    return LFDM_EINVAL;
  }
  LFDM_LAUNCH(lfae_region_stats_kernel, dim3((unsigned)regions, (unsigned)n_img), dim3(256), 0, stream, logits, ldh, regions, h, w,
              temperature, heatmap, shift, covar, affine, u, d);
  return lfdm_check_launch("lfae_region_stats");
}

extern "C" int lfdm_svd2x2_sym_f32(const float* abc, int64_t n, float* u, float* s, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!abc || !u || !s || n <= 0) { lfdm_set_error("svd2x2_sym: bad arguments"); return LFDM_EINVAL; }
  LFDM_LAUNCH(svd2x2_sym_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, abc, n, u, s);
  return lfdm_check_launch("svd2x2_sym");
}
