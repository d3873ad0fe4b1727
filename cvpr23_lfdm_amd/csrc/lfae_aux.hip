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
