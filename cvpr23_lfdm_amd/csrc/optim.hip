// Fused Adam over one flat fp32 parameter buffer (include/lfdm_hip.h: lfdm_adam_step_f32) - the
// `optimizer_diff.step()` of DM/modules/video_flow_diffusion_model.py:113-114,188 (torch.optim.Adam,
// betas (0.9, 0.99), no amsgrad).  One launch updates all 42.7 M parameters: 16 B/lane streaming reads
// of (p, g, m, v), writes of (p, m, v) - HBM-bound, 28 B per parameter.  grad_scale folds the 1/world
// of the data-parallel gradient average (RCCL all-reduce is a sum) into the same pass.
#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float lr, float beta1, float beta2, float eps, float weight_decay,
                                                   float bias1, float bias2_sqrt, float grad_scale) {
  const float step_size = lr / bias1;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* pp = &pv.x;
    const float* gg = &gv.x;
    float* mm = &mv.x;
    float* vq = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float grad = gg[k] * grad_scale + weight_decay * pp[k];
      mm[k] = beta1 * mm[k] + (1.f - beta1) * grad;
      vq[k] = beta2 * vq[k] + (1.f - beta2) * grad * grad;
      const float denom = sqrtf(vq[k]) / bias2_sqrt + eps;
      pp[k] = pp[k] - step_size * (mm[k] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // tail (n % 4)
  const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n) {
    const float grad = g[t] * grad_scale + weight_decay * p[t];
    const float m1 = beta1 * m[t] + (1.f - beta1) * grad;
    const float v1 = beta2 * v[t] + (1.f - beta2) * grad * grad;
    m[t] = m1;
    v[t] = v1;
    p[t] = p[t] - step_size * (m1 / (sqrtf(v1) / bias2_sqrt + eps));
  }
}

}  // namespace

extern "C" int lfdm_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                  float grad_scale, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step < 1 || (((uintptr_t)param | (uintptr_t)grad |
      (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15)) {
    lfdm_set_error("adam: bad arguments (16-byte aligned flat buffers, step >= 1)");
    return LFDM_EINVAL;
  }
  // bias corrections in double on the host, like torch.optim.Adam's scalar path
  const double b1 = 1.0 - pow((double)beta1, (double)step);
  const double b2 = sqrt(1.0 - pow((double)beta2, (double)step));
  int64_t nb = ((n >> 2) + 255) / 256;
  if (nb < 1) nb = 1;
  if (nb > 4096) nb = 4096;
  LFDM_LAUNCH(adam_kernel, dim3((unsigned)nb), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2,
              eps, weight_decay, (float)b1, (float)b2, grad_scale);
  return lfdm_check_launch("adam");
}
