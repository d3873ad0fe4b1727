// Small-batch Linear layers of the conditioning path, forward AND backward, for the DM training step
// (include/lfdm_hip.h: lfdm_multi_linear_f32 / lfdm_multi_linear_bwd_f32).
//
// Unet3D feeds every ResnetBlock a (scale, shift) pair from `mlp = SiLU -> Linear(cond_dim, 2*dim_out)` applied to the SAME
// (B, cond_dim) vector cat(time_emb, cond) (DM/modules/video_flow_diffusion.py:230-233,240-245,562), and builds time_emb with
// `time_mlp = SinusoidalPosEmb -> Linear -> GELU -> Linear` (:441-447).  With B = 8 rows these are weight-streaming problems (the 18
// block MLPs of the MUG configuration hold 40 MB of weights), not GEMMs: one launch reads every weight once.
//   forward      y_j = act(x) W_j^T + b_j                         one wave per output column, x staged in LDS once per 16 columns
//   backward     dW_j = dy_j^T act(x), db_j = colsum(dy_j)        thread = 4 input features, 16 columns per workgroup, float4 stores
//                dx   = act'(x) * sum_j dy_j W_j                  partial sums over 64 columns per workgroup, then a fixed-order reduce
// "multi": the blocks j share the input x (and its gradient), each has its own weight / bias / output tensors - the pointers travel by
// value in the kernel arguments (no device-side table to upload).  Sums are taken in a fixed order (no atomics).
#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

constexpr int ML_MAX = LFDM_MULTI_LINEAR_MAX;
constexpr int ML_ROWS = 16;        // rows (batch) supported
constexpr int ML_KMAX = 1024;      // input features supported (one float4 per thread of a 256-thread workgroup)

struct MlFwd {
  const float* x;
  const float* w[ML_MAX];
  const float* bias[ML_MAX];
  float* y[ML_MAX];
  int col0[ML_MAX + 1];            // first global column of block j; col0[nblk] = total
  int nblk, rows, k, act;
};

struct MlBwd {
  const float* x;
  const float* w[ML_MAX];
  const float* dy[ML_MAX];
  float* dw[ML_MAX];
  float* dbias[ML_MAX];
  int col0[ML_MAX + 1];
  int nblk, rows, k, act;
};

__device__ __forceinline__ float ml_act(float v, int act) {
  if (act == 1) return v / (1.0f + expf(-v));
  if (act == 2) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  return v;
}
__device__ __forceinline__ float ml_dact(float v, int act) {
  if (act == 1) {
    const float s = 1.0f / (1.0f + expf(-v));
    return s * (1.0f + v * (1.0f - s));
  }
  if (act == 2) return 0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * expf(-0.5f * v * v) * 0.39894228040143267794f;
  return 1.0f;
}

template <typename P>
__device__ __forceinline__ int ml_block_of(const P& p, int col) {      // uniform: the block that owns global column `col`
  int j = 0;
  while (j + 1 < p.nblk && col >= p.col0[j + 1]) ++j;
  return j;
}

// act(x) -> LDS, rows x k floats
__device__ __forceinline__ void ml_stage_x(float* ax, const float* x, int rows, int k, int act) {
  for (int i = threadIdx.x; i < rows * k / 4; i += 256) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x = ml_act(v.x, act); v.y = ml_act(v.y, act); v.z = ml_act(v.z, act); v.w = ml_act(v.w, act);
    reinterpret_cast<float4*>(ax)[i] = v;
  }
}

template <int R>      // R = rows the LDS tile and the accumulators are sized for (8 | 16)
__global__ __launch_bounds__(256) void multi_linear_fwd_kernel(MlFwd p) {
  __shared__ __attribute__((aligned(16))) float ax[R * ML_KMAX];      // rows * k used
  ml_stage_x(ax, p.x, p.rows, p.k, p.act);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = lfdm_uniform(threadIdx.x >> 6);
  const int k4n = p.k >> 2, total = p.col0[p.nblk];
  for (int cc = 0; cc < 4; ++cc) {
    const int col = (int)blockIdx.x * 16 + wave * 4 + cc;
    if (col >= total) break;
    const int j = ml_block_of(p, col);
    const int nl = col - p.col0[j], nj = p.col0[j + 1] - p.col0[j];
    const float4* wrow = reinterpret_cast<const float4*>(p.w[j] + (int64_t)nl * p.k);
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    for (int k4 = lane; k4 < k4n; k4 += 64) {
      const float4 wv = wrow[k4];
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (r < p.rows) {
          const float4 a = reinterpret_cast<const float4*>(ax)[r * k4n + k4];
          acc[r] += (a.x * wv.x + a.y * wv.y) + (a.z * wv.z + a.w * wv.w);
        }
    }
    const float bv = p.bias[j] ? p.bias[j][nl] : 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (r < p.rows) {
        const float s = wave_sum(acc[r]);
        if (lane == 0) p.y[j][(int64_t)r * nj + nl] = s + bv;
      }
  }
}

// dW and dbias: 16 columns per workgroup, thread = float4 of input features
template <int R>
__global__ __launch_bounds__(256) void multi_linear_dw_kernel(MlBwd p) {
  __shared__ __attribute__((aligned(16))) float ax[R * ML_KMAX];
  __shared__ float dyl[16 * ML_ROWS];                                 // [16 columns][ML_ROWS]
  const int tid = threadIdx.x, total = p.col0[p.nblk], k4n = p.k >> 2;
  ml_stage_x(ax, p.x, p.rows, p.k, p.act);
  {
    const int c = tid >> 4, r = tid & 15, col = (int)blockIdx.x * 16 + c;      // 16 columns x 16 rows
    float v = 0.f;
    if (col < total && r < p.rows) {
      const int j = ml_block_of(p, col);
      if (p.dy[j]) v = p.dy[j][(int64_t)r * (p.col0[j + 1] - p.col0[j]) + col - p.col0[j]];
    }
    dyl[c * ML_ROWS + r] = v;
  }
  __syncthreads();
  for (int c = 0; c < 16; ++c) {
    const int col = (int)blockIdx.x * 16 + c;
    if (col >= total) break;
    const int j = ml_block_of(p, col);
    const int nl = col - p.col0[j];
    if (p.dw[j] && tid < k4n) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (r < p.rows) {
          const float d = dyl[c * ML_ROWS + r];
          const float4 a = reinterpret_cast<const float4*>(ax)[r * k4n + tid];
          acc.x += d * a.x; acc.y += d * a.y; acc.z += d * a.z; acc.w += d * a.w;
        }
      reinterpret_cast<float4*>(p.dw[j] + (int64_t)nl * p.k)[tid] = acc;
    }
    if (p.dbias[j] && tid == 255) {
      float s = 0.f;
      for (int r = 0; r < p.rows; ++r) s += dyl[c * ML_ROWS + r];
      p.dbias[j][nl] = s;
    }
  }
}

// partial[blk][r][k] = sum over the workgroup's 64 columns of dy[r][col] * W[col][k]
template <int R>
__global__ __launch_bounds__(256) void multi_linear_dx_partial_kernel(MlBwd p, float* __restrict__ partial) {
  __shared__ float dyl[64 * ML_ROWS];
  const int tid = threadIdx.x, total = p.col0[p.nblk], k4n = p.k >> 2;
  for (int i = tid; i < 64 * ML_ROWS; i += 256) {
    const int c = i >> 4, r = i & 15, col = (int)blockIdx.x * 64 + c;
    float v = 0.f;
    if (col < total && r < p.rows) {
      const int j = ml_block_of(p, col);
      if (p.dy[j]) v = p.dy[j][(int64_t)r * (p.col0[j + 1] - p.col0[j]) + col - p.col0[j]];
    }
    dyl[i] = v;
  }
  __syncthreads();
  if (tid >= k4n) return;
  float4 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  int j = ml_block_of(p, (int)blockIdx.x * 64);
  for (int c = 0; c < 64; ++c) {
    const int col = (int)blockIdx.x * 64 + c;
    if (col >= total) break;
    while (col >= p.col0[j + 1]) ++j;
    if (!p.dy[j]) continue;
    const float4 wv = reinterpret_cast<const float4*>(p.w[j] + (int64_t)(col - p.col0[j]) * p.k)[tid];
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (r < p.rows) {
        const float d = dyl[c * ML_ROWS + r];
        acc[r].x += d * wv.x; acc[r].y += d * wv.y; acc[r].z += d * wv.z; acc[r].w += d * wv.w;
      }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (r < p.rows) reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * p.rows + r) * p.k)[tid] = acc[r];
}

// dx = act'(x) * sum_blk partial[blk]   (fixed order)
__global__ __launch_bounds__(256) void multi_linear_dx_reduce_kernel(const float* __restrict__ partial, int nparts, const float* __restrict__ x,
                                                                    float* __restrict__ dx, int n4, int act) {
  const int i = (int)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int s = 0; s < nparts; ++s) {
    const float4 v = reinterpret_cast<const float4*>(partial)[(int64_t)s * n4 + i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const float4 xv = reinterpret_cast<const float4*>(x)[i];
  acc.x *= ml_dact(xv.x, act); acc.y *= ml_dact(xv.y, act); acc.z *= ml_dact(xv.z, act); acc.w *= ml_dact(xv.w, act);
  reinterpret_cast<float4*>(dx)[i] = acc;
}

int ml_check(const lfdm_multi_linear_params& p, const char* who) {
  if (p.n_blocks <= 0 || p.n_blocks > ML_MAX || p.rows <= 0 || p.rows > ML_ROWS || p.k <= 0 || p.k % 4 != 0 || p.k > ML_KMAX || !p.x ||
      p.act < 0 || p.act > 2 || (((uintptr_t)p.x) & 15)) {
    (void)who;
    lfdm_set_error("multi_linear: 1..32 blocks, 1..16 rows, input features a multiple of 4 up to 1024, x 16-byte aligned, act in {0, 1, 2}");
    return LFDM_EINVAL;
  }
  for (int j = 0; j < p.n_blocks; ++j)
    if (!p.w[j] || p.n[j] <= 0 || (((uintptr_t)p.w[j]) & 15)) {
      lfdm_set_error("multi_linear: every block needs a 16-byte aligned weight and n > 0");
      return LFDM_EINVAL;
    }
  return LFDM_OK;
}

}  // namespace

extern "C" int lfdm_multi_linear_f32(const lfdm_multi_linear_params* pp, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!pp) { lfdm_set_error("multi_linear: null params"); return LFDM_EINVAL; }
  const lfdm_multi_linear_params& p = *pp;
  int rc = ml_check(p, "multi_linear");
  if (rc) return rc;
  MlFwd a;
  a.x = p.x; a.nblk = p.n_blocks; a.rows = p.rows; a.k = p.k; a.act = p.act;
  a.col0[0] = 0;
  for (int j = 0; j < p.n_blocks; ++j) {
    if (!p.y[j]) { lfdm_set_error("multi_linear: a block has no output tensor"); return LFDM_EINVAL; }
    a.w[j] = p.w[j]; a.bias[j] = p.bias[j]; a.y[j] = p.y[j];
    a.col0[j + 1] = a.col0[j] + p.n[j];
  }
  const int total = a.col0[p.n_blocks];
  if (p.rows <= 8) LFDM_LAUNCH((multi_linear_fwd_kernel<8>), dim3((unsigned)((total + 15) / 16)), dim3(256), 0, stream, a);
  else LFDM_LAUNCH((multi_linear_fwd_kernel<16>), dim3((unsigned)((total + 15) / 16)), dim3(256), 0, stream, a);
  return lfdm_check_launch("multi_linear");
}

extern "C" size_t lfdm_multi_linear_bwd_ws_bytes(const lfdm_multi_linear_params* p) {
  if (!p || !p->dx) return 0;
  int64_t total = 0;
  for (int j = 0; j < p->n_blocks && j < ML_MAX; ++j) total += p->n[j];
  return (size_t)((total + 63) / 64) * p->rows * p->k * sizeof(float);
}

extern "C" int lfdm_multi_linear_bwd_f32(const lfdm_multi_linear_params* pp, void* ws, size_t ws_bytes, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!pp) { lfdm_set_error("multi_linear_bwd: null params"); return LFDM_EINVAL; }
  const lfdm_multi_linear_params& p = *pp;
  int rc = ml_check(p, "multi_linear_bwd");
  if (rc) return rc;
  MlBwd a;
  a.x = p.x; a.nblk = p.n_blocks; a.rows = p.rows; a.k = p.k; a.act = p.act;
  a.col0[0] = 0;
  bool any_dw = false;
  for (int j = 0; j < p.n_blocks; ++j) {
    if (p.dw[j] && (((uintptr_t)p.dw[j]) & 15)) { lfdm_set_error("multi_linear_bwd: a dw tensor is not 16-byte aligned"); return LFDM_EINVAL; }
    a.w[j] = p.w[j]; a.dy[j] = p.dy[j]; a.dw[j] = p.dw[j]; a.dbias[j] = p.dbias[j];
    a.col0[j + 1] = a.col0[j] + p.n[j];
    any_dw = any_dw || p.dw[j] || p.dbias[j];
  }
  const int total = a.col0[p.n_blocks];
  if (any_dw) {
    if (p.rows <= 8) LFDM_LAUNCH((multi_linear_dw_kernel<8>), dim3((unsigned)((total + 15) / 16)), dim3(256), 0, stream, a);
    else LFDM_LAUNCH((multi_linear_dw_kernel<16>), dim3((unsigned)((total + 15) / 16)), dim3(256), 0, stream, a);
    rc = lfdm_check_launch("multi_linear_dw");
    if (rc) return rc;
  }
  if (p.dx) {
    const size_t need = lfdm_multi_linear_bwd_ws_bytes(pp);
    if (!ws || ws_bytes < need || (((uintptr_t)ws) & 15) || (((uintptr_t)p.dx) & 15)) {
      lfdm_set_error("multi_linear_bwd: workspace too small or unaligned (lfdm_multi_linear_bwd_ws_bytes)");
      return LFDM_EWORKSPACE;
    }
    const int nparts = (total + 63) / 64;
    if (p.rows <= 8) LFDM_LAUNCH((multi_linear_dx_partial_kernel<8>), dim3((unsigned)nparts), dim3(256), 0, stream, a, (float*)ws);
    else LFDM_LAUNCH((multi_linear_dx_partial_kernel<16>), dim3((unsigned)nparts), dim3(256), 0, stream, a, (float*)ws);
    rc = lfdm_check_launch("multi_linear_dx_partial");
    if (rc) return rc;
    const int n4 = p.rows * p.k / 4;
    LFDM_LAUNCH(multi_linear_dx_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, (const float*)ws, nparts, p.x, p.dx, n4,
                p.act);
    rc = lfdm_check_launch("multi_linear_dx_reduce");
  }
  return rc;
}
