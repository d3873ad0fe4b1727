// Backward kernels of the convolution family for the DM training step
// (include/lfdm_hip.h: lfdm_conv2d_wgrad_cl_f32, lfdm_colsum_f32, lfdm_sum_leading_f32).
//
// What autograd does for `loss.backward()` through every Conv3d/Linear of Unet3D
// (DM/modules/video_flow_diffusion.py:199,224,246-247,300-301,158,167,410 via
// DM/modules/video_flow_diffusion_model.py:181-188) splits into
//   data gradient   = a convolution with the transposed / flipped filter -> the forward kernels
//                     (lfdm_conv2d_cl_f32) with re-packed weights, no new code;
//   weight gradient = dW[tap][ci][co] = sum_r X[r shifted by tap][ci] * dY[r][co]  -> this file;
//   bias gradient   = column sums of dY                                             -> this file.
//
// Weight gradient as an fp32-MFMA GEMM whose reduction axis is the PIXEL axis: with channels-last
// rows both operands are read exactly as they lie in memory (row = pixel, contiguous channels), so a
// chunk of 32 pixel rows is staged row-major in LDS (16-byte coalesced loads, no transpose) and
// v_mfma_f32_32x32x2 consumes two pixel rows per instruction: lane l holds A[ci = l&31][r = l>>5]
// and B[r = l>>5][co = l&31].  One workgroup owns one filter tap x (64*WM input channels) x
// (64*WN output channels) and a slice of the rows (grid.z); slices land in `partial` and are summed in
// a fixed order by lfdm_sum_leading_f32 (deterministic - no float atomics).
#include <stdlib.h>

#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

template <int WM, int WN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(lfdm_wgrad_params p, int splits, float* dst_base, float* bias_partial) {
  constexpr int TCI = 64 * WM, TCO = 64 * WN, BR = 32;
  constexpr int A_F4 = BR * TCI / 4 / 256, B_F4 = BR * TCO / 4 / 256;
  constexpr int STAGE = BR * (TCI + TCO);
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kk = lane >> 5;
  const int ci_tiles = (p.cin + TCI - 1) / TCI;
  const int tap = blockIdx.x / ci_tiles;
  const int ci0 = (blockIdx.x - tap * ci_tiles) * TCI;
  const int co0 = blockIdx.y * TCO;
  const int ky = tap / p.kw, kx = tap - ky * p.kw;
  const int hqwq = p.hq * p.wq;
  const int M = p.n_img * hqwq;                     // < 2^31 (host check)
  const int nchunks = (M + BR - 1) / BR;
  const int c_begin = (int)((int64_t)nchunks * blockIdx.z / splits);
  const int c_end = (int)((int64_t)nchunks * (blockIdx.z + 1) / splits);
  const int nk = c_end - c_begin;

  const int64_t in_rows = (int64_t)p.n_img * p.hi * p.wi;
  const lfdm_buf bufx = lfdm_make_buf(p.x, (uint32_t)(((in_rows - 1) * p.ldx + p.cin) * 4));
  const lfdm_buf bufy = lfdm_make_buf(p.dy, (uint32_t)((((int64_t)M - 1) * p.lddy + p.cout) * 4));

  // Bias gradient from the same pass (bias_partial != NULL): the workgroups of tap 0 / input-channel tile 0 see every dy row of their
  // slice exactly once on its way into LDS - each thread's float4 column position is the same for all its loads (256 % (TCO/4) == 0).
  const bool do_bias = bias_partial != nullptr && blockIdx.x == 0;
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  // Two register sets: the global loads of chunk c + 2 and c + 3 are in flight while chunk c is multiplied (one set - loads issued one chunk =
  // ~1.7 us of MFMA work ahead of their LDS store - left the kernel waiting for HBM at 0.39-0.53 of the MFMA peak, profiles/r05_c_bench_wgrad.txt).
  // Chunks past the slice's end are fetched as zeros (rows >= row_end take the out-of-range offset), so the loop needs no tail handling
  // and runs an even number of trips.
  const int row_end = c_end * BR < M ? c_end * BR : M;
  float4 ra0[A_F4], rb0[B_F4], ra1[A_F4], rb1[B_F4];
  auto fetch = [&](int chunk, float4 (&ra)[A_F4], float4 (&rb)[B_F4]) {
    const int r0 = chunk * BR;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int f = tid + 256 * i;
      const int row = f / (TCI / 4), c4 = f - row * (TCI / 4);
      const int r = r0 + row;
      uint32_t off = LFDM_BUF_OOB;
      if (r < row_end && ci0 + 4 * c4 < p.cin) {
        const int img = r / hqwq;
        const int rem = r - img * hqwq;
        const int qy = rem / p.wq, qx = rem - qy * p.wq;
        const int iy = qy * p.stride + ky - p.pad_y, ix = qx * p.stride + kx - p.pad_x;
        if (iy >= 0 && iy < p.hi && ix >= 0 && ix < p.wi)
          off = (uint32_t)((((int64_t)(img * p.hi + iy) * p.wi + ix) * p.ldx + ci0 + 4 * c4) * 4);
      }
      ra[i] = lfdm_buf_load_f4(bufx, off);
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      const int f = tid + 256 * i;
      const int row = f / (TCO / 4), c4 = f - row * (TCO / 4);
      const int r = r0 + row;
      const uint32_t off = (r < row_end && co0 + 4 * c4 < p.cout) ? (uint32_t)(((int64_t)r * p.lddy + co0 + 4 * c4) * 4)
                                                                 : LFDM_BUF_OOB;
      rb[i] = lfdm_buf_load_f4(bufy, off);
    }
  };
  auto stage = [&](int buf, const float4 (&ra)[A_F4], const float4 (&rb)[B_F4]) {
    float* const As = smem + buf * STAGE;
    float* const Bs = As + BR * TCI;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) *reinterpret_cast<float4*>(As + 4 * (tid + 256 * i)) = ra[i];
#pragma unroll
    for (int i = 0; i < B_F4; ++i) *reinterpret_cast<float4*>(Bs + 4 * (tid + 256 * i)) = rb[i];
    if (do_bias) {            // (every chunk is staged exactly once; chunks past the slice are zeros)
#pragma unroll
      for (int i = 0; i < B_F4; ++i) { bsum.x += rb[i].x; bsum.y += rb[i].y; bsum.z += rb[i].z; bsum.w += rb[i].w; }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto multiply = [&](int cur) {
    const float* const As = smem + cur * STAGE + wm * (32 * WM) + l31;
    const float* const Bs = smem + cur * STAGE + BR * TCI + wn * (32 * WN) + l31;
#pragma unroll
    for (int s = 0; s < BR / 2; ++s) {
      const int r = 2 * s + kk;
      float a[WM], b[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) a[i] = As[r * TCI + 32 * i];
#pragma unroll
      for (int j = 0; j < WN; ++j) b[j] = Bs[r * TCO + 32 * j];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = mfma_32x32x2(a[i], b[j], acc[i][j]);
    }
  };

  if (nk > 0) {
    fetch(c_begin, ra0, rb0);
    stage(0, ra0, rb0);
    __syncthreads();
    fetch(c_begin + 1, ra0, rb0);
    fetch(c_begin + 2, ra1, rb1);
    const int nk2 = (nk + 1) & ~1;
    for (int c = 0; c < nk2; c += 2) {
      stage(1, ra0, rb0);                       // chunk c + 1 -> buffer 1
      fetch(c_begin + c + 3, ra0, rb0);
      multiply(0);                              // chunk c
      __syncthreads();
      stage(0, ra1, rb1);                       // chunk c + 2 -> buffer 0
      fetch(c_begin + c + 4, ra1, rb1);
      multiply(1);                              // chunk c + 1 (zeros when nk is odd and this is the last trip)
      __syncthreads();
    }
  }

  float* const dst = dst_base + (int64_t)blockIdx.z * ((int64_t)p.kh * p.kw * p.cin * p.cout);
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int co = co0 + wn * (32 * WN) + 32 * j + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + wm * (32 * WM) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (ci < p.cin && co < p.cout) dst[((int64_t)tap * p.cin + ci) * p.cout + co] = acc[i][j][r];
      }
    }
  if (do_bias) {      // (uniform per workgroup; the K loop ended with a barrier, the staging buffers are free)
    constexpr int C4 = TCO / 4, RL = 256 / C4;
    *reinterpret_cast<float4*>(smem + (tid / C4) * TCO + 4 * (tid % C4)) = bsum;
    __syncthreads();
    if (tid < TCO && co0 + tid < p.cout) {
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < RL; ++k) sum += smem[k * TCO + tid];
      bias_partial[(int64_t)blockIdx.z * p.cout + co0 + tid] = sum;
    }
  }
}

// 3x3 / stride 1 / pad 1 (every ResnetBlock, ResBlock2d, hourglass and VGG convolution): ONE workgroup owns ALL NINE taps of a 64 ci x 64 co
// block.  The per-tap kernel above streams X and dY once per tap and (ci, co) tile pair - 1.5 GB for the 64 -> 64 convolution of a B = 8
// training step, 3.8-4.7 TB/s of re-reads at 0.39-0.53 of the matrix-pipe peak (profiles/r05_c_bench_wgrad.txt).  Here a chunk is a TW x 4
// pixel tile (x 2 images at TW = 4): its dY rows (32) and its X window with a one-pixel halo ((TW+2) x 6 rows per image, out-of-image pixels
// = out-of-range offsets = zeros, so the taps need no masks) are staged once, and each k-step feeds nine MFMAs - one per tap - from window rows at
// compile-time offsets.  Nine accumulator tiles per wave (144 registers), 2 x 26.6 KB of LDS, two workgroups per CU.
template <int TW>
__global__ __launch_bounds__(256) void conv_wgrad3_kernel(lfdm_wgrad_params p, int splits, float* dst_base, float* bias_partial) {
  constexpr int TH = 4, TI = 32 / (TW * TH), WW = TW + 2, WH = TH + 2, WROWS = TI * WW * WH;      // window rows: 60 (TW = 8) / 72 (TW = 4)
  constexpr int XF4 = (WROWS * 16 + 255) / 256;
  constexpr int STAGE = WROWS * 64 + 32 * 64;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kk = lane >> 5;
  const int ci_tiles = (p.cin + 63) / 64;
  const int ci_tile = blockIdx.x % ci_tiles;
  const int ci0 = ci_tile * 64, co0 = (blockIdx.x / ci_tiles) * 64;
  const int tiles_x = p.wi / TW, per_img = tiles_x * (p.hi / TH);
  const int n_tiles = (p.n_img / TI) * per_img;
  const int t_begin = (int)((int64_t)n_tiles * blockIdx.z / splits);
  const int t_end = (int)((int64_t)n_tiles * (blockIdx.z + 1) / splits);
  const int nk = t_end - t_begin;

  const int64_t rows = (int64_t)p.n_img * p.hi * p.wi;
  const lfdm_buf bufx = lfdm_make_buf(p.x, (uint32_t)(((rows - 1) * p.ldx + p.cin) * 4));
  const lfdm_buf bufy = lfdm_make_buf(p.dy, (uint32_t)(((rows - 1) * p.lddy + p.cout) * 4));
  const bool do_bias = bias_partial != nullptr && ci_tile == 0;
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 rx[XF4], ry[2];
  const int c4 = tid & 15;                                   // this thread's float4 column in both operands
  auto fetch = [&](int tile) {
    const bool live = tile < t_end;
    const int g = tile / per_img, rem = tile - g * per_img;
    const int tyi = rem / tiles_x;
    const int ty0 = tyi * TH, tx0 = (rem - tyi * tiles_x) * TW, img0 = g * TI;
#pragma unroll
    for (int i = 0; i < XF4; ++i) {
      const int w = (tid >> 4) + 16 * i;                     // window row
      uint32_t off = LFDM_BUF_OOB;
      if (live && w < WROWS && ci0 + 4 * c4 < p.cin) {
        const int ti = w / (WW * WH), wr = w - ti * (WW * WH);
        const int wy = wr / WW, wx = wr - wy * WW;
        const int iy = ty0 + wy - 1, ix = tx0 + wx - 1;
        if (iy >= 0 && iy < p.hi && ix >= 0 && ix < p.wi)
          off = (uint32_t)((((int64_t)((img0 + ti) * p.hi + iy) * p.wi + ix) * p.ldx + ci0 + 4 * c4) * 4);
      }
      rx[i] = lfdm_buf_load_f4(bufx, off);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (tid >> 4) + 16 * i;                     // pixel of the tile
      const int ti = r / (TW * TH), pr = r - ti * (TW * TH);
      const int py = pr / TW, px = pr - py * TW;
      const uint32_t off = (live && co0 + 4 * c4 < p.cout)
                               ? (uint32_t)((((int64_t)((img0 + ti) * p.hi + ty0 + py) * p.wi + tx0 + px) * p.lddy + co0 + 4 * c4) * 4)
                               : LFDM_BUF_OOB;
      ry[i] = lfdm_buf_load_f4(bufy, off);
    }
  };
  auto stage = [&](int buf) {
    float* const Xs = smem + buf * STAGE;
    float* const Ys = Xs + WROWS * 64;
#pragma unroll
    for (int i = 0; i < XF4; ++i)
      if ((tid >> 4) + 16 * i < WROWS) *reinterpret_cast<float4*>(Xs + 4 * (tid + 256 * i)) = rx[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(Ys + 4 * (tid + 256 * i)) = ry[i];
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { bsum.x += ry[i].x; bsum.y += ry[i].y; bsum.z += ry[i].z; bsum.w += ry[i].w; }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  auto multiply = [&](int buf) {
    const float* const Xb = smem + buf * STAGE + kk * 64 + wm * 32 + l31;      // k-slot kk = the second pixel of the pair: one window row on
    const float* const Yb = smem + buf * STAGE + WROWS * 64 + kk * 64 + wn * 32 + l31;
    lfdm_static_for<0, 16>([&](auto S) {
      constexpr int s = decltype(S)::value;
      constexpr int ti = (2 * s) / (TW * TH), pr = (2 * s) % (TW * TH), py = pr / TW, px = pr % TW;
      const float b = Yb[(2 * s) * 64];
      lfdm_static_for<0, 9>([&](auto T) {
        constexpr int tap = decltype(T)::value;
        constexpr int wrow = ti * (WW * WH) + (py + tap / 3) * WW + px + tap % 3;
        acc[tap] = mfma_32x32x2(Xb[wrow * 64], b, acc[tap]);
      });
    });
  };

  if (nk > 0) {
    fetch(t_begin);
    stage(0);
    __syncthreads();
    fetch(t_begin + 1);
    for (int c = 0; c < nk; ++c) {
      const int cur = c & 1;
      stage(cur ^ 1);                           // tile c + 1 (zeros past the slice)
      fetch(t_begin + c + 2);
      multiply(cur);
      __syncthreads();
    }
  }

  float* const dst = dst_base + (int64_t)blockIdx.z * ((int64_t)9 * p.cin * p.cout);
  const int co = co0 + wn * 32 + l31;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
      if (ci < p.cin && co < p.cout) dst[((int64_t)t * p.cin + ci) * p.cout + co] = acc[t][r];
    }
  if (do_bias) {      // (uniform per workgroup; the loop ended with a barrier, the staging buffers are free)
    *reinterpret_cast<float4*>(smem + (tid >> 4) * 64 + 4 * c4) = bsum;
    __syncthreads();
    if (tid < 64 && co0 + tid < p.cout) {
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) sum += smem[k * 64 + tid];
      bias_partial[(int64_t)blockIdx.z * p.cout + co0 + tid] = sum;
    }
  }
}

// Second stage of a split weight gradient in the REFERENCE layout (lfdm_wgrad_params.dw_layout = 1):
//   out[(co * cin_total + ci_off + ci) * taps + tap] = sum_s slab[s][tap][ci][co]   (s in fixed order, like sum_leading4_kernel)
// i.e. the (cout, cin, [1,] kh, kw) tensor autograd wants for Conv3d.weight (video_flow_diffusion.py:199,224,...) without the
// permute-copy, written straight into the optimizer's flat gradient slot.  Workgroup = CI_T input channels x 64 output channels x all
// taps: float4 loads along co (one item = one float4 of one (ci, tap), all slabs), a transposition through LDS, then per output
// channel one contiguous run of CI_T * taps floats.  The trailing workgroups sum the bias partials (256 columns each).
template <int CI_T>
__global__ __launch_bounds__(256) void wgrad_reduce_ref_kernel(const float* __restrict__ slabs, int s, int64_t slab_stride, int taps, int cin,
                                                               int cout, float* __restrict__ out, int cin_total, int ci_off,
                                                               const float* __restrict__ bias_partial, float* __restrict__ dbias,
                                                               int n_dw_blocks) {
  constexpr int TMAX = CI_T == 16 ? 1 : 16;
  __shared__ float T[64 * CI_T * TMAX];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= n_dw_blocks) {
    const int co = ((int)blockIdx.x - n_dw_blocks) * 256 + tid;
    if (co < cout) {
      float sum = 0.f;
      for (int k = 0; k < s; ++k) sum += bias_partial[(int64_t)k * cout + co];
      dbias[co] = sum;
    }
    return;
  }
  const int co_tiles = (cout + 63) / 64;
  const int ci0 = ((int)blockIdx.x / co_tiles) * CI_T, co0 = ((int)blockIdx.x % co_tiles) * 64;
  const int row = CI_T * taps;
  for (int it = tid; it < row * 16; it += 256) {
    const int c4 = it & 15, rest = it >> 4;
    const int ci_l = rest / taps, tap = rest - ci_l * taps;
    const int ci = ci0 + ci_l, co = co0 + 4 * c4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ci < cin && co < cout) {
      const float* src = slabs + ((int64_t)tap * cin + ci) * cout + co;
#pragma unroll 8
      for (int k = 0; k < s; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)k * slab_stride);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    float* t = T + (4 * c4) * row + ci_l * taps + tap;
    t[0] = acc.x; t[row] = acc.y; t[2 * row] = acc.z; t[3 * row] = acc.w;
  }
  __syncthreads();
  for (int idx = tid; idx < 64 * row; idx += 256) {
    const int co_l = idx / row, j = idx - co_l * row;
    if (co0 + co_l < cout && ci0 + j / taps < cin) out[((int64_t)(co0 + co_l) * cin_total + ci_off + ci0) * taps + j] = T[idx];
  }
}

// out[i] = sum_s in[s*n + i]  (fixed order)
__global__ __launch_bounds__(256) void sum_leading_kernel(const float* in, float* out, int64_t n, int s) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float acc = 0.f;
    for (int k = 0; k < s; ++k) acc += in[(int64_t)k * n + i];
    out[i] = acc;
  }
}

// same, 16 bytes per lane (n % 4 == 0, 16-byte aligned): the weight-gradient slabs are tens of MB
__global__ __launch_bounds__(256) void sum_leading4_kernel(const float4* in, float4* out, int64_t n4, int s) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int k = 0; k < s; ++k) {
      const float4 v = in[(int64_t)k * n4 + i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    out[i] = acc;
  }
}

// float4 variant: 16 float4 columns (64 channels) x 16 row lanes per workgroup
__global__ __launch_bounds__(256) void colsum4_kernel(const float* x, int64_t rows, int c, int ld, float* partial,
                                                      int rows_per_blk) {
  __shared__ __attribute__((aligned(16))) float red[16][64];
  const int c4 = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int col = blockIdx.x * 64 + 4 * c4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_blk;
  int64_t r1 = r0 + rows_per_blk;
  if (r1 > rows) r1 = rows;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < c)
    for (int64_t r = r0 + rl; r < r1; r += 16) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * ld + col);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  *reinterpret_cast<float4*>(&red[rl][4 * c4]) = acc;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cc = blockIdx.x * 64 + threadIdx.x;
    if (cc < c) {
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) sum += red[k][threadIdx.x];
      partial[(int64_t)blockIdx.y * c + cc] = sum;
    }
  }
}

// partial[blk][c] = sum over the block's rows of x[r][c]; 64 columns per workgroup column strip
__global__ __launch_bounds__(256) void colsum_kernel(const float* x, int64_t rows, int c, int ld, float* partial,
                                                     int rows_per_blk) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_blk;
  int64_t r1 = r0 + rows_per_blk;
  if (r1 > rows) r1 = rows;
  float acc = 0.f;
  if (col < c)
    for (int64_t r = r0 + rl; r < r1; r += 4) acc += x[r * ld + col];
  red[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && col < c)
    partial[(int64_t)blockIdx.y * c + col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Upsample (nearest x2) + pad by P (zeros or reflect) materialised for the training path of the
// `use_deconv=False` Upsample (video_flow_diffusion.py:160-163): forward gathers, backward gathers the
// adjoint (every input pixel sums the <= 3x3 output positions that read it).  P in {0, 1}.
__device__ __forceinline__ int up_src(int p, int pad, int n2, int reflect) {   // padded coord -> upsampled coord or -1
  int u = p - pad;
  if (u < 0) u = reflect ? -u : -1;
  else if (u >= n2) u = reflect ? 2 * n2 - 2 - u : -1;
  return u;
}

__global__ __launch_bounds__(256) void upsample_pad_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int n_img,
                                                               int h, int w, int c, int pad, int reflect) {
  const int c4n = c >> 2, ho = 2 * h + 2 * pad, wo = 2 * w + 2 * pad;
  const int64_t total = (int64_t)n_img * ho * wo * c4n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % c4n);
    int64_t r = i / c4n;
    const int px = (int)(r % wo);
    r /= wo;
    const int py = (int)(r % ho);
    const int n = (int)(r / ho);
    const int uy = up_src(py, pad, 2 * h, reflect), ux = up_src(px, pad, 2 * w, reflect);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (uy >= 0 && ux >= 0) v = *reinterpret_cast<const float4*>(x + (((int64_t)n * h + (uy >> 1)) * w + (ux >> 1)) * c + 4 * c4);
    *reinterpret_cast<float4*>(out + (((int64_t)n * ho + py) * wo + px) * c + 4 * c4) = v;
  }
}

__global__ __launch_bounds__(256) void upsample_pad_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int n_img,
                                                               int h, int w, int c, int pad, int reflect) {
  const int c4n = c >> 2, ho = 2 * h + 2 * pad, wo = 2 * w + 2 * pad;
  const int64_t total = (int64_t)n_img * h * w * c4n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % c4n);
    int64_t r = i / c4n;
    const int j = (int)(r % w);
    r /= w;
    const int ii = (int)(r % h);
    const int n = (int)(r / h);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // candidate padded rows / cols: the two direct ones and the reflected border ones
    for (int py = 2 * ii + pad - 2; py <= 2 * ii + pad + 3; ++py) {
      if (py < 0 || py >= ho) continue;
      const int uy = up_src(py, pad, 2 * h, reflect);
      if (uy < 0 || (uy >> 1) != ii) continue;
      for (int px = 2 * j + pad - 2; px <= 2 * j + pad + 3; ++px) {
        if (px < 0 || px >= wo) continue;
        const int ux = up_src(px, pad, 2 * w, reflect);
        if (ux < 0 || (ux >> 1) != j) continue;
        const float4 v = *reinterpret_cast<const float4*>(dy + (((int64_t)n * ho + py) * wo + px) * c + 4 * c4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    *reinterpret_cast<float4*>(dx + (((int64_t)n * h + ii) * w + j) * c + 4 * c4) = acc;
  }
}

struct WgradPlan {
  int wm, wn, splits;
  int tile_w;       // 0: one workgroup per tap (conv_wgrad_kernel); 8 / 4: all nine taps of a TW x 4 pixel tile (conv_wgrad3_kernel)
};

WgradPlan wgrad_plan(const lfdm_wgrad_params& p) {
  WgradPlan pl;
  pl.tile_w = 0;
  {
    const char* e = getenv("LFDM_WGRAD3");        // experiment knob (tools/bench_wgrad.py): 0 = the per-tap kernel everywhere
    const bool on = !(e && e[0] == '0');
    if (on && p.kh == 3 && p.kw == 3 && p.stride == 1 && p.pad_y == 1 && p.pad_x == 1 && p.hq == p.hi && p.wq == p.wi && p.hi % 4 == 0 &&
        p.cin >= 32 && p.cout >= 32) {
      const int tw = p.wi % 8 == 0 ? 8 : ((p.wi % 4 == 0 && p.n_img % 2 == 0) ? 4 : 0);
      if (tw) {
        const int64_t n_tiles = (int64_t)(p.n_img / (32 / (tw * 4))) * (p.hi / 4) * (p.wi / tw);
        const int64_t blocks = (int64_t)((p.cin + 63) / 64) * ((p.cout + 63) / 64);
        if (n_tiles >= 32) {
          // two workgroups per CU and no more: every further split is another 9 * cin * cout slab to write and re-read (sweep in
          // profiles/r05_d_bench_wgrad3.txt: 1024 / 2048 workgroups lose 10-25 % on the 128 ... 512-channel shapes)
          const char* ew = lfdm_knob("LFDM_WGRAD3_WGS");      // experiment knob: workgroups aimed at
          const int64_t want = ew ? atol(ew) : 512;
          const char* ec = lfdm_knob("LFDM_WGRAD3_MAXSPLIT");
          const int64_t cap = ec ? atol(ec) : 256;
          int64_t s = (want + blocks - 1) / blocks;
          if (s > n_tiles / 8) s = n_tiles / 8;
          if (s > cap) s = cap;
          if (s < 1) s = 1;
          pl.tile_w = tw;
          pl.wm = pl.wn = 1;
          pl.splits = (int)s;
          return pl;
        }
      }
    }
  }
  pl.wm = p.cin >= 96 ? 2 : 1;
  pl.wn = p.cout >= 96 ? 2 : 1;
  const int64_t M = (int64_t)p.n_img * p.hq * p.wq;
  const int64_t nchunks = (M + 31) / 32;
  const int64_t tiles = (int64_t)p.kh * p.kw * ((p.cin + 64 * pl.wm - 1) / (64 * pl.wm)) * ((p.cout + 64 * pl.wn - 1) / (64 * pl.wn));
  int64_t s = (1024 + tiles - 1) / tiles;
  if (s > nchunks / 8) s = nchunks / 8;
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  pl.splits = (int)s;
  return pl;
}

}  // namespace

// Workspace = [splits slabs of kh*kw*cin*cout floats, when the result does not come straight out of the GEMM kernel: splits > 1 or
// dw_layout = 1] followed by [splits rows of cout bias partials, when dbias is wanted].
static bool wgrad_needs_slabs(const lfdm_wgrad_params& p, const WgradPlan& pl) { return pl.splits > 1 || p.dw_layout == 1; }

extern "C" size_t lfdm_conv2d_wgrad_ws_bytes(const lfdm_wgrad_params* p) {
  if (!p) return 0;
  const WgradPlan pl = wgrad_plan(*p);
  size_t floats = 0;
  if (wgrad_needs_slabs(*p, pl)) floats += (size_t)pl.splits * p->kh * p->kw * p->cin * p->cout;
  if (p->dbias) floats += (size_t)pl.splits * p->cout;
  return floats * sizeof(float);
}

extern "C" int lfdm_sum_leading_f32(const float* in, float* out, int64_t n, int s, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!in || !out || n <= 0 || s <= 0) { lfdm_set_error("sum_leading: bad arguments"); return LFDM_EINVAL; }
  if (n % 4 == 0 && ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0) {
    int64_t nb = (n / 4 + 255) / 256;
    if (nb > 65536) nb = 65536;
    LFDM_LAUNCH(sum_leading4_kernel, dim3((unsigned)nb), dim3(256), 0, stream, reinterpret_cast<const float4*>(in),
                reinterpret_cast<float4*>(out), n / 4, s);
    return lfdm_check_launch("sum_leading");
  }
  int64_t nb = (n + 255) / 256;
  if (nb > 65536) nb = 65536;
  LFDM_LAUNCH(sum_leading_kernel, dim3((unsigned)nb), dim3(256), 0, stream, in, out, n, s);
  return lfdm_check_launch("sum_leading");
}

extern "C" int lfdm_conv2d_wgrad_cl_f32(const lfdm_wgrad_params* pp, void* ws, size_t ws_bytes, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!pp) { lfdm_set_error("wgrad: null params"); return LFDM_EINVAL; }
  const lfdm_wgrad_params p = *pp;
  const int64_t M = (int64_t)p.n_img * p.hq * p.wq;
  const int64_t in_rows = (int64_t)p.n_img * p.hi * p.wi;
  if (!p.x || !p.dy || !p.dw || p.cin <= 0 || p.cout <= 0 || p.n_img <= 0 || p.hi <= 0 || p.wi <= 0 || p.hq <= 0 ||
      p.wq <= 0 || p.kh <= 0 || p.kw <= 0 || p.stride <= 0 || p.ldx < p.cin || p.lddy < p.cout || M >= (1ll << 31) ||
      in_rows >= (1ll << 31)) {
    lfdm_set_error("wgrad: invalid geometry");
    return LFDM_EINVAL;
  }
  if (p.cin % 4 || p.cout % 4 || p.ldx % 4 || p.lddy % 4 || (((uintptr_t)p.x) & 15) || (((uintptr_t)p.dy) & 15) ||
      in_rows * p.ldx * 4 >= (1ll << 32) - 64 || M * p.lddy * 4 >= (1ll << 32) - 64) {
    lfdm_set_error("wgrad: channels and row strides must be multiples of 4, buffers 16-byte aligned and < 4 GiB");
    return LFDM_EINVAL;
  }
  const int taps = p.kh * p.kw;
  if (p.dw_layout != 0 && p.dw_layout != 1) { lfdm_set_error("wgrad: dw_layout must be 0 (tap-major) or 1 (reference layout)"); return LFDM_EINVAL; }
  if (p.dw_layout == 1 && (taps > 16 || p.dw_cin_total < p.dw_ci_off + p.cin || p.dw_ci_off < 0)) {
    lfdm_set_error("wgrad: the reference layout covers filters of up to 16 taps; dw_ci_off + cin must fit dw_cin_total");
    return LFDM_EINVAL;
  }
  const WgradPlan pl = wgrad_plan(p);
  const size_t need = lfdm_conv2d_wgrad_ws_bytes(&p);
  if (need > 0 && (!ws || ws_bytes < need || (((uintptr_t)ws) & 15))) {
    lfdm_set_error("wgrad: workspace too small or not 16-byte aligned (lfdm_conv2d_wgrad_ws_bytes)");
    return LFDM_EWORKSPACE;
  }
  const bool slabs = wgrad_needs_slabs(p, pl);
  const int64_t n_dw = (int64_t)taps * p.cin * p.cout;
  float* dst = slabs ? (float*)ws : p.dw;
  float* bias_partial = p.dbias ? (float*)ws + (slabs ? (int64_t)pl.splits * n_dw : 0) : nullptr;
  const dim3 grid((unsigned)(taps * ((p.cin + 64 * pl.wm - 1) / (64 * pl.wm))), (unsigned)((p.cout + 64 * pl.wn - 1) / (64 * pl.wn)),
                  (unsigned)pl.splits);
  if (pl.tile_w) {
    const dim3 grid3((unsigned)(((p.cin + 63) / 64) * ((p.cout + 63) / 64)), 1, (unsigned)pl.splits);
    if (pl.tile_w == 8) LFDM_LAUNCH((conv_wgrad3_kernel<8>), grid3, dim3(256), 0, stream, p, pl.splits, dst, bias_partial);
    else LFDM_LAUNCH((conv_wgrad3_kernel<4>), grid3, dim3(256), 0, stream, p, pl.splits, dst, bias_partial);
  } else if (pl.wm == 2 && pl.wn == 2) LFDM_LAUNCH((conv_wgrad_kernel<2, 2>), grid, dim3(256), 0, stream, p, pl.splits, dst, bias_partial);
  else if (pl.wm == 2) LFDM_LAUNCH((conv_wgrad_kernel<2, 1>), grid, dim3(256), 0, stream, p, pl.splits, dst, bias_partial);
  else if (pl.wn == 2) LFDM_LAUNCH((conv_wgrad_kernel<1, 2>), grid, dim3(256), 0, stream, p, pl.splits, dst, bias_partial);
  else LFDM_LAUNCH((conv_wgrad_kernel<1, 1>), grid, dim3(256), 0, stream, p, pl.splits, dst, bias_partial);
  int rc = lfdm_check_launch("conv_wgrad");
  if (rc) return rc;
  if (p.dw_layout == 1) {
    // input channels per workgroup: 16 for 1x1 filters (64-byte runs), else 4 - or 1 while that leaves fewer than 256 workgroups
    const int co_tiles = (p.cout + 63) / 64;
    int ci_t = taps == 1 ? 16 : 4;
    if (ci_t == 4 && (int64_t)((p.cin + 3) / 4) * co_tiles < 256) ci_t = 1;
    const int n_dw_blocks = ((p.cin + ci_t - 1) / ci_t) * co_tiles;
    const int n_bias_blocks = p.dbias ? (p.cout + 255) / 256 : 0;
    const dim3 rgrid((unsigned)(n_dw_blocks + n_bias_blocks));
#define LFDM_WGRAD_REDUCE(CT)                                                                                                        \
    LFDM_LAUNCH((wgrad_reduce_ref_kernel<CT>), rgrid, dim3(256), 0, stream, (const float*)ws, pl.splits, n_dw, taps, p.cin, p.cout, p.dw, \
                p.dw_cin_total, p.dw_ci_off, (const float*)bias_partial, p.dbias, n_dw_blocks)
    if (ci_t == 16) LFDM_WGRAD_REDUCE(16);
    else if (ci_t == 4) LFDM_WGRAD_REDUCE(4);
    else LFDM_WGRAD_REDUCE(1);
#undef LFDM_WGRAD_REDUCE
    return lfdm_check_launch("wgrad_reduce_ref");
  }
  if (pl.splits > 1) {
    rc = lfdm_sum_leading_f32((const float*)ws, p.dw, n_dw, pl.splits, stream_);
    if (rc) return rc;
  }
  if (p.dbias) return lfdm_sum_leading_f32(bias_partial, p.dbias, p.cout, pl.splits, stream_);
  return LFDM_OK;
}

extern "C" size_t lfdm_colsum_ws_bytes(int64_t rows, int c) {
  int64_t nblk = (rows + 1023) / 1024;
  if (nblk > 512) nblk = 512;
  return (size_t)nblk * c * sizeof(float);
}

extern "C" int lfdm_colsum_f32(const float* x, int64_t rows, int c, int ld, float* out, void* ws, size_t ws_bytes,
                               lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !out || rows <= 0 || c <= 0 || ld < c) { lfdm_set_error("colsum: bad arguments"); return LFDM_EINVAL; }
  int64_t nblk = (rows + 1023) / 1024;
  if (nblk > 512) nblk = 512;
  if (!ws || ws_bytes < (size_t)nblk * c * sizeof(float)) { lfdm_set_error("colsum: workspace too small"); return LFDM_EWORKSPACE; }
  const int rows_per_blk = (int)((rows + nblk - 1) / nblk);
  if (c % 4 == 0 && ld % 4 == 0 && (((uintptr_t)x) & 15) == 0)
    LFDM_LAUNCH(colsum4_kernel, dim3((unsigned)((c + 63) / 64), (unsigned)nblk), dim3(256), 0, stream, x, rows, c, ld, (float*)ws,
                rows_per_blk);
  else
    LFDM_LAUNCH(colsum_kernel, dim3((unsigned)((c + 63) / 64), (unsigned)nblk), dim3(256), 0, stream, x, rows, c, ld, (float*)ws,
                rows_per_blk);
  int rc = lfdm_check_launch("colsum");
  if (rc) return rc;
  return lfdm_sum_leading_f32((const float*)ws, out, c, (int)nblk, stream_);
}

extern "C" int lfdm_upsample2_pad_cl_f32(const float* x, float* out, int n_img, int h, int w, int channels, int pad,
                                         int reflect, int backward, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !out || n_img <= 0 || h <= 0 || w <= 0 || channels <= 0 || channels % 4 != 0 || pad < 0 || pad > 1 ||
      (reflect && pad == 1 && (h < 1 || w < 1))) {
    lfdm_set_error("upsample2_pad: bad arguments (C % 4 == 0, pad in {0,1})");
    return LFDM_EINVAL;
  }
  const int64_t rows = backward ? (int64_t)n_img * h * w : (int64_t)n_img * (2 * h + 2 * pad) * (2 * w + 2 * pad);
  int64_t nb = (rows * (channels / 4) + 255) / 256;
  if (nb > 65536) nb = 65536;
  if (backward) LFDM_LAUNCH(upsample_pad_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, out, n_img, h, w, channels, pad, reflect);
  else LFDM_LAUNCH(upsample_pad_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, stream, x, out, n_img, h, w, channels, pad, reflect);
  return lfdm_check_launch("upsample2_pad");
}
