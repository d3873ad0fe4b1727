// fp32-MFMA implicit-GEMM convolution on channels-last rows (include/lfdm_hip.h:
// lfdm_conv2d_cl_f32).  One workgroup = 4 wavefronts (2x2) computing a BM x BN output tile with
// v_mfma_f32_32x32x2_f32; K is walked in 32-wide chunks (one filter tap x 32 input channels on
// the fast path, a flattened (tap, channel) index on the generic path for tiny C_in).
//
//  - A tile (pixels x k) lives in LDS as [BM][33] so the MFMA A-operand read
//    (lane -> row l&31, k = 2s + (l>>5)) touches 32 distinct banks per 32-lane group;
//  - B tile (k x cout) is [32][BN], read along cout -> conflict free;
//  - global->LDS staging is register double-buffered: chunk c+1 is fetched into VGPRs before the
//    MFMAs of chunk c are issued, and written to LDS after them;
//  - D layout: col = lane&31 is the output channel, so every store instruction writes two
//    128-byte row segments of the channels-last output.
#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

struct RowInfo {
  int img;   // -1 = row beyond M
  int qy, qx;
};

__device__ __forceinline__ bool src_pixel(const lfdm_conv_params& p, int img, int qy, int qx,
                                          int tap, int64_t& pix) {
  const int ky = tap / p.kw, kx = tap - ky * p.kw;
  int iy = qy * p.stride + ky - p.pad_y;
  int ix = qx * p.stride + kx - p.pad_x;
  const int H = p.upsample ? 2 * p.hi : p.hi;
  const int W = p.upsample ? 2 * p.wi : p.wi;
  if (p.pad_mode == 1) {
    if (iy < 0) iy = -iy;
    if (iy >= H) iy = 2 * (H - 1) - iy;
    if (ix < 0) ix = -ix;
    if (ix >= W) ix = 2 * (W - 1) - ix;
  } else if (iy < 0 || iy >= H || ix < 0 || ix >= W) {
    return false;
  }
  if (p.upsample) {
    iy >>= 1;
    ix >>= 1;
  }
  pix = ((int64_t)img * p.hi + iy) * p.wi + ix;
  return true;
}

template <int BM, int BN, bool FAST>
__global__ __launch_bounds__(256) void conv_igemm_kernel(lfdm_conv_params p) {
  constexpr int BK = 32;
  constexpr int LDA = BK + 1;
  constexpr int WM = BM / 2;         // rows per wave
  constexpr int WN = BN / 2;         // cols per wave
  constexpr int TM = WM / 32;        // 32x32 tiles per wave along M
  constexpr int TN = WN / 32;        // and along N
  constexpr int A_F4 = BM * BK / 4 / 256;   // float4 per thread (fast path)
  constexpr int A_F1 = BM * BK / 256;       // floats per thread (generic path)
  constexpr int B_F4 = BK * BN / 4 / 256;

  __shared__ __attribute__((aligned(16))) float As[BM * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * BN];
  __shared__ int s_img[BM], s_qy[BM], s_qx[BM];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int hqwq = p.hq * p.wq;
  const int64_t M = (int64_t)p.n_img * hqwq;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int cin = p.c0 + p.c1;
  const int ntaps = p.kh * p.kw;
  const int ktotal = ntaps * cin;

  for (int r = tid; r < BM; r += 256) {
    int64_t m = m0 + r;
    if (m < M) {
      int img = (int)(m / hqwq);
      int rem = (int)(m - (int64_t)img * hqwq);
      int qy = rem / p.wq;
      s_img[r] = img;
      s_qy[r] = qy;
      s_qx[r] = rem - qy * p.wq;
    } else {
      s_img[r] = -1;
      s_qy[r] = 0;
      s_qx[r] = 0;
    }
  }
  __syncthreads();

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks_all = FAST ? ntaps * (cin / BK) : (ktotal + BK - 1) / BK;
  const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
  const int kc_begin = (int)((int64_t)nchunks_all * blockIdx.z / ksplit);
  const int kc_end = (int)((int64_t)nchunks_all * (blockIdx.z + 1) / ksplit);

  float4 ra4[FAST ? A_F4 : 1];
  float ra1[FAST ? 1 : A_F1];
  float4 rb[B_F4];

  auto fetch = [&](int kc) {
    if (FAST) {
      const int cpt = cin / BK;
      const int tap = kc / cpt;
      int cc = (kc - tap * cpt) * BK;
      const float* src = p.src0;
      int ld = p.ld0;
      if (cc >= p.c0) {
        cc -= p.c0;
        src = p.src1;
        ld = p.ld1;
      }
      const int cq = tid & 7;
#pragma unroll
      for (int i = 0; i < A_F4; ++i) {
        const int r = (tid >> 3) + 32 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int img = s_img[r];
        int64_t pix;
        if (img >= 0 && src_pixel(p, img, s_qy[r], s_qx[r], tap, pix))
          v = *reinterpret_cast<const float4*>(src + pix * ld + cc + 4 * cq);
        ra4[i] = v;
      }
    } else {
      const int k = kc * BK + (tid & 31);
      const bool kok = k < ktotal;
      const int tap = kok ? k / cin : 0;
      int c = kok ? k - tap * cin : 0;
      const float* src = p.src0;
      int ld = p.ld0;
      if (c >= p.c0) {
        c -= p.c0;
        src = p.src1;
        ld = p.ld1;
      }
#pragma unroll
      for (int i = 0; i < A_F1; ++i) {
        const int r = (tid >> 5) + 8 * i;
        float v = 0.f;
        const int img = s_img[r];
        int64_t pix;
        if (kok && img >= 0 && src_pixel(p, img, s_qy[r], s_qx[r], tap, pix)) v = src[pix * ld + c];
        ra1[i] = v;
      }
    }
    const int krow0 = kc * BK;
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      const int f = tid + 256 * i;
      const int row = f / (BN / 4);
      const int c4 = f - row * (BN / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int col = n0 + 4 * c4;
      if (krow0 + row < ktotal && col < p.coutp)
        v = *reinterpret_cast<const float4*>(p.weight + (int64_t)(krow0 + row) * p.coutp + col);
      rb[i] = v;
    }
  };

  auto stage = [&]() {
    if (FAST) {
      const int cq = tid & 7;
#pragma unroll
      for (int i = 0; i < A_F4; ++i) {
        const int r = (tid >> 3) + 32 * i;
        float* d = As + r * LDA + 4 * cq;
        d[0] = ra4[i].x;
        d[1] = ra4[i].y;
        d[2] = ra4[i].z;
        d[3] = ra4[i].w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_F1; ++i) {
        const int r = (tid >> 5) + 8 * i;
        As[r * LDA + (tid & 31)] = ra1[i];
      }
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      const int f = tid + 256 * i;
      *reinterpret_cast<float4*>(Bs + 4 * f) = rb[i];
    }
  };

  if (kc_begin < kc_end) {
    fetch(kc_begin);
    stage();
  }
  __syncthreads();

  const int arow = wm * WM + (lane & 31);
  const int bcol = wn * WN + (lane & 31);
  const int khalf = lane >> 5;

  for (int kc = kc_begin; kc < kc_end; ++kc) {
    const bool more = kc + 1 < kc_end;
    if (more) fetch(kc + 1);
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      const int kk = 2 * s + khalf;
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[(arow + 32 * i) * LDA + kk];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[kk * BN + bcol + 32 * j];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_32x32x2(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
    if (more) {
      stage();
      __syncthreads();
    }
  }

  // epilogue
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int img = s_img[row];
      if (img < 0) continue;
      if (ksplit > 1) {
        float* dst = p.partial + ((int64_t)blockIdx.z * M + (m0 + row)) * p.coutp;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = n0 + wn * WN + 32 * j + (lane & 31);
          if (col < p.coutp) dst[col] = acc[i][j][r];
        }
      } else {
        const int oy = s_qy[row] * p.out_scale + p.out_off_y;
        const int ox = s_qx[row] * p.out_scale + p.out_off_x;
        const int64_t orow = ((int64_t)img * p.ho + oy) * p.wo + ox;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = n0 + wn * WN + 32 * j + (lane & 31);
          if (col < p.cout) {
            float v = acc[i][j][r];
            if (p.bias) v += p.bias[col];
            if (p.residual) v += p.residual[orow * p.ldr + col];
            p.out[orow * p.ldo + col] = apply_act(v, p.act);
          }
        }
      }
    }
  }
}

// split-K epilogue: out = act(sum_z partial[z] + bias + residual)
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(lfdm_conv_params p) {
  const int hqwq = p.hq * p.wq;
  const int64_t M = (int64_t)p.n_img * hqwq;
  const int c4n = p.coutp / 4;
  const int64_t total = M * c4n;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / c4n;
    const int col = (int)(idx - m * c4n) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < p.ksplit; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(p.partial + ((int64_t)z * M + m) * p.coutp + col);
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
    const int img = (int)(m / hqwq);
    const int rem = (int)(m - (int64_t)img * hqwq);
    const int qy = rem / p.wq, qx = rem - qy * p.wq;
    const int64_t orow = ((int64_t)img * p.ho + qy * p.out_scale + p.out_off_y) * p.wo +
                         qx * p.out_scale + p.out_off_x;
    float vals[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = col + e;
      if (c < p.cout) {
        float v = vals[e];
        if (p.bias) v += p.bias[c];
        if (p.residual) v += p.residual[orow * p.ldr + c];
        p.out[orow * p.ldo + c] = apply_act(v, p.act);
      }
    }
  }
}

}  // namespace

extern "C" size_t lfdm_conv2d_partial_bytes(const lfdm_conv_params* p) {
  if (!p || p->ksplit <= 1) return 0;
  return (size_t)p->ksplit * (size_t)p->n_img * p->hq * p->wq * p->coutp * sizeof(float);
}

extern "C" int lfdm_conv2d_cl_f32(const lfdm_conv_params* pp, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!pp) { lfdm_set_error("conv2d: null params"); return LFDM_EINVAL; }
  lfdm_conv_params p = *pp;
  const int cin = p.c0 + p.c1;
  if (!p.src0 || !p.weight || !p.out || p.c0 <= 0 || p.c1 < 0 || (p.c1 > 0 && !p.src1) ||
      p.n_img <= 0 || p.hq <= 0 || p.wq <= 0 || p.kh <= 0 || p.kw <= 0 || p.cout <= 0 ||
      p.coutp < p.cout || (p.coutp % 32) != 0 || p.stride <= 0 || p.out_scale <= 0 ||
      p.ld0 < p.c0 || (p.c1 > 0 && p.ld1 < p.c1) || p.ldo < p.cout) {
    lfdm_set_error("conv2d: invalid geometry");
    return LFDM_EINVAL;
  }
  if (p.pad_mode == 1) {
    const int H = p.upsample ? 2 * p.hi : p.hi, W = p.upsample ? 2 * p.wi : p.wi;
    if (p.pad_y >= H || p.pad_x >= W || p.kh - 1 - p.pad_y >= H || p.kw - 1 - p.pad_x >= W) {
      lfdm_set_error("conv2d: reflect padding larger than the input");
      return LFDM_EINVAL;
    }
  }
  if (p.ksplit > 1 && !p.partial) { lfdm_set_error("conv2d: ksplit without partial buffer"); return LFDM_EWORKSPACE; }
  if (p.ksplit < 1) p.ksplit = 1;
  const bool fast = (p.c0 % 32 == 0) && (p.c1 % 32 == 0) && (p.ld0 % 4 == 0) &&
                    (p.c1 == 0 || p.ld1 % 4 == 0) &&
                    (((uintptr_t)p.src0 & 15) == 0) && (p.c1 == 0 || ((uintptr_t)p.src1 & 15) == 0);
  const int64_t M = (int64_t)p.n_img * p.hq * p.wq;
  const int nchunks = fast ? p.kh * p.kw * (cin / 32) : (p.kh * p.kw * cin + 31) / 32;
  if (p.ksplit > nchunks) p.ksplit = nchunks;
  // tile choice (measured on MI355X, tools/bench_conv.py): 128x128 tiles only when they still give
  // >= 256 workgroups; otherwise 64x64 tiles (more, smaller workgroups balance the 256 CUs better
  // than 128-row tiles at the UNet's M = 40*S*S) and split-K (caller) for the low-resolution levels.
  const bool wide = p.coutp >= 128 && (M / 128) * (p.coutp / 128) >= 256;
  const bool small_m = M * (int64_t)((p.coutp + 63) / 64) < 128 * 512;
  dim3 block(256);
  if (wide) {
    dim3 grid((unsigned)((M + 127) / 128), (unsigned)((p.coutp + 127) / 128), p.ksplit);
    if (fast) LFDM_LAUNCH((conv_igemm_kernel<128, 128, true>), grid, block, 0, stream, p);
    else LFDM_LAUNCH((conv_igemm_kernel<128, 128, false>), grid, block, 0, stream, p);
  } else if (small_m) {
    dim3 grid((unsigned)((M + 63) / 64), (unsigned)((p.coutp + 63) / 64), p.ksplit);
    if (fast) LFDM_LAUNCH((conv_igemm_kernel<64, 64, true>), grid, block, 0, stream, p);
    else LFDM_LAUNCH((conv_igemm_kernel<64, 64, false>), grid, block, 0, stream, p);
  } else {
    dim3 grid((unsigned)((M + 127) / 128), (unsigned)((p.coutp + 63) / 64), p.ksplit);
    if (fast) LFDM_LAUNCH((conv_igemm_kernel<128, 64, true>), grid, block, 0, stream, p);
    else LFDM_LAUNCH((conv_igemm_kernel<128, 64, false>), grid, block, 0, stream, p);
  }
  int rc = lfdm_check_launch("conv_igemm");
  if (rc) return rc;
  if (p.ksplit > 1) {
    const int64_t total = M * (p.coutp / 4);
    unsigned nb = (unsigned)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    LFDM_LAUNCH(conv_splitk_reduce_kernel, dim3(nb), block, 0, stream, p);
    rc = lfdm_check_launch("conv_splitk_reduce");
  }
  return rc;
}
