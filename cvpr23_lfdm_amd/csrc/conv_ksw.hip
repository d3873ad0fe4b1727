// "K-split across wavefronts" variant of the fp32-MFMA implicit-GEMM convolution
// (include/lfdm_hip.h: lfdm_conv2d_cl_f32 picks it for K >= 256 convolutions with zero padding).
//
// Why a second schedule: at B = 1 the UNet's full-resolution level has M = 40*32*32 = 40 960 rows and
// only 64 output channels.  128-row tiles give 320 workgroups (a wasted second wave on 256 CUs) and
// 64x64 tiles leave each wavefront one accumulator, two LDS reads and a barrier pair per 16 MFMAs -
// measured 38 % MFMA utilisation (profiles/r01_b).  Here the tile is 160 x {64,32}: 40 960 / 160 =
// 256 tiles = one per CU, and the four wavefronts of a workgroup do not split the tile, they split K:
// wave w multiplies k in [8w, 8w+8) of every 32-wide chunk for the WHOLE tile.  Per chunk a wave
// issues TM+TN ds_read_b128 and 4*TM*TN MFMAs on TM*TN independent accumulators (40 MFMAs on 10
// chains for 160x64), A and B are read from LDS exactly once, and one barrier per chunk suffices
// (LDS double buffer).  The four partial tiles are summed through LDS once at the end, which also
// gives every thread a float4 of one output row -> coalesced epilogue with bias / residual /
// activation / GroupNorm partial sums, or a split-K partial write.
#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

template <int TM, int TN>
__global__ __launch_bounds__(256) void conv_ksw_kernel(lfdm_conv_params p) {
  constexpr int BM = 32 * TM, BN = 32 * TN, BK = 32, LD = BK + 4;
  constexpr int STAGE = (BM + BN) * LD;
  constexpr int A_F4 = BM * 8 / 256;
  constexpr int B_F4 = BN * 8 / 256;
  static_assert(BM * 8 % 256 == 0 && BN * 8 % 256 == 0, "tile must split evenly over 256 threads");
  static_assert(2 * STAGE >= 4 * 32 * LD, "reduction scratch does not fit");

  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
  __shared__ int s_img[BM], s_qy[BM], s_qx[BM], s_pix[BM];
  __shared__ unsigned long long s_mask[BM];
  __shared__ float s_gn[2][4][BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int hqwq = p.hq * p.wq;
  const int64_t M = (int64_t)p.n_img * hqwq;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int cin = p.c0 + p.c1;
  const int ntaps = p.kh * p.kw;

  for (int r = tid; r < BM; r += 256) {
    const int64_t m = m0 + r;
    int img = -1, qy = 0, qx = 0;
    unsigned long long mask = 0ull;
    if (m < M) {
      img = (int)(m / hqwq);
      const int rem = (int)(m - (int64_t)img * hqwq);
      qy = rem / p.wq;
      qx = rem - qy * p.wq;
      for (int t = 0; t < ntaps; ++t) {
        const int ky = t / p.kw, kx = t - ky * p.kw;
        const int iy = qy * p.stride + ky - p.pad_y, ix = qx * p.stride + kx - p.pad_x;
        if (iy >= 0 && iy < p.hi && ix >= 0 && ix < p.wi) mask |= 1ull << t;
      }
    }
    s_img[r] = img;
    s_qy[r] = qy;
    s_qx[r] = qx;
    s_mask[r] = mask;
    s_pix[r] = (img * p.hi + qy * p.stride - p.pad_y) * p.wi + qx * p.stride - p.pad_x;
  }
  __syncthreads();

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks_all = ntaps * (cin / BK);
  const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
  const int kc_begin = (int)((int64_t)nchunks_all * blockIdx.z / ksplit);
  const int kc_end = (int)((int64_t)nchunks_all * (blockIdx.z + 1) / ksplit);
  const int nk = kc_end - kc_begin;

  const int cq = tid & 7;
  int a_pix[A_F4];
  unsigned long long a_mask[A_F4];
#pragma unroll
  for (int i = 0; i < A_F4; ++i) {
    const int r = (tid >> 3) + 32 * i;
    a_pix[i] = s_pix[r];
    a_mask[i] = s_mask[r];
  }
  float4 ra[A_F4], rb[B_F4];

  auto fetch = [&](int kc) {
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
    const int ky = tap / p.kw, kx = tap - ky * p.kw;
    const int tap_off = ky * p.wi + kx;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((a_mask[i] >> tap) & 1ull)
        v = *reinterpret_cast<const float4*>(src + (int64_t)(a_pix[i] + tap_off) * ld + cc + 4 * cq);
      ra[i] = v;
    }
    const float* wchunk = p.weight + (int64_t)kc * p.coutp * BK;
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      const int f = tid + 256 * i;
      const int n = f >> 3;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n0 + n < p.coutp) v = *reinterpret_cast<const float4*>(wchunk + (int64_t)(n0 + n) * BK + 4 * (f & 7));
      rb[i] = v;
    }
  };
  auto stage = [&](int buf) {
    float* const As = smem + buf * STAGE;
    float* const Bs = As + BM * LD;
#pragma unroll
    for (int i = 0; i < A_F4; ++i)
      *reinterpret_cast<float4*>(As + ((tid >> 3) + 32 * i) * LD + 4 * cq) = ra[i];
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      const int f = tid + 256 * i;
      *reinterpret_cast<float4*>(Bs + (f >> 3) * LD + 4 * (f & 7)) = rb[i];
    }
  };

  if (nk > 0) {
    fetch(kc_begin);
    stage(0);
  }
  __syncthreads();
  if (nk > 1) fetch(kc_begin + 1);

  const int koff = 8 * wave + 4 * (lane >> 5);   // this wave's k slice of a chunk, this lane half's quad
  const int l31 = lane & 31;
  for (int c = 0; c < nk; ++c) {
    const int cur = c & 1;
    if (c + 1 < nk) stage(cur ^ 1);               // registers hold chunk c+1
#ifndef LFDM_PROBE_NOFETCH
    if (c + 2 < nk) fetch(kc_begin + c + 2);
#endif
    const float* const As = smem + cur * STAGE;
    const float* const Bs = As + BM * LD;
    float4 a[TM], b[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(As + (32 * i + l31) * LD + koff);
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(Bs + (32 * j + l31) * LD + koff);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#ifdef LFDM_PROBE_NOMFMA
        acc[i][j][0] += a[i].x * b[j].x + a[i].y * b[j].y + a[i].z * b[j].z + a[i].w * b[j].w;
#else
        acc[i][j] = mfma_32x32x2(a[i].x, b[j].x, acc[i][j]);
        acc[i][j] = mfma_32x32x2(a[i].y, b[j].y, acc[i][j]);
        acc[i][j] = mfma_32x32x2(a[i].z, b[j].z, acc[i][j]);
        acc[i][j] = mfma_32x32x2(a[i].w, b[j].w, acc[i][j]);
#endif
      }
    __syncthreads();
  }

  // ---- cross-wave reduction + epilogue, one 32x32 tile at a time ----
  float* const scratch = smem;                     // [4 waves][32][LD]
  const int trow = tid >> 3, c4 = tid & 7;
  const bool vec_ok = (p.cout % 4 == 0) && (p.ldo % 4 == 0) && ((((uintptr_t)p.out) & 15) == 0) &&
                      (!p.residual || ((p.ldr % 4 == 0) && ((((uintptr_t)p.residual) & 15) == 0)));
  float gs[TN][4], gq[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) gs[j][e] = gq[j][e] = 0.f;

#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        scratch[wave * (32 * LD) + ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LD + l31] = acc[i][j][r];
      __syncthreads();
      float4 v = *reinterpret_cast<const float4*>(scratch + trow * LD + 4 * c4);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float4 u = *reinterpret_cast<const float4*>(scratch + w * (32 * LD) + trow * LD + 4 * c4);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      const int row = 32 * i + trow;
      const int img = s_img[row];
      const int colbase = n0 + 32 * j + 4 * c4;
      if (img >= 0) {
        if (ksplit > 1) {
          if (colbase < p.coutp)
            *reinterpret_cast<float4*>(p.partial + ((int64_t)blockIdx.z * M + (m0 + row)) * p.coutp + colbase) = v;
        } else if (colbase < p.cout) {
          const int oy = s_qy[row] * p.out_scale + p.out_off_y;
          const int ox = s_qx[row] * p.out_scale + p.out_off_x;
          const int64_t orow = ((int64_t)img * p.ho + oy) * p.wo + ox;
          if (vec_ok) {
            if (p.bias) {
              const float4 bb = *reinterpret_cast<const float4*>(p.bias + colbase);
              v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
            }
            if (p.gn_partial) {
              gs[j][0] += v.x; gs[j][1] += v.y; gs[j][2] += v.z; gs[j][3] += v.w;
              gq[j][0] += v.x * v.x; gq[j][1] += v.y * v.y; gq[j][2] += v.z * v.z; gq[j][3] += v.w * v.w;
            }
            if (p.residual) {
              const float4 rr = *reinterpret_cast<const float4*>(p.residual + orow * p.ldr + colbase);
              v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
            }
            v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act);
            v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
            *reinterpret_cast<float4*>(p.out + orow * p.ldo + colbase) = v;
          } else {
            const float vals[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int col = colbase + e;
              if (col < p.cout) {
                float t = vals[e];
                if (p.bias) t += p.bias[col];
                if (p.residual) t += p.residual[orow * p.ldr + col];
                p.out[orow * p.ldo + col] = apply_act(t, p.act);
              }
            }
          }
        }
      }
      __syncthreads();
    }
  }

  if (p.gn_partial && ksplit == 1) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float s = gs[j][e], q = gq[j][e];
#pragma unroll
        for (int m = 8; m <= 32; m <<= 1) {
          s += __shfl_xor(s, m);
          q += __shfl_xor(q, m);
        }
        if (lane < 8) {
          s_gn[0][wave][32 * j + 4 * c4 + e] = s;
          s_gn[1][wave][32 * j + 4 * c4 + e] = q;
        }
      }
    __syncthreads();
    const int cg = p.cout / p.gn_groups;
    const int gpt = BN / cg;
    if (tid < gpt && n0 + tid * cg < p.cout) {
      float s = 0.f, q = 0.f;
      for (int c = 0; c < cg; ++c)
        for (int w = 0; w < 4; ++w) {
          s += s_gn[0][w][tid * cg + c];
          q += s_gn[1][w][tid * cg + c];
        }
      const int64_t tile = m0 / BM;
      float* dst = p.gn_partial + (tile * p.gn_groups + (n0 / cg + tid)) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

}  // namespace

// bn = 64 or 32; grid z = ksplit.  Called by lfdm_conv2d_cl_f32 (conv_igemm.hip).
int lfdm_conv_ksw_launch(const lfdm_conv_params& p, int bn, hipStream_t stream) {
  const int64_t M = (int64_t)p.n_img * p.hq * p.wq;
  const dim3 grid((unsigned)((M + 159) / 160), (unsigned)((p.coutp + bn - 1) / bn), p.ksplit > 1 ? p.ksplit : 1);
  if (bn == 64) LFDM_LAUNCH((conv_ksw_kernel<5, 2>), grid, dim3(256), 0, stream, p);
  else LFDM_LAUNCH((conv_ksw_kernel<5, 1>), grid, dim3(256), 0, stream, p);
  return lfdm_check_launch("conv_ksw");
}
