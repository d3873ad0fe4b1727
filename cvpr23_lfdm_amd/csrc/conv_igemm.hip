// fp32-MFMA implicit-GEMM convolution on channels-last rows (include/lfdm_hip.h:
// lfdm_conv2d_cl_f32).  One workgroup = 4 wavefronts (2x2) computing a BM x BN output tile with
// v_mfma_f32_32x32x2_f32; K is walked in 32-wide chunks (one filter tap x 32 input channels on
// the fast path, a flattened (tap, channel) index on the generic path for tiny C_in).
//
// What keeps the non-MFMA instruction count per MFMA low (the fp32 MFMA is slow enough - 64
// cycles - that address arithmetic and LDS traffic, not the matrix pipe, bound a naive kernel):
//  - both tiles sit in LDS k-contiguous ([row][32+4]): one ds_read_b128 feeds FOUR MFMA k-steps.
//    MFMA step 4q+e takes k = 8q + 4*kh + e from lane half kh (any k pairing is legal as long as A
//    and B agree), so each lane's four steps are 4 consecutive floats; the 36-float row stride
//    makes the 16-lane b128 groups conflict-free;
//  - weights are packed [chunk][cout][32] so the B tile is a coalesced 128-byte-row copy;
//  - staging is float4 in, ds_write_b128 out, register double-buffered (chunk c+1 is fetched
//    before the MFMAs of chunk c);
//  - for stride-1 zero-padded convolutions every output row keeps a 64-bit tap-validity mask and
//    a base pixel index computed once per tile: per chunk the source address is one add;
//  - the epilogue transposes each 32x32 accumulator tile through LDS so that global stores are
//    float4 per lane (8 x 128-byte row segments per instruction) with bias / residual /
//    activation applied vectorised; optionally it also emits the per-tile GroupNorm partial sums
//    (sum, sum of squares per group) so the following GroupNorm needs no statistics pass.
#include <stdlib.h>

#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

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

// FAST: every source has a multiple of 32 channels (chunk = one tap x 32 channels, float4 loads)
// SIMPLE: zero padding, no up-sampling, <= 64 taps: mask-based addressing
template <int BM, int BN, bool FAST, bool SIMPLE, bool LN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(lfdm_conv_params p) {
  constexpr int BK = 32;
  constexpr int LD = BK + 4;         // LDS row stride of both tiles (floats)
  constexpr int WM = BM / 2;         // rows per wave
  constexpr int WN = BN / 2;         // cols per wave
  constexpr int TM = WM / 32;        // 32x32 tiles per wave along M
  constexpr int TN = WN / 32;        // and along N
  constexpr int A_F4 = BM * BK / 4 / 256;   // float4 per thread (fast path)
  constexpr int A_F1 = BM * BK / 256;       // floats per thread (generic path)
  constexpr int B_F4 = BN * BK / 4 / 256;

  // one array: A tile | B tile; the epilogue reuses it as 4 wave-private 32 x LD transpose scratches
  // (an LDS double-buffered variant of this loop measured 6 % slower on MI355X and was dropped)
  constexpr int STAGE = (BM + BN) * LD;
  __shared__ __attribute__((aligned(16))) float smem[STAGE];
  static_assert(STAGE >= 4 * 32 * LD, "epilogue scratch does not fit");
  __shared__ int s_img[BM], s_qy[BM], s_qx[BM];
  __shared__ int s_pix[BM];
  __shared__ unsigned long long s_mask[BM];
  __shared__ float s_gn[2][2][BN];   // [sum|sumsq][wm][col]
  __shared__ float s_lnm[BM], s_lnr[BM];   // fused channel-LayerNorm: per-row mean, 1/sqrt(var+eps)

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
  const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
  int zk = blockIdx.z;                             // K slice of this workgroup
  if (p.deconv4) {                                  // grid z = parity * ksplit + slice (see lfdm_conv_params.deconv4)
    const int par = zk / ksplit;
    zk -= par * ksplit;
    p.pad_y = 1 - (par >> 1);
    p.pad_x = 1 - (par & 1);
    p.out_off_y = par >> 1;
    p.out_off_x = par & 1;
    p.weight += (int64_t)par * ((ktotal + 31) / 32) * p.coutp * 32;
  }

  for (int r = tid; r < BM; r += 256) {
    int64_t m = m0 + r;
    int img = -1, qy = 0, qx = 0;
    if (m < M) {
      const int mi = (int)m;                       // M < 2^31 (checked on the host)
      img = mi / hqwq;
      const int rem = mi - img * hqwq;
      qy = rem / p.wq;
      qx = rem - qy * p.wq;
    }
    s_img[r] = img;
    s_qy[r] = qy;
    s_qx[r] = qx;
    if (SIMPLE) {
      unsigned long long mask = 0ull;
      if (img >= 0) {
        int t = 0;
        for (int ky = 0; ky < p.kh; ++ky) {
          const int iy = qy * p.stride + ky - p.pad_y;
          const bool yok = iy >= 0 && iy < p.hi;
          for (int kx = 0; kx < p.kw; ++kx, ++t) {
            const int ix = qx * p.stride + kx - p.pad_x;
            if (yok && ix >= 0 && ix < p.wi) mask |= 1ull << t;
          }
        }
      }
      s_mask[r] = mask;
      s_pix[r] = (img * p.hi + qy * p.stride - p.pad_y) * p.wi + qx * p.stride - p.pad_x;   // tap (0,0)
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

  const int nchunks_all = (ktotal + BK - 1) / BK;
  const int kc_begin = (int)((int64_t)nchunks_all * zk / ksplit);
  const int kc_end = (int)((int64_t)nchunks_all * (zk + 1) / ksplit);

  float4 ra4[FAST ? A_F4 : 1];
  float ra1[FAST ? 1 : A_F1];
  float4 rb[B_F4];

  // per-thread constants of the fast path.  FAST && SIMPLE loads go through buffer descriptors: an
  // invalid tap becomes an out-of-range offset (returns 0) instead of a branch, and chunk indices are
  // clamped instead of guarded, so the MFMA loop is one basic block (no accumulator copies).
  uint32_t a_off0[FAST ? A_F4 : 1], a_off1[FAST ? A_F4 : 1];
  unsigned long long a_mask[FAST ? A_F4 : 1];
  uint32_t b_off[B_F4];
  const int64_t in_rows = (int64_t)p.n_img * p.hi * p.wi;
  const bool use_buf = FAST && SIMPLE;
  const lfdm_buf buf0 = lfdm_make_buf(p.src0, use_buf ? (uint32_t)(((in_rows - 1) * p.ld0 + p.c0) * 4) : 0u);
  const lfdm_buf buf1 = (use_buf && p.c1 > 0) ? lfdm_make_buf(p.src1, (uint32_t)(((in_rows - 1) * p.ld1 + p.c1) * 4)) : buf0;
  const lfdm_buf bufw = lfdm_make_buf(p.weight, use_buf ? (uint32_t)((int64_t)((ktotal + BK - 1) / BK) * p.coutp * BK * 4) : 0u);
  if (FAST && SIMPLE) {
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int r = (tid >> 3) + 32 * i;
      a_off0[i] = ((uint32_t)s_pix[r] * (uint32_t)p.ld0 + 4u * (tid & 7)) * 4u;   // wraps; valid taps un-wrap
      a_off1[i] = ((uint32_t)s_pix[r] * (uint32_t)p.ld1 + 4u * (tid & 7)) * 4u;
      a_mask[i] = s_mask[r];
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      const int f = tid + 256 * i;
      b_off[i] = (n0 + (f >> 3) < p.coutp) ? (uint32_t)(((n0 + (f >> 3)) * BK + 4 * (f & 7)) * 4) : LFDM_BUF_OOB;
    }
  }

  float ln_s[FAST ? A_F4 : 1], ln_q[FAST ? A_F4 : 1];
#pragma unroll
  for (int i = 0; i < (FAST ? A_F4 : 1); ++i) ln_s[i] = ln_q[i] = 0.f;

  auto fetch = [&](int kc, float lnw) {
    if (FAST && SIMPLE) {
      const int cpt = cin / BK;
      const int tap = kc / cpt;
      int cc = (kc - tap * cpt) * BK;
      const bool second = cc >= p.c0;
      if (second) cc -= p.c0;
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      const uint32_t chunk_off = (uint32_t)(((ky * p.wi + kx) * (second ? p.ld1 : p.ld0) + cc) * 4);
      const lfdm_buf buf = second ? buf1 : buf0;
#pragma unroll
      for (int i = 0; i < A_F4; ++i) {
        const uint32_t base = second ? a_off1[i] : a_off0[i];
        const uint32_t off = ((a_mask[i] >> tap) & 1ull) ? base + chunk_off : LFDM_BUF_OOB;
        const float4 v = lfdm_buf_load_f4(buf, off);
        ra4[i] = v;
        if (LN) {
          ln_s[i] += lnw * ((v.x + v.y) + (v.z + v.w));
          ln_q[i] += lnw * ((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
        }
      }
      const uint32_t wbase = (uint32_t)kc * (uint32_t)p.coutp * BK * 4;
#pragma unroll
      for (int i = 0; i < B_F4; ++i)
        rb[i] = lfdm_buf_load_f4(bufw, b_off[i] == LFDM_BUF_OOB ? LFDM_BUF_OOB : wbase + b_off[i]);
      return;
    }
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
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int r = (tid >> 3) + 32 * i;
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
    const float* wchunk = p.weight + (int64_t)kc * p.coutp * BK;
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      const int f = tid + 256 * i;
      const int n = f >> 3, k4 = f & 7;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n0 + n < p.coutp) v = *reinterpret_cast<const float4*>(wchunk + (int64_t)(n0 + n) * BK + 4 * k4);
      rb[i] = v;
    }
  };

  auto stage = [&]() {
    float* const As = smem;
    float* const Bs = As + BM * LD;
    if (FAST) {
      const int cq = tid & 7;
#pragma unroll
      for (int i = 0; i < A_F4; ++i) {
        const int r = (tid >> 3) + 32 * i;
        *reinterpret_cast<float4*>(As + r * LD + 4 * cq) = ra4[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_F1; ++i) {
        const int r = (tid >> 5) + 8 * i;
        As[r * LD + (tid & 31)] = ra1[i];
      }
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      const int f = tid + 256 * i;
      *reinterpret_cast<float4*>(Bs + (f >> 3) * LD + 4 * (f & 7)) = rb[i];
    }
  };

  const int arow = wm * WM + (lane & 31);
  const int bcol = wn * WN + (lane & 31);
  const int khalf = lane >> 5;
  const int nk = kc_end - kc_begin;

  if (nk > 0) {
    fetch(kc_begin, 1.f);
    stage();
  }
  __syncthreads();

  for (int c = 0; c < nk; ++c) {
    const bool more = c + 1 < nk;
#ifndef LFDM_PROBE_NOFETCH
    fetch(kc_begin + (more ? c + 1 : c), more ? 1.f : 0.f);     // clamped: the last iteration re-reads its chunk
#endif
    const float* const As = smem;
    const float* const Bs = As + BM * LD;
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      float4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *reinterpret_cast<const float4*>(As + (arow + 32 * i) * LD + 8 * q + 4 * khalf);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const float4*>(Bs + (bcol + 32 * j) * LD + 8 * q + 4 * khalf);
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
    }
    __syncthreads();
    stage();                          // unconditional (harmless after the last chunk): no branch in the loop
    __syncthreads();
  }

  if (FAST && LN) {
    // channel-LayerNorm statistics of every row of the tile (the 1x1 conv streamed all C channels)
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      float sv = ln_s[i], qv = ln_q[i];
#pragma unroll
      for (int m = 1; m <= 4; m <<= 1) {
        sv += __shfl_xor(sv, m);
        qv += __shfl_xor(qv, m);
      }
      if (ksplit > 1) {
        // split-K: this workgroup saw only its K slice of the rows - the (sum, sum of squares) slices go behind the slabs
        // ([ksplit][M][2]) and conv_splitk_reduce_kernel finishes the LayerNorm (one column tile writes them)
        const int r = (tid >> 3) + 32 * i;
        if ((tid & 7) == 0 && blockIdx.y == 0 && s_img[r] >= 0) {
          float* st = p.partial + (int64_t)ksplit * M * p.coutp + ((int64_t)blockIdx.z * M + (m0 + r)) * 2;
          st[0] = sv;
          st[1] = qv;
        }
        continue;
      }
      if ((tid & 7) == 0) {
        const float inv_c = 1.0f / (float)cin;
        const float mean = sv * inv_c;
        float var = qv * inv_c - mean * mean;
        if (var < 0.f) var = 0.f;
        const int r = (tid >> 3) + 32 * i;
        s_lnm[r] = mean;
        s_lnr[r] = 1.0f / sqrtf(var + p.ln_eps);
      }
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogue
  if (ksplit > 1) {
    // raw partial sums, reduced (with bias / residual / activation) by conv_splitk_reduce_kernel
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (s_img[row] < 0) continue;
        float* dst = p.partial + ((int64_t)blockIdx.z * M + (m0 + row)) * p.coutp;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = n0 + wn * WN + 32 * j + (lane & 31);
          if (col < p.coutp) dst[col] = acc[i][j][r];
        }
      }
    return;
  }

  // transpose each 32x32 accumulator tile through a wave-private LDS scratch (aliases the A tile)
  float* scratch = smem + wave * (32 * LD);
  const bool vec_ok = (p.cout % 4 == 0) && (p.ldo % 4 == 0) && ((((uintptr_t)p.out) & 15) == 0) &&
                      (!p.residual || ((p.ldr % 4 == 0) && ((((uintptr_t)p.residual) & 15) == 0)));
  const int c4 = lane & 7, rsub = lane >> 3;
  float gs[TN][4], gq[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) gs[j][e] = gq[j][e] = 0.f;
  // bias / LayerNorm column sums (per column tile) and the residual float4 of EVERY pass are requested before the first pass: inside
  // the pass loop (a barrier pair per 32x32 tile) each was a load -> wait -> store round trip, 16 of them for a 128x128 tile
  // (the residual rows one row tile i at a time: all TM*TN*4 float4 at once spilled in the 128x128 instantiation)
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 pre_b[TN], pre_w[TN], pre_r[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int colbase = n0 + wn * WN + 32 * j + 4 * c4;
    const bool cok = vec_ok && colbase < p.cout;
    pre_b[j] = (cok && p.bias) ? *reinterpret_cast<const float4*>(p.bias + colbase) : z4;
    pre_w[j] = (FAST && LN && cok) ? *reinterpret_cast<const float4*>(p.ln_wsum + colbase) : z4;
  }

#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = wm * WM + 32 * i + it * 8 + rsub;
        const int colbase = n0 + wn * WN + 32 * j + 4 * c4;
        pre_r[j][it] = z4;
        if (vec_ok && p.residual && colbase < p.cout && s_img[row] >= 0) {
          const int64_t orow = ((int64_t)s_img[row] * p.ho + s_qy[row] * p.out_scale + p.out_off_y) * p.wo + s_qx[row] * p.out_scale + p.out_off_x;
          pre_r[j][it] = *reinterpret_cast<const float4*>(p.residual + orow * p.ldr + colbase);
        }
      }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r)
        scratch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LD + (lane & 31)] = acc[i][j][r];
      __syncthreads();
      const int colbase = n0 + wn * WN + 32 * j + 4 * c4;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int trow = it * 8 + rsub;
        const int row = wm * WM + 32 * i + trow;
        const int img = s_img[row];
        if (img < 0 || colbase >= p.cout) continue;
        float4 v = *reinterpret_cast<const float4*>(scratch + trow * LD + 4 * c4);
        const int oy = s_qy[row] * p.out_scale + p.out_off_y;
        const int ox = s_qx[row] * p.out_scale + p.out_off_x;
        const int64_t orow = ((int64_t)img * p.ho + oy) * p.wo + ox;
        if (vec_ok) {
          if (FAST && LN) {      // y = rstd * (x.W' - mean * sum_c W')
            const float4 ws = pre_w[j];
            const float mu = s_lnm[row], rs = s_lnr[row];
            v.x = rs * (v.x - mu * ws.x); v.y = rs * (v.y - mu * ws.y);
            v.z = rs * (v.z - mu * ws.z); v.w = rs * (v.w - mu * ws.w);
          }
          {
            const float4 bb = pre_b[j];
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
          }
          if (p.gn_partial) {
            gs[j][0] += v.x; gs[j][1] += v.y; gs[j][2] += v.z; gs[j][3] += v.w;
            gq[j][0] += v.x * v.x; gq[j][1] += v.y * v.y; gq[j][2] += v.z * v.z; gq[j][3] += v.w * v.w;
          }
          {
            const float4 rr = pre_r[j][it];
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
  }

  if (p.gn_partial) {
    // per-tile GroupNorm partial sums: lanes with equal c4 hold the same 4 columns
    __syncthreads();
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
          const int col = wn * WN + 32 * j + 4 * c4 + e;
          s_gn[0][wm][col] = s;
          s_gn[1][wm][col] = q;
        }
      }
    __syncthreads();
    const int cg = p.cout / p.gn_groups;              // channels per group
    const int gpt = BN / cg;                          // groups covered by this N tile (BN % cg == 0)
    if (tid < gpt && n0 + tid * cg < p.cout) {
      float s = 0.f, q = 0.f;
      for (int c = 0; c < cg; ++c) {
        s += s_gn[0][0][tid * cg + c] + s_gn[0][1][tid * cg + c];
        q += s_gn[1][0][tid * cg + c] + s_gn[1][1][tid * cg + c];
      }
      const int64_t tile = m0 / BM;                   // gn_pixels % BM == 0: one sample per tile
      float* dst = p.gn_partial + (tile * p.gn_groups + (n0 / cg + tid)) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

// split-K epilogue: out = act(sum_z partial[z] + bias + residual).  One workgroup owns SPLITK_ROWS
// consecutive output rows (all channels), so it can also emit the GroupNorm partial sums of its rows
// (gn_partial[(rowblock*groups + g)*2]); gn_pixels % SPLITK_ROWS == 0 keeps a block inside one sample.
constexpr int SPLITK_ROWS = 16;
// grid (row blocks, column chunks): blockIdx.y owns float4 columns [y*cw, (y+1)*cw), cw = c4n / gridDim.y
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(lfdm_conv_params p) {
  __shared__ float red_s[256], red_q[256];
  const int tid = threadIdx.x;
  const int hqwq = p.hq * p.wq;
  const int M = p.n_img * hqwq;
  const int c4n = p.coutp / 4;
  const int cw = c4n / gridDim.y;                  // float4 columns of this block
  const int cq0 = blockIdx.y * cw;
  const int m0 = blockIdx.x * SPLITK_ROWS;
  const int items = SPLITK_ROWS * cw;
  if (p.deconv4) {                                  // grid z = parity: slabs [4][ksplit][M][coutp]
    const int par = blockIdx.z;
    p.partial += (int64_t)par * p.ksplit * M * p.coutp;
    p.out_off_y = par >> 1;
    p.out_off_x = par & 1;
  }
  float gs = 0.f, gq = 0.f;           // with 256 % cw == 0 a thread always sees the same column quad
  for (int it = tid; it < items; it += 256) {
    const int r = it / cw;
    const int col = (cq0 + it - r * cw) * 4;
    const int m = m0 + r;
    if (m >= M) continue;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* pp = p.partial + (int64_t)m * p.coutp + col;
    const int64_t zs = (int64_t)M * p.coutp;
    int z = 0;
    for (; z + 4 <= p.ksplit; z += 4) {            // four independent loads in flight
      const float4 v0 = *reinterpret_cast<const float4*>(pp + (z + 0) * zs);
      const float4 v1 = *reinterpret_cast<const float4*>(pp + (z + 1) * zs);
      const float4 v2 = *reinterpret_cast<const float4*>(pp + (z + 2) * zs);
      const float4 v3 = *reinterpret_cast<const float4*>(pp + (z + 3) * zs);
      s.x += (v0.x + v1.x) + (v2.x + v3.x);
      s.y += (v0.y + v1.y) + (v2.y + v3.y);
      s.z += (v0.z + v1.z) + (v2.z + v3.z);
      s.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; z < p.ksplit; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(pp + z * zs);
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
    const int img = m / hqwq;
    const int rem = m - img * hqwq;
    const int qy = rem / p.wq, qx = rem - qy * p.wq;
    const int64_t orow = ((int64_t)img * p.ho + qy * p.out_scale + p.out_off_y) * p.wo +
                         qx * p.out_scale + p.out_off_x;
    float vals[4] = {s.x, s.y, s.z, s.w};
    if (p.ln_wsum) {                                 // fused channel LayerNorm: y = rstd * (x.W' - mean * sum_c W')
      const float* st = p.partial + (int64_t)p.ksplit * zs + (int64_t)m * 2;
      float sv = 0.f, qv = 0.f;
      for (int zz = 0; zz < p.ksplit; ++zz) {
        sv += st[(int64_t)zz * M * 2];
        qv += st[(int64_t)zz * M * 2 + 1];
      }
      const float inv_c = 1.0f / (float)(p.c0 + p.c1);
      const float mean = sv * inv_c;
      float var = qv * inv_c - mean * mean;
      if (var < 0.f) var = 0.f;
      const float rs = 1.0f / sqrtf(var + p.ln_eps);
#pragma unroll
      for (int e = 0; e < 4; ++e) vals[e] = rs * (vals[e] - mean * (col + e < p.coutp ? p.ln_wsum[col + e] : 0.f));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = col + e;
      if (c < p.cout) {
        float v = vals[e];
        if (p.bias) v += p.bias[c];
        gs += v;
        gq += v * v;
        if (p.residual) v += p.residual[orow * p.ldr + c];
        p.out[orow * p.ldo + c] = apply_act(v, p.act);
      }
    }
  }
  if (p.gn_partial) {
    red_s[tid] = gs;
    red_q[tid] = gq;
    __syncthreads();
    const int cg4 = (p.cout / p.gn_groups) / 4;      // float4 columns per group
    const int gpb = cw / cg4;                        // groups owned by this block (cw % cg4 == 0)
    if (tid < gpb) {
      float ts = 0.f, tq = 0.f;
      // thread t touched local column quad t % cw (256 % cw == 0): local group tid owns [tid*cg4, +cg4)
      for (int rep = 0; rep < 256; rep += cw)
        for (int k = 0; k < cg4; ++k) {
          ts += red_s[rep + tid * cg4 + k];
          tq += red_q[rep + tid * cg4 + k];
        }
      float* dst = p.gn_partial + ((int64_t)blockIdx.x * p.gn_groups + cq0 / cg4 + tid) * 2;
      dst[0] = ts;
      dst[1] = tq;
    }
  }
}

// The same pass for the shapes the sampler produces (round 4): KS slabs known at compile time, all KS 16-byte loads of an item in flight
// together (one memory round trip instead of two), bias / residual / result as float4, no per-element index arithmetic.  Needs what the
// launch site checks: cout == coutp, 16-byte aligned rows, no LayerNorm fold, output pixel = row (no out_scale / offsets), not deconv4.
// Summation order z = 0, 1, ... (fixed: bit reproducible).  Same grid, same GroupNorm partial layout as the generic kernel.
template <int KS>
__global__ __launch_bounds__(256) void conv_splitk_reduce_vec_kernel(lfdm_conv_params p) {
  __shared__ float red_s[256], red_q[256];
  const int tid = threadIdx.x;
  const int M = p.n_img * p.hq * p.wq;
  const int c4n = p.coutp / 4;
  const int cw = c4n / gridDim.y;
  const int cq0 = blockIdx.y * cw;
  const int m0 = blockIdx.x * SPLITK_ROWS;
  const int items = SPLITK_ROWS * cw;
  const int64_t zs = (int64_t)M * p.coutp;
  float gs = 0.f, gq = 0.f;
  for (int it = tid; it < items; it += 256) {
    const int r = it / cw;
    const int col = (cq0 + it - r * cw) * 4;
    const int m = m0 + r;
    if (m >= M) continue;
    const float* pp = p.partial + (int64_t)m * p.coutp + col;
    float4 v[KS];
#pragma unroll
    for (int z = 0; z < KS; ++z) v[z] = *reinterpret_cast<const float4*>(pp + z * zs);
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), rr = bb;
    if (p.bias) bb = *reinterpret_cast<const float4*>(p.bias + col);
    if (p.residual) rr = *reinterpret_cast<const float4*>(p.residual + (int64_t)m * p.ldr + col);
    float4 s = v[0];
#pragma unroll
    for (int z = 1; z < KS; ++z) { s.x += v[z].x; s.y += v[z].y; s.z += v[z].z; s.w += v[z].w; }
    s.x += bb.x; s.y += bb.y; s.z += bb.z; s.w += bb.w;
    gs += (s.x + s.y) + (s.z + s.w);
    gq += (s.x * s.x + s.y * s.y) + (s.z * s.z + s.w * s.w);
    s.x = apply_act(s.x + rr.x, p.act); s.y = apply_act(s.y + rr.y, p.act);
    s.z = apply_act(s.z + rr.z, p.act); s.w = apply_act(s.w + rr.w, p.act);
    *reinterpret_cast<float4*>(p.out + (int64_t)m * p.ldo + col) = s;
  }
  if (p.gn_partial) {
    red_s[tid] = gs;
    red_q[tid] = gq;
    __syncthreads();
    const int cg4 = (p.cout / p.gn_groups) / 4;
    const int gpb = cw / cg4;
    if (tid < gpb) {
      float ts = 0.f, tq = 0.f;
      for (int rep = 0; rep < 256; rep += cw)
        for (int k = 0; k < cg4; ++k) {
          ts += red_s[rep + tid * cg4 + k];
          tq += red_q[rep + tid * cg4 + k];
        }
      float* dst = p.gn_partial + ((int64_t)blockIdx.x * p.gn_groups + cq0 / cg4 + tid) * 2;
      dst[0] = ts;
      dst[1] = tq;
    }
  }
}

// column chunks for the reduce grid: enough workgroups to cover the chip, chunks of >= 16 float4 that
// divide the row and (with fused GroupNorm statistics) hold whole groups and divide 256
int splitk_col_chunks(const lfdm_conv_params& p, int64_t M) {
  const int c4n = p.coutp / 4;
  const int64_t rowblocks = (M + SPLITK_ROWS - 1) / SPLITK_ROWS;
  int gy = 1;
  while (rowblocks * gy < 256 && gy < 16) {
    const int cand = gy * 2;
    if (c4n % cand != 0) break;
    const int cw = c4n / cand;
    if (cw < 16) break;
    if (p.gn_partial) {
      const int cg4 = (p.cout / p.gn_groups) / 4;
      if (cw % cg4 != 0 || 256 % cw != 0) break;
    }
    gy = cand;
  }
  return gy;
}

template <int BM, int BN>
void launch_conv(const lfdm_conv_params& p, bool fast, bool simple, dim3 grid, hipStream_t stream) {
  dim3 block(256);
  if (fast && simple && p.ln_wsum) LFDM_LAUNCH((conv_igemm_kernel<BM, BN, true, true, true>), grid, block, 0, stream, p);
  else if (fast && simple) LFDM_LAUNCH((conv_igemm_kernel<BM, BN, true, true, false>), grid, block, 0, stream, p);
  else if (fast) LFDM_LAUNCH((conv_igemm_kernel<BM, BN, true, false, false>), grid, block, 0, stream, p);
  else LFDM_LAUNCH((conv_igemm_kernel<BM, BN, false, false, false>), grid, block, 0, stream, p);
}

}  // namespace

int lfdm_conv_ksw_launch(const lfdm_conv_params& p, int bn, hipStream_t stream);   // conv_ksw.hip
int lfdm_conv_wino_launch(const lfdm_conv_params& p, int bn, bool fuse_reduce, int bal_whole, hipStream_t stream);  // conv_wino.hip
int lfdm_conv_pw_launch(const lfdm_conv_params& p, hipStream_t stream);            // conv_pw.hip
int lfdm_conv_wino4_launch(const lfdm_conv_params& p, hipStream_t stream);         // conv_wino4.hip

namespace {

struct ConvPlan;
bool splitk_fused(const ConvPlan& pl, const lfdm_conv_params& p);
bool wino_fuse_geometry_ok(const lfdm_conv_params& p, int bn);
int wino_balance(const ConvPlan& pl, const lfdm_conv_params& p, int* extra_slabs);

struct ConvPlan {
  int kind;        // 0 = 2x2-wave tiles (this file), 1 = K-split-across-waves 160-row tiles (conv_ksw.hip),
                   // 2 = Winograd F(2x2,3x3), 128-pixel x 32-column tiles (conv_wino.hip),
                   // 3 = pointwise register-operand GEMM, 32-row tiles, never split-K (conv_pw.hip)
                   // 4 = Winograd F(4x4,3x3), 512-pixel x 32-column tiles, batched shapes only (conv_wino4.hip)
  int bm, bn, ksplit;
  bool fast, simple;
};

int conv_force() {   // debugging aid for tools/bench_conv.py: LFDM_CONV_FORCE=igemm|ksw
  static int v = -2;
  if (v == -2) {
    const char* e = getenv("LFDM_CONV_FORCE");
    v = !e ? -1 : (e[0] == 'k' ? 1 : 0);
  }
  return v;
}

// One place decides tile shape and split-K (measured on MI355X with tools/bench_conv.py):
//  - K >= 256, zero padding, C_in % 32 == 0  -> 160-row K-split-across-waves tiles; 160x64 when that
//    already yields >= 224 workgroups, else 160x32, else additionally split K over blockIdx.z so that
//    about 256 workgroups (one per CU) exist;
//  - otherwise the 2x2-wave kernel: 128x128 tiles when they give >= 256 workgroups, else 64x64, with
//    split-K for the low-resolution levels.
bool wino_enabled() {   // on by default (faster than the direct kernels on every 3x3 shape of tools/bench_conv.py); LFDM_WINO=0
  const char* e = getenv("LFDM_WINO");       // forces the direct form.  Read per call: tests and tools toggle it at run time
  return !(e && e[0] == '0');
}

ConvPlan make_plan(const lfdm_conv_params& p) {
  ConvPlan pl;
  const int cin = p.c0 + p.c1;
  const int64_t M = (int64_t)p.n_img * p.hq * p.wq;
  const int nchunks = (p.kh * p.kw * cin + 31) / 32;
  pl.fast = (p.c0 % 32 == 0) && (p.c1 % 32 == 0) && (p.ld0 % 4 == 0) && (p.c1 == 0 || p.ld1 % 4 == 0) &&
            (((uintptr_t)p.src0 & 15) == 0) && (p.c1 == 0 || ((uintptr_t)p.src1 & 15) == 0);
  pl.simple = !p.upsample && p.pad_mode == 0 && p.kh * p.kw <= 64 &&
              (int64_t)p.n_img * p.hi * p.wi < (1ll << 31) - (1 << 20);
  const int user_k = p.ksplit;                       // 0 = choose
  const int64_t in_rows = (int64_t)p.n_img * p.hi * p.wi;
  const bool fits32 = in_rows * p.ld0 * 4 < (1ll << 32) - 64 && in_rows * (p.c1 ? p.ld1 : 1) * 4 < (1ll << 32) - 64 &&
                      (int64_t)nchunks * p.coutp * 128 < (1ll << 32) - 64;      // buffer descriptors: 32-bit offsets
  const bool vec_ok = (p.cout % 4 == 0) && (p.ldo % 4 == 0) && ((((uintptr_t)p.out) & 15) == 0) &&
                      (!p.residual || ((p.ldr % 4 == 0) && ((((uintptr_t)p.residual) & 15) == 0))) &&
                      (!p.bias || (((uintptr_t)p.bias) & 15) == 0);
  pl.simple = pl.simple && fits32;      // the mask/buffer-descriptor path uses 32-bit byte offsets
  bool ksw = pl.fast && pl.simple && nchunks >= 8 && M >= 160 && !p.ln_wsum && fits32 && vec_ok;
  if (conv_force() == 0) ksw = false;
  if (conv_force() == 1 && pl.fast && pl.simple && !p.ln_wsum && fits32 && vec_ok) ksw = true;
  const int up = p.upsample ? 2 : 1;
  const bool wino = p.weight_wino && wino_enabled() && p.kh == 3 && p.kw == 3 && p.stride == 1 && p.pad_y == 1 && p.pad_x == 1 &&
                    p.pad_mode == 0 && fits32 && (int64_t)p.n_img * p.hq * p.wq < (1ll << 31) - (1 << 20) &&
                    p.hq == up * p.hi && p.wq == up * p.wi && p.hq % 2 == 0 && p.wq % 2 == 0 && p.c0 % 16 == 0 &&
                    p.c1 % 16 == 0 && p.ld0 % 4 == 0 && (p.c1 == 0 || p.ld1 % 4 == 0) && (((uintptr_t)p.src0 & 15) == 0) &&
                    (p.c1 == 0 || ((uintptr_t)p.src1 & 15) == 0) && (((uintptr_t)p.weight_wino) & 15) == 0 && !p.ln_wsum &&
                    p.out_scale == 1 && p.out_off_y == 0 && p.out_off_x == 0 && (p.pool2 ? (2 * p.ho == p.hq && 2 * p.wo == p.wq) : (p.ho == p.hq && p.wo == p.wq)) && (int64_t)16 * (cin / 16) * p.coutp * 64 < (1ll << 32) - 64 &&
                    !p.deconv4 && vec_ok;      // (float4 epilogue)
  // 1x1 / stride 1 projections (to_qkv with the LayerNorm fold, to_out, res_conv): the register-operand GEMM of conv_pw.hip has no
  // staging prologue and never needs split-K slabs + a reduce launch.  Measured per shape against the LDS-staged schedules
  // (tools/bench_pw.py, profiles/r03_a_bench_pw.txt): it wins wherever those would split K (every projection of the 4x4 level,
  // the K >= 256 ones with few output columns above it) and at M <= 1024 rows; with one K slice and thousands of rows the staged,
  // fully coalesced tiles are as fast or faster (its fragment-shaped loads cost four L1 line look-ups per 128-byte line).
  // LFDM_PW=0 disables it, LFDM_PW=2 takes it for every eligible geometry, LFDM_PW_MAXM bounds the row count.
  static const long pw_max_m = [] { const char* e = lfdm_knob("LFDM_PW_MAXM"); return e ? atol(e) : 16384l; }();
  const char* pw_env = getenv("LFDM_PW");
  const int pw_mode = pw_env ? atoi(pw_env) : 1;
  const bool pw_ok = pw_mode != 0 && conv_force() < 0 && p.kh == 1 && p.kw == 1 && p.stride == 1 && !p.upsample && p.pad_y == 0 &&
                     p.pad_x == 0 && pl.fast && fits32 && vec_ok && user_k <= 1 && !p.gn_partial && !p.deconv4 && !(p.groups > 1) && !p.pool2 &&
                     p.out_scale == 1 && p.out_off_y == 0 && p.out_off_x == 0 && p.ho == p.hq && p.wo == p.wq && p.hq == p.hi && p.wq == p.wi &&
                     (M <= pw_max_m || (p.res_gn_partial && M <= 65536)) && (!p.ln_wsum || (p.c1 == 0 && (((uintptr_t)p.ln_wsum) & 15) == 0)) && !p.tile_counters;
  // ... and its gather form (round 6) for the Downsample convolutions: 4x4 / stride 2 / zero pad over one source, K = 16 C a multiple of 128,
  // K <= 2048, M <= 16 384 output rows (the KSW schedule + its reduce launch: 28 us for 1.3 GFLOP at every level of a B = 1 step; measured in the
  // step: 19.9 / 22.2 us at the first two levels, but 34.0 us for the 256-channel one - 160 workgroups walking K = 4096 - which stays on KSW)
  const bool pwg_ok = pw_mode != 0 && conv_force() < 0 && p.kh == 4 && p.kw == 4 && p.stride == 2 && !p.upsample && p.pad_mode == 0 && p.c1 == 0 &&
                      p.c0 % 32 == 0 && (16 * p.c0) % 128 == 0 && 16 * p.c0 <= 2048 && p.ld0 % 4 == 0 && (((uintptr_t)p.src0) & 15) == 0 && fits32 && vec_ok && user_k <= 1 &&
                      !p.gn_partial && !p.deconv4 && !(p.groups > 1) && !p.pool2 && p.out_scale == 1 && p.out_off_y == 0 && p.out_off_x == 0 &&
                      p.ho == p.hq && p.wo == p.wq && !p.ln_wsum && !p.res_gn_partial && M <= pw_max_m && !p.tile_counters &&
                      (int64_t)p.n_img * p.hi * p.wi < (1ll << 31) - (1 << 20);
  // F(4x4,3x3) (conv_wino4.hip): an opt-in of the caller (weight_wino4 given: its fp32 error is ~4e-6 of the output scale against
  // ~1e-6 for F(2x2)) and only where every CU gets several of its one-per-CU workgroups - the batched shapes of training / throughput
  // mode.  LFDM_WINO4=0 disables it, LFDM_WINO4_MIN overrides the workgroup-count threshold.
  if (wino && p.weight_wino4 && p.c1 == 0 && p.c0 % 32 == 0 && p.hq % 4 == 0 && p.wq % 4 == 0 && !(p.groups > 1) && !p.pool2 &&
      !p.gn_partial && user_k <= 1 && p.coutp % 32 == 0 && (((uintptr_t)p.weight_wino4) & 15) == 0 &&
      (int64_t)36 * (cin / 8) * p.coutp * 32 < (1ll << 32) - 64) {
    const char* em = getenv("LFDM_WINO4_MIN");      // (read per call: tests and tools toggle it at run time)
    const long min_blocks4 = em ? atol(em) : 2048l;
    const char* e4 = getenv("LFDM_WINO4");
    const int64_t blocks4 = (((int64_t)p.n_img * (p.hq / 4) * (p.wq / 4) + 31) / 32) * (p.coutp / 32);
    if (!(e4 && e4[0] == '0') && blocks4 >= min_blocks4) {
      pl.kind = 4;
      pl.bm = 512;
      pl.bn = 32;
      pl.ksplit = 1;
      return pl;
    }
  }
  if (wino) {
    pl.kind = 2;
    pl.bm = 128;
    pl.bn = 32;
    const int64_t ntiles = (int64_t)p.n_img * (p.hq / 2) * (p.wq / 2);
    if (const char* e = getenv("LFDM_WINO_BN"))       // 64-column workgroups (tools/bench_conv.py, tests)
      if (e[0] == '6' && p.coutp % 64 == 0) pl.bn = 64;
    {
      // 64-column workgroups (half the patch loads / transforms per MFMA, two workgroups per CU) pay where every CU still gets
      // several of them: the batched shapes of training / throughput mode (B = 8 training step 157.0 -> 150.6 ms at >= 1536
      // workgroups = six per CU, 151.7 at 3072, 150.8 at 768: profiles/r02_u_bn64_train.txt), never the B = 1 sampler
      // (<= 640 such workgroups).  LFDM_WINO_BN64_MIN (--knobs builds) overrides the threshold, 0 disables.
      // (round 6: 1280 - the LFAE decoder's 256 -> 256 convolutions over 40 frames of 32x32, 320 x 4 such workgroups, twelve per video: -0.5 ms per video,
      //  profiles/r06_aj_bn64_decode_ab.txt; the training step measured the same at 768 and 1536)
      static const long min_blocks64 = [] { const char* e = lfdm_knob("LFDM_WINO_BN64_MIN"); return e ? atol(e) : 1280l; }();
      if (min_blocks64 > 0 && p.coutp % 64 == 0 && !(p.groups > 1) && ((ntiles + 31) / 32) * (p.coutp / 64) >= min_blocks64) pl.bn = 64;
    }
    const int64_t blocks = ((ntiles + 31) / 32) * ((p.coutp + pl.bn - 1) / pl.bn);
    const int nch = cin / 16 / (p.groups > 1 ? p.groups : 1);      // chunks of one output channel's reduction
    // (16-tile workgroups on v_mfma_f32_16x16x4 - twice the workgroups at half the matrix work, half the split-K factor - were built
    //  and measured in round 2, profiles/r02_n_tile16_sweep.txt: two-wave form slower everywhere; four-wave form -2..3 us at 16x16,
    //  +1..2 us at 32x32, equal at 4x4 / 8x8, +0.45 % end to end when selected below 512 workgroups - not worth a second kernel; removed)
    int k = 1;
    // Slice length and the smallest reduction that is split (sweep knobs LFDM_WINO_SLICE_CHUNKS / LFDM_WINO_SPLIT_MIN_CHUNKS, read ONCE:
    // tools/sweep_wino_slices.sh runs one process per setting).  Round 5: with the slabs reduced inside the launch (conv_wino.hip FUSE) the
    // 8-chunk convolutions of the 16x16 level pay for two slices of four chunks (profiles/r05_t_sweep_wino_slices.txt; only for callers
    // that hand in tile_counters - without the in-launch reduction the extra slices would buy a reduce launch).  Round 6: the slabs
    // became 16-byte write-through accesses and four chunks per slice win at every depth (profiles/r06_i_sweep_wino_slices.txt: 274.3 ms
    // per video against 275.0 with five chunks from 16 chunks on; 3 / 6 per slice lose).
    static const int env_sc = [] { const char* e = lfdm_knob("LFDM_WINO_SLICE_CHUNKS"); return e ? atoi(e) : 0; }();
    static const int env_mc = [] { const char* e = lfdm_knob("LFDM_WINO_SPLIT_MIN_CHUNKS"); return e ? atoi(e) : 0; }();
    const bool fuse = wino_fuse_geometry_ok(p, pl.bn);      // (not merely "ticket words were handed in")
    const int min_chunks = env_mc > 0 ? env_mc : (fuse ? 8 : 16);
    const int slice_chunks = env_sc > 0 ? env_sc : (fuse ? 4 : (nch < 16 ? 4 : 5));
    if (blocks < 512 && nch >= min_chunks) {
      // Split-K from tools/sweep_ksplit.sh (profiles/r02_c_ksplit_sweep.txt): a workgroup that is alone on its CU runs a
      // chunk in ~1.8 us (the matrix pipe needs 1.0), fixed costs are ~8 us per workgroup, and in the sampler the filters
      // arrive cold from HBM - so below two workgroups per CU slices of ~5 chunks win although the slabs need a reduce pass
      k = nch / slice_chunks;
      if (k > 1024 / blocks) k = (int)(1024 / blocks);
      if (k > 8) k = 8;
      if (k < 1) k = 1;
    }
    pl.ksplit = user_k >= 1 ? user_k : k;
    if (pl.ksplit > nch) pl.ksplit = nch;
    if (p.pool2) pl.ksplit = 1;                     // the pooled epilogue needs the finished sums
    return pl;
  }
  if (ksw) {
    pl.kind = 1;
    pl.bm = 160;
    const int64_t mt = (M + 159) / 160 * (p.deconv4 ? 4 : 1);      // deconv4: four problems share the launch
    const int64_t t64 = mt * ((p.coutp + 63) / 64), t32 = mt * ((p.coutp + 31) / 32);
    pl.bn = (t64 >= 224 && p.coutp > 32) ? 64 : 32;      // <= 32 (padded) output channels: the 64-column tile would be half empty
    int k = 1;
    if (t64 < 224 && t32 < 224) {
      k = (int)(256 / t32);
      if (k > nchunks / 4) k = nchunks / 4;
      if (k < 1) k = 1;
    }
    pl.ksplit = user_k >= 1 ? user_k : k;
  } else {
    pl.kind = 0;
    // 128x128 tiles only from two workgroups per CU up (round 4: the 320 / 480-workgroup launches of a B = 1 step - the merged output heads'
    // res_conv, the LayerNorm-fused qkv projections at 16x16 - run 128x64 tiles instead: 295.7 -> 293.0 ms per video, profiles/r04_f_*)
    static const long wide_min = [] { const char* e = lfdm_knob("LFDM_IGEMM_WIDE_MIN"); return e ? atol(e) : 512l; }();
    const bool wide = p.coutp >= 128 && (M / 128) * (p.coutp / 128) >= wide_min;
    const bool small_m = M * (int64_t)((p.coutp + 63) / 64) < 128 * 512;
    pl.bm = wide ? 128 : (small_m ? 64 : 128);
    pl.bn = wide ? 128 : 64;
    int k = 1;
    const int64_t tiles = ((M + pl.bm - 1) / pl.bm) * ((p.coutp + pl.bn - 1) / pl.bn) * (p.deconv4 ? 4 : 1);
    if (tiles < 256 && nchunks >= 8) {
      k = (int)(512 / tiles);
      if (k > nchunks / 4) k = nchunks / 4;
      if (k > 16) k = 16;
      if (k < 1) k = 1;
    }
    pl.ksplit = user_k >= 1 ? user_k : k;
  }
  if (pl.ksplit > nchunks) pl.ksplit = nchunks;
  if (pl.ksplit < 1) pl.ksplit = 1;
  // (res_gn_*: the residual's GroupNorm + SiLU in the epilogue exists on the pointwise schedule only - it takes every eligible geometry then)
  // (round 6: without a LayerNorm fold the pointwise schedule also takes the 10 240-row projections of the 16x16 level - to_out after the attention cores,
  //  128 workgroups on the K-split-across-waves schedule; with the fold it is 2.4 ms per video slower there: profiles/r06_au_*, r06_bg_*)
  //  (only the narrow ones, K <= 256 into <= 128 columns: at K = N = 512 and 5 120 rows - the 4x4 level of a B = 8 training step - the staged schedules win)
  const int64_t pw_rows = (!p.ln_wsum && p.c0 + p.c1 <= 256 && p.coutp <= 128) ? 10240 : 1024;
  if (pwg_ok || (pw_ok && (pw_mode == 2 || M <= pw_rows || pl.ksplit > 1 || p.res_gn_partial))) {      // see above: where the staged schedules would split K, or few rows
    pl.kind = 3;
    pl.bm = 32;
    pl.bn = 32;
    pl.ksplit = 1;
  }
  return pl;
}

// what the in-launch slab reduction of the Winograd schedule needs apart from a split plan: the plain 32-column workgroup (conv_wino_kernel FUSE), an epilogue
// that stores whole float4 columns of real channels, enough zeroed ticket words.  make_plan asks this BEFORE it takes the finer slices that only pay with the
// in-launch reduction (advisor, round 5: ticket words handed in for a launch the fused path then refuses bought extra slabs and a reduce launch).
bool wino_fuse_geometry_ok(const lfdm_conv_params& p, int bn) {
  static const bool on = [] { const char* e = getenv("LFDM_WINO_FUSE_REDUCE"); return !(e && e[0] == '0'); }();      // (A/B and bit-compare switch: tests)
  if (!on || !p.tile_counters || p.deconv4 || bn != 32 || p.groups > 1 || p.pool2 || p.cout != p.coutp || p.ldo % 4 != 0 ||
      (((uintptr_t)p.out) & 15) != 0 ||
      (p.bias && (((uintptr_t)p.bias) & 15) != 0) ||
      (p.residual && (p.ldr % 4 != 0 || (((uintptr_t)p.residual) & 15) != 0)))
    return false;
  const int64_t ntiles = (int64_t)p.n_img * (p.hq / 2) * (p.wq / 2);
  return (int64_t)p.tile_counters_len >= ((ntiles + 31) / 32) * (p.coutp / 32);
}

// in-launch slab reduction (Winograd F(2x2) schedule, enough zeroed tile counters)
bool splitk_fused(const ConvPlan& pl, const lfdm_conv_params& p) {
  if (pl.ksplit <= 1 || pl.kind != 2 || !wino_fuse_geometry_ok(p, pl.bn)) return false;
  const int64_t M = (int64_t)p.n_img * p.hq * p.wq;
  return (int64_t)pl.ksplit * M * p.coutp * 4 < (1ll << 32) - 64;      // (the slabs go through a buffer descriptor: 32-bit byte offsets)
}

// Balanced Winograd launch (conv_wino.hip, round 6): a launch of exactly 640 (tile, K slice) jobs - every 3x3 convolution of a B = 1 sampler step but
// two - leaves 128 CUs with three workgroups and 128 with two; the launch then runs 512 whole jobs and both halves of the other 128 (768 workgroups =
// three per CU, two whole + one half each).  Needs what the in-launch reduction needs (ticket words, 16-byte-legal epilogue) plus slabs for
// ksplit + *extra_slabs slices; returns the number of whole jobs (512) or 0.  The caller opts in by handing in tile_counters AND a partial buffer
// of lfdm_conv2d_partial_bytes - a binding that sizes nothing for ksplit = 1 plans keeps the plain launch.
int wino_balance(const ConvPlan& pl, const lfdm_conv_params& p, int* extra_slabs) {
  if (extra_slabs) *extra_slabs = 0;
  static const bool on = [] { const char* e = getenv("LFDM_WINO_BALANCE"); return !(e && e[0] == '0'); }();      // (A/B and bit-compare switch)
  static const bool fuse_on = [] { const char* e = getenv("LFDM_WINO_FUSE_REDUCE"); return !(e && e[0] == '0'); }();
  if (!on || !fuse_on || pl.kind != 2 || pl.bn != 32 || !p.tile_counters || p.deconv4 || p.groups > 1 || p.pool2 || p.cout != p.coutp || p.ldo % 4 != 0 ||
      (((uintptr_t)p.out) & 15) != 0 || (p.bias && (((uintptr_t)p.bias) & 15) != 0) ||
      (p.residual && (p.ldr % 4 != 0 || (((uintptr_t)p.residual) & 15) != 0)))
    return 0;
  if (pl.ksplit > 1 && !splitk_fused(pl, p)) return 0;
  const int64_t ntiles = (int64_t)p.n_img * (p.hq / 2) * (p.wq / 2);
  const int64_t T = ((ntiles + 31) / 32) * (p.coutp / 32);
  const int whole = 512, jobs = 640;
  if (T * pl.ksplit != jobs || (int64_t)p.tile_counters_len < T) return 0;
  const int nch = (p.c0 + p.c1) / 16;
  if (nch / pl.ksplit < 2) return 0;                                   // every slice can be halved
  // slice z of the tile at position tl of a slice layer has job id z * T + tl; ids >= whole are halved: the last tile (tl = T - 1) has the most
  const int z_first = T - 1 >= whole ? 0 : (int)((whole - (T - 1) + T - 1) / T);
  const int extra = pl.ksplit - z_first;
  if (extra < 1) return 0;
  const int64_t M = (int64_t)p.n_img * p.hq * p.wq;
  if ((int64_t)(pl.ksplit + extra) * M * p.coutp * 4 >= (1ll << 32) - 64) return 0;
  if (extra_slabs) *extra_slabs = extra;
  return whole;
}

}  // namespace

extern "C" int lfdm_conv2d_plan(const lfdm_conv_params* p, int* tile_rows, int* ksplit) {
  if (!p) return LFDM_EINVAL;
  const ConvPlan pl = make_plan(*p);
  if (tile_rows) *tile_rows = (pl.ksplit > 1 && !splitk_fused(pl, *p)) ? SPLITK_ROWS : pl.bm;   // granularity of gn_partial
  if (ksplit) *ksplit = pl.ksplit;
  return LFDM_OK;
}

extern "C" int lfdm_conv2d_schedule(const lfdm_conv_params* p) {
  if (!p) return LFDM_EINVAL;
  return make_plan(*p).kind;
}

extern "C" size_t lfdm_conv2d_partial_bytes(const lfdm_conv_params* p) {
  if (!p) return 0;
  const ConvPlan pl = make_plan(*p);
  int extra = 0;
  wino_balance(pl, *p, &extra);            // (balanced Winograd launch: one more slab per halved K slice of a tile - also where ksplit is 1)
  if (pl.ksplit <= 1 && extra == 0) return 0;
  const size_t rows = (size_t)p->n_img * p->hq * p->wq;
  // (+ 128: lfdm_conv2d_cl_f32 rounds the slab base up to a 128-byte boundary - the in-launch reduction needs every column tile's slab
  //  rows to be whole cache lines, lfdm_device.h - so the caller's buffer may start anywhere and the plan never depends on its address)
  return ((size_t)(pl.ksplit + extra) * (p->deconv4 ? 4 : 1) * rows * p->coutp + (p->ln_wsum ? (size_t)pl.ksplit * rows * 2 : 0)) * sizeof(float) + 128;
}

extern "C" int lfdm_conv2d_plan_slabs(const lfdm_conv_params* p) {
  if (!p) return LFDM_EINVAL;
  const ConvPlan pl = make_plan(*p);
  int extra = 0;
  wino_balance(pl, *p, &extra);
  return pl.ksplit + extra;
}

extern "C" int lfdm_conv2d_cl_f32(const lfdm_conv_params* pp, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!pp) { lfdm_set_error("conv2d: null params"); return LFDM_EINVAL; }
  lfdm_conv_params p = *pp;
  if (!p.src0 || !p.weight || !p.out || p.c0 <= 0 || p.c1 < 0 || (p.c1 > 0 && !p.src1) ||
      p.n_img <= 0 || p.hq <= 0 || p.wq <= 0 || p.kh <= 0 || p.kw <= 0 || p.cout <= 0 ||
      (int64_t)p.n_img * p.hq * p.wq >= (1ll << 31) || p.coutp < p.cout || (p.coutp % 32) != 0 || p.stride <= 0 || p.out_scale <= 0 || p.ksplit < 0 ||
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
  if (p.deconv4 && (p.kh != 2 || p.kw != 2 || p.stride != 1 || p.out_scale != 2 || p.upsample || p.pad_mode != 0 || p.ln_wsum ||
                    p.gn_partial || p.hq != p.hi || p.wq != p.wi || p.ho != 2 * p.hi || p.wo != 2 * p.wi)) {
    lfdm_set_error("conv2d: deconv4 = the four 2x2 parity convolutions of a ConvTranspose k4 s2 p1 (kh = kw = 2, out_scale 2, no fused norms)");
    return LFDM_EINVAL;
  }
  if (p.groups > 1 && !(p.weight_wino && p.c1 == 0 && p.c0 % (16 * p.groups) == 0 && p.cout % (32 * p.groups) == 0 && p.cout == p.coutp)) {
    lfdm_set_error("conv2d: groups > 1 needs the Winograd form, one source, c0/groups % 16 == 0 and cout/groups % 32 == 0");
    return LFDM_EINVAL;
  }
  const ConvPlan pl = make_plan(p);
  if (p.groups > 1 && pl.kind != 2) { lfdm_set_error("conv2d: groups > 1 is only built for the Winograd schedule (3x3, stride 1, zero pad)"); return LFDM_EINVAL; }
  if (p.pool2 && (pl.kind != 2 || p.residual || p.gn_partial || pp->ksplit > 1 || p.act == LFDM_ACT_NONE)) {
    lfdm_set_error("conv2d: pool2 exists on the Winograd schedule only (3x3, stride 1, zero pad 1, even size, C % 16 == 0): out = (hq/2, wq/2), "
                   "an output activation, no residual / fused GroupNorm / split-K - see lfdm_conv2d_schedule");
    return LFDM_EINVAL;
  }
  if (p.res_gn_partial) {
    const int g = p.res_gn_groups;
    if (pl.kind != 3 || !p.residual || g <= 0 || g > 64 || p.cout % g != 0 || (p.cout / g) % 4 != 0 || p.res_gn_pixels <= 0 || p.res_gn_pixels % 32 != 0 ||
        ((int64_t)p.n_img * p.hq * p.wq) % p.res_gn_pixels != 0 || p.res_gn_nchunk <= 0 || !p.res_gn_gamma || !p.res_gn_beta) {
      lfdm_set_error("conv2d: res_gn_* (GroupNorm + SiLU of the residual in the epilogue) needs the pointwise schedule (see lfdm_conv2d_schedule), a residual, "
                     "cout % groups == 0 with groups of a multiple of 4 channels, pixels per sample % 32 == 0");
      return LFDM_EINVAL;
    }
  }
  if (p.gn_in_partial || p.defer_reduce) {
    lfdm_set_error("conv2d: gn_in_* / defer_reduce are reserved since ABI 12 (the variants were measured slower and removed): pass NULL / 0");
    return LFDM_EINVAL;
  }
  p.ksplit = pl.ksplit;
  if (p.ksplit > 1 && !p.partial) { lfdm_set_error("conv2d: split-K needs the partial buffer (lfdm_conv2d_partial_bytes)"); return LFDM_EWORKSPACE; }
  if (p.partial) p.partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(p.partial) + 127) & ~(uintptr_t)127);   // (slack: partial_bytes)
  const int64_t M = (int64_t)p.n_img * p.hq * p.wq;
  if (p.gn_partial) {
    const int cg = p.gn_groups > 0 ? p.cout / p.gn_groups : 0;
    const bool fused = splitk_fused(pl, p);
    const int rows = (p.ksplit > 1 && !fused) ? SPLITK_ROWS : pl.bm;
    const bool ok = p.gn_groups > 0 && p.cout % p.gn_groups == 0 && cg % 4 == 0 && p.gn_pixels > 0 &&
                    p.gn_pixels % rows == 0 && p.cout % 4 == 0 && p.ldo % 4 == 0 &&
                    (((uintptr_t)p.out) & 15) == 0 &&
                    ((p.ksplit > 1 && !fused) ? (256 % (p.coutp / 4) == 0 && p.cout == p.coutp)
                                              : (pl.bn % cg == 0 || (pl.kind == 2 && cg % pl.bn == 0)));      // (Winograd: a group may span column tiles)
    if (!ok) {
      lfdm_set_error("conv2d: fused GroupNorm statistics need pixels % tile_rows == 0 and a group size dividing "
                     "the column tile (see lfdm_conv2d_plan)");
      return LFDM_EINVAL;
    }
  }
  if (p.ln_wsum) {
    if (!pl.fast || !pl.simple || p.kh != 1 || p.kw != 1 || p.c1 != 0 || p.stride != 1 || p.cout % 4 != 0 ||
        p.ldo % 4 != 0 || (((uintptr_t)p.out) & 15) != 0 || (((uintptr_t)p.ln_wsum) & 15) != 0) {
      lfdm_set_error("conv2d: fused LayerNorm needs a 1x1 convolution over one source with C % 32 == 0");
      return LFDM_EINVAL;
    }
  }
  int rc;
  if (pl.kind == 1) {
    rc = lfdm_conv_ksw_launch(p, pl.bn, stream);
  } else if (pl.kind == 2) {
    const int bal = p.partial ? wino_balance(pl, p, nullptr) : 0;
    rc = lfdm_conv_wino_launch(p, pl.bn, bal > 0 || splitk_fused(pl, p), bal, stream);
  } else if (pl.kind == 3) {
    rc = lfdm_conv_pw_launch(p, stream);
  } else if (pl.kind == 4) {
    rc = lfdm_conv_wino4_launch(p, stream);
  } else {
    const dim3 grid((unsigned)((M + pl.bm - 1) / pl.bm), (unsigned)((p.coutp + pl.bn - 1) / pl.bn), p.ksplit * (p.deconv4 ? 4 : 1));
    if (pl.bm == 128 && pl.bn == 128) launch_conv<128, 128>(p, pl.fast, pl.simple, grid, stream);
    else if (pl.bm == 64) launch_conv<64, 64>(p, pl.fast, pl.simple, grid, stream);
    else launch_conv<128, 64>(p, pl.fast, pl.simple, grid, stream);
    rc = lfdm_check_launch("conv_igemm");
  }
  if (rc) return rc;
  if (p.ksplit > 1 && !splitk_fused(pl, p)) {
    const dim3 rgrid((unsigned)((M + SPLITK_ROWS - 1) / SPLITK_ROWS), (unsigned)splitk_col_chunks(p, M), p.deconv4 ? 4 : 1);
    static const bool vec_on = [] { const char* e = lfdm_knob("LFDM_REDUCE_VEC"); return !(e && e[0] == '0'); }();      // (A/B knob)
    const bool vec = vec_on && p.ksplit <= 8 && !p.deconv4 && !p.ln_wsum && p.cout == p.coutp && p.ldo % 4 == 0 && (((uintptr_t)p.out) & 15) == 0 &&
                     p.out_scale == 1 && p.out_off_y == 0 && p.out_off_x == 0 && p.ho == p.hq && p.wo == p.wq &&
                     (!p.bias || (((uintptr_t)p.bias) & 15) == 0) && (!p.residual || (p.ldr % 4 == 0 && (((uintptr_t)p.residual) & 15) == 0)) &&
                     (((uintptr_t)p.partial) & 15) == 0;
    if (vec) {
      switch (p.ksplit) {
        case 2: LFDM_LAUNCH(conv_splitk_reduce_vec_kernel<2>, rgrid, dim3(256), 0, stream, p); break;
        case 3: LFDM_LAUNCH(conv_splitk_reduce_vec_kernel<3>, rgrid, dim3(256), 0, stream, p); break;
        case 4: LFDM_LAUNCH(conv_splitk_reduce_vec_kernel<4>, rgrid, dim3(256), 0, stream, p); break;
        case 5: LFDM_LAUNCH(conv_splitk_reduce_vec_kernel<5>, rgrid, dim3(256), 0, stream, p); break;
        case 6: LFDM_LAUNCH(conv_splitk_reduce_vec_kernel<6>, rgrid, dim3(256), 0, stream, p); break;
        case 7: LFDM_LAUNCH(conv_splitk_reduce_vec_kernel<7>, rgrid, dim3(256), 0, stream, p); break;
        default: LFDM_LAUNCH(conv_splitk_reduce_vec_kernel<8>, rgrid, dim3(256), 0, stream, p); break;
      }
    } else {
      LFDM_LAUNCH(conv_splitk_reduce_kernel, rgrid, dim3(256), 0, stream, p);
    }
    rc = lfdm_check_launch("conv_splitk_reduce");
  }
  return rc;
}
