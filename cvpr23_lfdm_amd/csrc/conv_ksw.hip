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

// ACT = false: no activation code in the (fully unrolled, executed-once) epilogue - the UNet's convolutions have
// none, and the unrolled expf / division sequences of the generic epilogue tripled its instruction footprint.
template <int TM, int TN, bool ACT>
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
#ifdef LFDM_KSW_TIMING
  // probe build only (tools/probe_ksw_phases.py): phase time stamps of every workgroup go to p.partial
  unsigned long long tstamp[5];
  tstamp[0] = __builtin_readcyclecounter();
#endif
  const int hqwq = p.hq * p.wq;
  const int64_t M = (int64_t)p.n_img * hqwq;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int cin = p.c0 + p.c1;
  const int ntaps = p.kh * p.kw;
  const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
  int zk = blockIdx.z;                             // K slice of this workgroup
  if (p.deconv4) {                                  // grid z = parity * ksplit + slice: the four 2x2 parity problems of a
    const int par = zk / ksplit;                    // ConvTranspose k4 s2 p1 in one launch (p is this workgroup's copy)
    zk -= par * ksplit;
    p.pad_y = 1 - (par >> 1);
    p.pad_x = 1 - (par & 1);
    p.out_off_y = par >> 1;
    p.out_off_x = par & 1;
    p.weight += (int64_t)par * (ntaps * (cin / 32)) * p.coutp * 32;
  }

  for (int r = tid; r < BM; r += 256) {
    const int64_t m = m0 + r;
    int img = -1, qy = 0, qx = 0;
    unsigned long long mask = 0ull;
    if (m < M) {
      const int mi = (int)m;                       // M < 2^31 (checked on the host)
      img = mi / hqwq;
      const int rem = mi - img * hqwq;
      qy = rem / p.wq;
      qx = rem - qy * p.wq;
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
  const int kc_begin = (int)((int64_t)nchunks_all * zk / ksplit);
  const int kc_end = (int)((int64_t)nchunks_all * (zk + 1) / ksplit);
  const int nk = kc_end - kc_begin;

  // Branch-free main loop: every load goes through a buffer descriptor and an invalid tap is an
  // out-of-range offset (returns 0); chunk indices are clamped instead of guarded, so the loop body is
  // one basic block and the 16*TM*TN accumulator registers stay where they are.
  const int cq = tid & 7;
  const int64_t in_rows = (int64_t)p.n_img * p.hi * p.wi;
  const lfdm_buf buf0 = lfdm_make_buf(p.src0, (uint32_t)(((in_rows - 1) * p.ld0 + p.c0) * 4));
  const lfdm_buf buf1 = p.c1 > 0 ? lfdm_make_buf(p.src1, (uint32_t)(((in_rows - 1) * p.ld1 + p.c1) * 4)) : buf0;
  const lfdm_buf bufw = lfdm_make_buf(p.weight, (uint32_t)((int64_t)nchunks_all * p.coutp * BK * 4));
  uint32_t a_off0[A_F4], a_off1[A_F4];
  unsigned long long a_mask[A_F4];
#pragma unroll
  for (int i = 0; i < A_F4; ++i) {
    const int r = (tid >> 3) + 32 * i;
    a_off0[i] = ((uint32_t)s_pix[r] * (uint32_t)p.ld0 + 4u * cq) * 4u;     // wraps; valid taps un-wrap it
    a_off1[i] = ((uint32_t)s_pix[r] * (uint32_t)p.ld1 + 4u * cq) * 4u;
    a_mask[i] = s_mask[r];
  }
  uint32_t b_off[B_F4];
#pragma unroll
  for (int i = 0; i < B_F4; ++i) {
    const int f = tid + 256 * i;
    b_off[i] = (n0 + (f >> 3) < p.coutp) ? (uint32_t)(((n0 + (f >> 3)) * BK + 4 * (f & 7)) * 4) : LFDM_BUF_OOB;
  }
  float4 ra[A_F4], rb[B_F4];
  const int cpt = cin / BK;
  const uint32_t wchunk_bytes = (uint32_t)p.coutp * BK * 4;

  auto fetch = [&](int kc) {
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
      ra[i] = lfdm_buf_load_f4(buf, off);
    }
    const uint32_t wbase = (uint32_t)kc * wchunk_bytes;
#pragma unroll
    for (int i = 0; i < B_F4; ++i) rb[i] = lfdm_buf_load_f4(bufw, b_off[i] == LFDM_BUF_OOB ? LFDM_BUF_OOB : wbase + b_off[i]);
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

#ifdef LFDM_KSW_TIMING
  tstamp[1] = __builtin_readcyclecounter();
#endif
  fetch(kc_begin);                                  // nk >= 1 (host clamps ksplit to the chunk count)
  stage(0);
  __syncthreads();
  fetch(kc_begin + (nk > 1 ? 1 : 0));
#ifdef LFDM_KSW_TIMING
  tstamp[2] = __builtin_readcyclecounter();
#endif

  const int koff = 8 * wave + 4 * (lane >> 5);   // this wave's k slice of a chunk, this lane half's quad
  const int l31 = lane & 31;
  for (int c = 0; c < nk; ++c) {
    const int cur = c & 1;
    stage(cur ^ 1);                                 // registers hold chunk min(c+1, nk-1); idle buffer
#ifndef LFDM_PROBE_NOFETCH
    {
      const int nxt = c + 2 < nk ? c + 2 : nk - 1;
      fetch(kc_begin + nxt);
    }
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

#ifdef LFDM_KSW_TIMING
  tstamp[3] = __builtin_readcyclecounter();
#endif
  // ---- cross-wave reduction + epilogue, one 32x32 tile at a time ----
  float* const scratch = smem;                     // [4 waves][32][LD]
  const int trow = tid >> 3, c4 = tid & 7;
  // (the in-launch slab reduction of the Winograd schedule was built here too - rounds 2 and 5 - and measured 0.9 ms per video SLOWER than
  //  this schedule's four reduce launches: 4-40 workgroups whose last arriver walks up to ten tiles serially; removed in round 6)
  // (the host only selects this kernel when float4 epilogue accesses are legal)
  float gs[TN][4], gq[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) gs[j][e] = gq[j][e] = 0.f;

  // bias depends only on the column: ONE load per column tile before the rounds; the residual float4 of round t+1 is
  // requested while round t is reduced.  (Per-round bias -> wait -> residual -> wait chains cost 0.6 us x 10 rounds.)
  float4 bias4[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int colbase = n0 + 32 * j + 4 * c4;
    bias4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && ksplit == 1 && colbase < p.cout) bias4[j] = *reinterpret_cast<const float4*>(p.bias + colbase);
  }
  auto out_row = [&](int i) -> int64_t {
    const int row = 32 * i + trow;
    const int oy = s_qy[row] * p.out_scale + p.out_off_y;
    const int ox = s_qx[row] * p.out_scale + p.out_off_x;
    return ((int64_t)s_img[row] * p.ho + oy) * p.wo + ox;
  };
  auto load_res = [&](int i, int j) -> float4 {
    const int colbase = n0 + 32 * j + 4 * c4;
    if (p.residual && ksplit == 1 && s_img[32 * i + trow] >= 0 && colbase < p.cout)
      return *reinterpret_cast<const float4*>(p.residual + out_row(i) * p.ldr + colbase);
    return make_float4(0.f, 0.f, 0.f, 0.f);
  };
  float4 res_next = ksplit == 1 ? load_res(0, 0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        scratch[wave * (32 * LD) + ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LD + l31] = acc[i][j][r];
      const float4 rr = res_next;
      {
        const int t = i * TN + j + 1;
        if (t < TM * TN && ksplit == 1) res_next = load_res(t / TN, t % TN);
      }
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
          if (colbase < p.coutp) {
            *reinterpret_cast<float4*>(p.partial + ((int64_t)blockIdx.z * M + (m0 + row)) * p.coutp + colbase) = v;
          }
        } else if (colbase < p.cout) {
          const int64_t orow = out_row(i);
          v.x += bias4[j].x; v.y += bias4[j].y; v.z += bias4[j].z; v.w += bias4[j].w;
          if (p.gn_partial) {
            gs[j][0] += v.x; gs[j][1] += v.y; gs[j][2] += v.z; gs[j][3] += v.w;
            gq[j][0] += v.x * v.x; gq[j][1] += v.y * v.y; gq[j][2] += v.z * v.z; gq[j][3] += v.w * v.w;
          }
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
          if (ACT) {
            v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act);
            v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
          }
#ifdef LFDM_PROBE_NOSTORE
          if (v.x == 123456.789f)
#endif
          *reinterpret_cast<float4*>(p.out + orow * p.ldo + colbase) = v;
        }
      }
      __syncthreads();
    }
  }


#ifdef LFDM_KSW_TIMING
  tstamp[4] = __builtin_readcyclecounter();
  if (tid == 0 && p.partial && ksplit == 1) {
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.partial) + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 5;
    for (int i = 0; i < 5; ++i) dst[i] = tstamp[i];
  }
#endif
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
  const dim3 grid((unsigned)((M + 159) / 160), (unsigned)((p.coutp + bn - 1) / bn), (p.ksplit > 1 ? p.ksplit : 1) * (p.deconv4 ? 4 : 1));
  const bool act = p.act != LFDM_ACT_NONE;
  if (bn == 64) {
    if (act) LFDM_LAUNCH((conv_ksw_kernel<5, 2, true>), grid, dim3(256), 0, stream, p);
    else LFDM_LAUNCH((conv_ksw_kernel<5, 2, false>), grid, dim3(256), 0, stream, p);
  } else {
    if (act) LFDM_LAUNCH((conv_ksw_kernel<5, 1, true>), grid, dim3(256), 0, stream, p);
    else LFDM_LAUNCH((conv_ksw_kernel<5, 1, false>), grid, dim3(256), 0, stream, p);
  }
  return lfdm_check_launch("conv_ksw");
}
