// Pointwise (1x1, stride 1) convolution / Linear on channels-last rows as a register-operand fp32-MFMA GEMM
// (include/lfdm_hip.h: lfdm_conv2d_cl_f32 picks it for the to_qkv / to_out / res_conv projections, schedule 3).
//
// Why a third direct schedule: in the B = 1 sampler these projections have K = 64 ... 1024 and M = 640 ... 10 240 rows.  The
// LDS-staged schedules (conv_igemm / conv_ksw) need ~10 us of fixed cost per launch (index tables, staging prologue, barrier
// per 32-wide chunk) and, below 256 output tiles, a split-K slab round trip plus a reduce launch - 15-21 us for 0.3-0.5 GFLOP.
// Here nothing is staged: with the contraction order free, MFMA k-step 4q+e of lane half kh takes k = 8q + 4kh + e for BOTH
// operands, so a lane's A fragment is 16 contiguous bytes of its row of x and its B fragment 16 contiguous bytes of its
// output channel's packed filter row ([chunk][cout][32]: k contiguous) - two buffer loads feed four v_mfma_f32_32x32x2_f32.
// A workgroup owns 32 rows x (4/KW * TN * 32) columns; its four wavefronts split K (KW of them) and the column tiles, every
// load of a 32-deep K group is issued one group ahead, and there is no barrier before the epilogue.  The K slices are summed
// through LDS (18 KB), which also turns the accumulator layout into float4 row segments for bias / LayerNorm fold / residual /
// activation.  No split-K across workgroups, no reduce launch: 32-row tiles give 20 x 16 workgroups even at M = 640.
// Workgroups are dealt so that the column tiles of one row tile share an XCD (x is read into ONE L2).
#include <stdlib.h>

#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

// GATHER (round 6): the same register-operand GEMM for a kh x kw / any-stride / zero-padded convolution over ONE source with C % 32 == 0 - the
// Downsample convolutions (4x4, stride 2) of a B = 1 step, which the K-split-across-waves schedule ran as 23 us + a 5 us reduce launch for 1.3 GFLOP.
// k = tap * C + c in the filter's own order; a 32-deep K group lies inside one tap, so a group's A fragments are 16 contiguous bytes of the lane's
// output pixel's input pixel (iy0 + ky, ix0 + kx) - a uniform (ky, kx) per group, a per-lane validity bit per tap (out of the image = out-of-range
// offset = zeros).  Nothing else changes: no im2col, no LDS staging, no split-K slabs.
template <int TN, int KW, bool LN, bool GATHER = false>
__global__ __launch_bounds__(256) void conv_pw_kernel(lfdm_conv_params p, int gx, int ny) {
  constexpr int NW = 4 / KW;           // wavefronts along N
  constexpr int LD = 36;               // scratch row stride (floats): conflict-free b128 reads
  __shared__ __attribute__((aligned(16))) float scratch[4 * 32 * LD];
  __shared__ float s_ln[2][4][32];     // fused LayerNorm: [sum | sum of squares][wave][row]
  constexpr int BN = (4 / KW) * TN * 32;                         // columns of this workgroup
  __shared__ __attribute__((aligned(16))) float s_ga[BN], s_gb[BN];      // res_gn_*: A[c] | B[c] of the workgroup's columns
  __shared__ float s_gstat[2][64];

  const int tid = threadIdx.x, lane = tid & 63;
  // the wave index is uniform, but derived from threadIdx the compiler cannot know it: every buffer load whose descriptor or
  // chunk offset depends on it was wrapped in a per-load "waterfall" loop (cdna_hip_programming.md T20) - readfirstlane makes it scalar
  const int wave = lfdm_uniform(tid >> 6);
  const int wk = wave % KW, wn = wave / KW;
  // block id -> (row tile, column tile): ids round-robin over the 8 XCDs, an XCD walks the column tiles of one row tile
  const int id = blockIdx.x, slot = id >> 3;
  const int bx = (id & 7) + 8 * (slot / ny), by = slot % ny;
  if (bx >= gx) return;
  const int M = p.n_img * p.hq * p.wq;
  const int m0 = bx * 32;
  const int n0 = by * (NW * TN * 32);
  const int K = GATHER ? p.kh * p.kw * p.c0 : p.c0 + p.c1;
  const int kw_len = K / KW;           // this wave's K slice: [wk*kw_len, +kw_len), a multiple of 32
  const int kbeg = wk * kw_len;
  const int ng = kw_len >> 5;
  const int l31 = lane & 31, kh = lane >> 5;

  const int64_t in_rows = GATHER ? (int64_t)p.n_img * p.hi * p.wi : (int64_t)M;
  const lfdm_buf buf0 = lfdm_make_buf(p.src0, (uint32_t)(((in_rows - 1) * p.ld0 + p.c0) * 4));
  const lfdm_buf buf1 = p.c1 > 0 ? lfdm_make_buf(p.src1, (uint32_t)(((int64_t)(M - 1) * p.ld1 + p.c1) * 4)) : buf0;
  // weight_pw (round 4): the same filter in MFMA-operand order [K/32][coutp/32][4 u][64 lanes = 32*kh + column][4] - one contiguous 1 KB
  // per fragment load instead of a 16-byte piece of each of 32 columns' 128-byte rows (four times the L1 line look-ups per byte)
  const bool wpk = p.weight_pw != nullptr;
  const lfdm_buf bufw = lfdm_make_buf(wpk ? p.weight_pw : p.weight, (uint32_t)((int64_t)(K >> 5) * p.coutp * 128));
  const uint32_t ustride = wpk ? 1024u : 32u;
  const int row = m0 + l31;
  const bool row_ok = row < M;
  // GATHER: this lane's output pixel -> the corner of its input window and one validity bit per tap
  int g_pix0 = 0;
  unsigned g_valid = 0;
  if (GATHER && row_ok) {
    const int ox = row % p.wq, t1 = row / p.wq;
    const int oy = t1 % p.hq, n = t1 / p.hq;
    const int iy0 = oy * p.stride - p.pad_y, ix0 = ox * p.stride - p.pad_x;
    g_pix0 = (n * p.hi + iy0) * p.wi + ix0;
    for (int ky = 0; ky < p.kh; ++ky)
      for (int kx = 0; kx < p.kw; ++kx)
        if (iy0 + ky >= 0 && iy0 + ky < p.hi && ix0 + kx >= 0 && ix0 + kx < p.wi) g_valid |= 1u << (ky * p.kw + kx);
  }
  const uint32_t a_off0 = row_ok ? ((uint32_t)row * (uint32_t)p.ld0 + 4u * kh) * 4u : LFDM_BUF_OOB;
  const uint32_t a_off1 = row_ok ? ((uint32_t)row * (uint32_t)p.ld1 + 4u * kh) * 4u : LFDM_BUF_OOB;
  uint32_t b_off[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + l31;
    b_off[j] = n >= p.coutp ? LFDM_BUF_OOB : wpk ? (uint32_t)(n >> 5) * 4096u + (uint32_t)lane * 16u : ((uint32_t)n * 32u + 4u * kh) * 4u;
  }
  const uint32_t wgroup_bytes = (uint32_t)p.coutp * 128u;

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float ln_s = 0.f, ln_q = 0.f;

  float4 ra[4], rb[4][TN];
  auto fetch = [&](int g) {
    const int kk = kbeg + 32 * g;
    const bool second = !GATHER && kk >= p.c0;
    const lfdm_buf buf = second ? buf1 : buf0;
    uint32_t abase = second ? a_off1 + (uint32_t)(kk - p.c0) * 4u : a_off0 + (uint32_t)kk * 4u;
    bool a_ok = row_ok;
    if (GATHER) {
      const int tap = kk / p.c0, cch = kk - tap * p.c0;             // (uniform)
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      a_ok = (g_valid >> tap) & 1u;
      abase = ((uint32_t)(g_pix0 + ky * p.wi + kx) * (uint32_t)p.ld0 + (uint32_t)cch + 4u * kh) * 4u;
    }
    const uint32_t wbase = (uint32_t)(kk >> 5) * wgroup_bytes;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ra[u] = lfdm_buf_load_f4(buf, a_ok ? abase + 32u * u : LFDM_BUF_OOB);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        rb[u][j] = lfdm_buf_load_f4(bufw, b_off[j] == LFDM_BUF_OOB ? LFDM_BUF_OOB : wbase + b_off[j] + ustride * u);
    }
  };

  fetch(0);
  if (p.res_gn_partial) {
    // (round 6) the residual is a RAW convolution output: its GroupNorm + SiLU is applied in the epilogue.  Statistics of this row tile's sample,
    // merged as gn_apply_kernel does (32 lanes walk one group's chunks, k ascending, double) while the first operand group is in flight; only the
    // groups that cover this workgroup's columns are merged.
    const int groups = p.res_gn_groups, nchunk = p.res_gn_nchunk, cg = p.cout / groups;
    const int b = m0 / p.res_gn_pixels;
    const int g_lo = n0 / cg;
    const int g_hi = (((n0 + BN < p.cout ? n0 + BN : p.cout) - 1) / cg);
    const int sub = tid & 31;
    // (this thread's gamma / beta are requested before the merge, not behind its barrier: BN <= 256 columns, one per thread)
    const int pcol = n0 + tid;
    const bool pcol_ok = tid < BN && pcol < p.cout;
    const float pre_gamma = p.res_gn_gamma[pcol_ok ? pcol : 0], pre_beta = p.res_gn_beta[pcol_ok ? pcol : 0];
    for (int g0 = g_lo; g0 <= g_hi; g0 += 8) {
      const int g = g0 + (tid >> 5);
      double sm = 0.0, sq = 0.0;
      if (g <= g_hi) {
        const float2* src = reinterpret_cast<const float2*>(p.res_gn_partial) + ((int64_t)b * nchunk) * groups + g;
        int k = sub;
        for (; k + 96 < nchunk; k += 128) {
          const float2 v0 = src[(int64_t)k * groups], v1 = src[(int64_t)(k + 32) * groups];
          const float2 v2 = src[(int64_t)(k + 64) * groups], v3 = src[(int64_t)(k + 96) * groups];
          sm += (double)v0.x; sq += (double)v0.y;
          sm += (double)v1.x; sq += (double)v1.y;
          sm += (double)v2.x; sq += (double)v2.y;
          sm += (double)v3.x; sq += (double)v3.y;
        }
        for (; k < nchunk; k += 32) {
          const float2 v = src[(int64_t)k * groups];
          sm += (double)v.x;
          sq += (double)v.y;
        }
      }
      for (int msk = 16; msk >= 1; msk >>= 1) {
        sm += __shfl_xor(sm, msk);
        sq += __shfl_xor(sq, msk);
      }
      if (g <= g_hi && sub == 0) {
        const double n = (double)p.res_gn_pixels * (double)cg;
        const double mean = sm / n;
        double var = sq / n - mean * mean;
        if (var < 0.0) var = 0.0;
        s_gstat[0][g - g_lo] = (float)mean;
        s_gstat[1][g - g_lo] = (float)(1.0 / sqrt(var + (double)p.res_gn_eps));
      }
    }
    __syncthreads();
    if (pcol_ok) {
      const int g = pcol / cg - g_lo;
      const float a = s_gstat[1][g] * pre_gamma;
      s_ga[tid] = a;
      s_gb[tid] = pre_beta - s_gstat[0][g] * a;
    }
    // (the table is read after the epilogue's first barrier)
  }
  for (int g = 0; g < ng; ++g) {
    float4 a[4], b[4][TN];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = ra[u];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[u][j] = rb[u][j];
    }
    fetch(g + 1 < ng ? g + 1 : g);       // clamped (the last group is re-read): no branch around the loads
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (LN) {
        ln_s += (a[u].x + a[u].y) + (a[u].z + a[u].w);
        ln_q += (a[u].x * a[u].x + a[u].y * a[u].y) + (a[u].z * a[u].z + a[u].w * a[u].w);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[j] = mfma_32x32x2(a[u].x, b[u][j].x, acc[j]);
        acc[j] = mfma_32x32x2(a[u].y, b[u][j].y, acc[j]);
        acc[j] = mfma_32x32x2(a[u].z, b[u][j].z, acc[j]);
        acc[j] = mfma_32x32x2(a[u].w, b[u][j].w, acc[j]);
      }
    }
  }

  if (LN) {
    ln_s += __shfl_xor(ln_s, 32);        // the two lane halves hold the two k quads of every group
    ln_q += __shfl_xor(ln_q, 32);
    if (lane < 32) {
      s_ln[0][wave][l31] = ln_s;
      s_ln[1][wave][l31] = ln_q;
    }
  }

  // ---- K-slice sum + epilogue: one 32x32 tile per pass, thread = (row, float4 of columns) ----
  const int trow = tid >> 3, c4 = tid & 7;
  const int orow = m0 + trow;
  const bool vec_row = orow < M;
  float mean = 0.f, rstd = 1.f;
  // bias / LayerNorm column sums / residual rows of EVERY pass are requested here, before the accumulators go through LDS: loaded
  // inside the pass loop each of them was a load -> wait -> store round trip (~1 us per pass on a launch that should take 5)
  constexpr int NP = NW * TN;
  float4 pre_b[NP], pre_w[NP], pre_r[NP];
#pragma unroll
  for (int g = 0; g < NW; ++g)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (g * TN + j) * 32 + 4 * c4;
      const bool ok = vec_row && col < p.cout;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      pre_b[g * TN + j] = (ok && p.bias) ? *reinterpret_cast<const float4*>(p.bias + col) : z;
      pre_w[g * TN + j] = (LN && ok) ? *reinterpret_cast<const float4*>(p.ln_wsum + col) : z;
      pre_r[g * TN + j] = (ok && p.residual) ? *reinterpret_cast<const float4*>(p.residual + (int64_t)orow * p.ldr + col) : z;
    }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    if (j > 0) __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r)
      scratch[wave * (32 * LD) + ((r & 3) + 8 * (r >> 2) + 4 * kh) * LD + l31] = acc[j][r];
    __syncthreads();
    if (LN && j == 0) {
      float sv = 0.f, qv = 0.f;
#pragma unroll
      for (int k = 0; k < KW; ++k) {     // the waves of column group 0 cover K exactly once
        sv += s_ln[0][k][trow];
        qv += s_ln[1][k][trow];
      }
      const float inv_c = 1.0f / (float)K;
      mean = sv * inv_c;
      float var = qv * inv_c - mean * mean;
      if (var < 0.f) var = 0.f;
      rstd = 1.0f / sqrtf(var + p.ln_eps);
    }
#pragma unroll
    for (int g = 0; g < NW; ++g) {
      float4 v = *reinterpret_cast<const float4*>(scratch + (g * KW) * (32 * LD) + trow * LD + 4 * c4);
#pragma unroll
      for (int k = 1; k < KW; ++k) {
        const float4 u = *reinterpret_cast<const float4*>(scratch + (g * KW + k) * (32 * LD) + trow * LD + 4 * c4);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      const int col = n0 + (g * TN + j) * 32 + 4 * c4;
      if (!vec_row || col >= p.cout) continue;
      if (LN) {                          // y = rstd * (x.W' - mean * sum_c W')
        const float4 ws = pre_w[g * TN + j];
        v.x = rstd * (v.x - mean * ws.x); v.y = rstd * (v.y - mean * ws.y);
        v.z = rstd * (v.z - mean * ws.z); v.w = rstd * (v.w - mean * ws.w);
      }
      {
        const float4 bb = pre_b[g * TN + j];
        float4 rr = pre_r[g * TN + j];
        if (p.res_gn_partial) {
          const float4 ga = *reinterpret_cast<const float4*>(s_ga + col - n0), gb = *reinterpret_cast<const float4*>(s_gb + col - n0);
          rr.x = siluf_(fmaf(rr.x, ga.x, gb.x)); rr.y = siluf_(fmaf(rr.y, ga.y, gb.y));
          rr.z = siluf_(fmaf(rr.z, ga.z, gb.z)); rr.w = siluf_(fmaf(rr.w, ga.w, gb.w));
        }
        v.x += bb.x + rr.x; v.y += bb.y + rr.y; v.z += bb.z + rr.z; v.w += bb.w + rr.w;
      }
      if (p.act) {
        v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act);
        v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
      }
      *reinterpret_cast<float4*>(p.out + (int64_t)orow * p.ldo + col) = v;
    }
  }
}

template <int TN, int KW>
void launch_pw(const lfdm_conv_params& p, int gx, int ny, hipStream_t stream) {
  const dim3 grid((unsigned)(((gx + 7) / 8) * 8 * ny)), block(256);
  if (p.kh * p.kw > 1) {                 // (the gather form: K over all four waves, one or two column tiles per wave - lfdm_conv_pw_shape)
    if constexpr (TN <= 2 && KW == 4) LFDM_LAUNCH((conv_pw_kernel<TN, 4, false, true>), grid, block, 0, stream, p, gx, ny);
    return;
  }
  if (p.ln_wsum) LFDM_LAUNCH((conv_pw_kernel<TN, KW, true>), grid, block, 0, stream, p, gx, ny);
  else LFDM_LAUNCH((conv_pw_kernel<TN, KW, false>), grid, block, 0, stream, p, gx, ny);
}

int env_int(const char* name, int dflt) {      // (sweep constants: read in `--knobs` builds only - lfdm_knob)
  const char* e = lfdm_knob(name);
  return e ? atoi(e) : dflt;
}

}  // namespace

// Tile shape for a geometry the caller has already checked (1x1, stride 1, C % 32 == 0, float4-legal epilogue):
// K across KW wavefronts when the slices stay multiples of 32; the widest column tile that still leaves >= 256 workgroups.
void lfdm_conv_pw_shape(const lfdm_conv_params& p, int* tn, int* kw) {
  const int K = p.kh * p.kw > 1 ? p.kh * p.kw * p.c0 : p.c0 + p.c1;
  const int64_t M = (int64_t)p.n_img * p.hq * p.wq;
  if (p.kh * p.kw > 1) {                 // gather form (the plan has checked K % 128 == 0): 64-column workgroups while >= 256 of them remain
    *tn = (p.coutp % 64 == 0 && ((M + 31) / 32) * (p.coutp / 64) >= 256) ? 2 : 1;
    *kw = 4;
    return;
  }
  // (two sources: the source is selected per 32-deep K group and c0 % 32 == 0, so a slice only has to be a multiple of 32)
  int k = K % 128 == 0 ? 4 : K % 64 == 0 ? 2 : 1;
  const int fk = env_int("LFDM_PW_KW", 0);
  if ((fk == 1 || fk == 2 || fk == 4) && K % (32 * fk) == 0) k = fk;
  // measured (tools/bench_pw.py): 96-column tiles for the 768-column to_qkv of the 4x4 level (160 workgroups of 3 accumulator
  // chains beat 480 of one), one 32-column tile per wave everywhere else
  int t = (M <= 1024 && p.coutp >= 768 && p.coutp % ((4 / k) * 96) == 0) ? 3 : 1;
  const int ft = env_int("LFDM_PW_TN", 0);
  if (ft >= 1 && ft <= 3) t = ft;
  *tn = t;
  *kw = k;
}

int lfdm_conv_pw_launch(const lfdm_conv_params& p, hipStream_t stream) {
  int tn, kw;
  lfdm_conv_pw_shape(p, &tn, &kw);
  const int64_t M = (int64_t)p.n_img * p.hq * p.wq;
  const int gx = (int)((M + 31) / 32);
  const int bn = (4 / kw) * tn * 32;
  const int ny = (p.coutp + bn - 1) / bn;
#define LFDM_PW_CASE(T, W) if (tn == T && kw == W) launch_pw<T, W>(p, gx, ny, stream)
  LFDM_PW_CASE(1, 4); else LFDM_PW_CASE(2, 4); else LFDM_PW_CASE(3, 4);
  else LFDM_PW_CASE(1, 2); else LFDM_PW_CASE(2, 2); else LFDM_PW_CASE(3, 2);
  else LFDM_PW_CASE(1, 1); else LFDM_PW_CASE(2, 1); else LFDM_PW_CASE(3, 1);
#undef LFDM_PW_CASE
  return lfdm_check_launch("conv_pw");
}
