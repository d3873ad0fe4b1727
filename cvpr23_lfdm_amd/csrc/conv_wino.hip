// Winograd F(2x2, 3x3) schedule of the 3x3 / stride-1 / zero-padded convolution on fp32 MFMA
// (include/lfdm_hip.h: lfdm_conv2d_cl_f32 with lfdm_conv_params.weight_wino; LFDM_WINO=0 forces the direct kernels).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A   per 4x4 input patch d / 2x2 output tile Y and (c_in, c_out) pair:
// 16 independent "frequency positions", each an ordinary GEMM over c_in: M[pos][tile][co] = sum_ci V[pos][tile][ci] U[pos][ci][co]
// -> 16/36 of the multiplications of the direct form.  One workgroup owns 32 output tiles (128 pixels) x 32 output channels;
// per 16-channel chunk 128 threads transform the patches (B^T d B, float4 over channels) into LDS, then wave w runs the
// MFMAs of positions 4w..4w+3 (A operand = V from LDS, B operand = the pre-transformed weights straight from global memory in
// operand order, prefetched one chunk ahead); the four waves' accumulators meet in LDS for the output transform A^T M A, which
// feeds the same epilogue as the direct kernels (bias, GroupNorm partial sums, residual, activation) or the split-K slabs.
#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

constexpr int WT = 32;            // tiles per workgroup
constexpr int WN = 32;            // output channels per workgroup
constexpr int WKC = 16;           // input channels per chunk
constexpr int LDV = WKC + 4;      // LDS row stride of V
constexpr int LDM = WN + 1;       // LDS row stride of M in the output transform

template <bool ACT>
__global__ __launch_bounds__(256) void conv_wino_kernel(lfdm_conv_params p) {
  __shared__ __attribute__((aligned(16))) float smem[16 * WT * LDM];      // >= 16*WT*LDV: V during the loop, M in the epilogue
  __shared__ int s_n[WT], s_ty[WT], s_tx[WT];
  __shared__ float s_gn[2][8][WN];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int th = p.hi >> 1, tw = p.wi >> 1;                 // tiles per image (H, W even: host check)
  const int64_t ntiles = (int64_t)p.n_img * th * tw;
  const int64_t t0 = (int64_t)blockIdx.x * WT;
  const int n0 = blockIdx.y * WN;
  const int cin = p.c0 + p.c1;
  const int nchunks_all = cin / WKC;
  const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
  const int kc_begin = (int)((int64_t)nchunks_all * blockIdx.z / ksplit);
  const int kc_end = (int)((int64_t)nchunks_all * (blockIdx.z + 1) / ksplit);
  const int64_t M = (int64_t)p.n_img * p.hi * p.wi;

  if (tid < WT) {
    const int64_t t = t0 + tid;
    int n = -1, ty = 0, tx = 0;
    if (t < ntiles) {
      n = (int)(t / (th * tw));
      const int rem = (int)(t - (int64_t)n * th * tw);
      ty = rem / tw;
      tx = rem - ty * tw;
    }
    s_n[tid] = n;
    s_ty[tid] = ty;
    s_tx[tid] = tx;
  }
  __syncthreads();

  // ---- transform threads: (tile, float4 of channels) ----
  const bool xform = tid < WT * (WKC / 4);
  const int x_tile = tid >> 2, x_c4 = tid & 3;
  const int64_t in_rows = (int64_t)p.n_img * p.hi * p.wi;
  const lfdm_buf buf0 = lfdm_make_buf(p.src0, (uint32_t)(((in_rows - 1) * p.ld0 + p.c0) * 4));
  const lfdm_buf buf1 = p.c1 > 0 ? lfdm_make_buf(p.src1, (uint32_t)(((in_rows - 1) * p.ld1 + p.c1) * 4)) : buf0;
  int pix_base = 0;                  // pixel index of the patch's (0,0) corner (may be outside the image)
  unsigned valid_mask = 0;           // bit (py*4+px): patch pixel inside the image
  if (xform && s_n[x_tile] >= 0) {
    const int n = s_n[x_tile], ty = s_ty[x_tile], tx = s_tx[x_tile];
    pix_base = (n * p.hi + 2 * ty - 1) * p.wi + 2 * tx - 1;
    for (int py = 0; py < 4; ++py)
      for (int px = 0; px < 4; ++px) {
        const int iy = 2 * ty - 1 + py, ix = 2 * tx - 1 + px;
        if (iy >= 0 && iy < p.hi && ix >= 0 && ix < p.wi) valid_mask |= 1u << (py * 4 + px);
      }
  }
  float4 patch[16];
  auto fetch_patch = [&](int chunk) {
    int cc = chunk * WKC;
    const bool second = cc >= p.c0;
    if (second) cc -= p.c0;
    const lfdm_buf buf = second ? buf1 : buf0;
    const int ld = second ? p.ld1 : p.ld0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int py = q >> 2, px = q & 3;
      const uint32_t off = ((valid_mask >> q) & 1u)
                               ? (uint32_t)((((int64_t)pix_base + py * p.wi + px) * ld + cc + 4 * x_c4) * 4)
                               : LFDM_BUF_OOB;
      patch[q] = lfdm_buf_load_f4(buf, off);
    }
  };
  // ---- weight fragments: lane (co = n0 + l31, k-slot kh) holds U[pos][16*chunk + 8*kh + s][co], s = 0..7 ----
  const lfdm_buf bufw = lfdm_make_buf(p.weight_wino, (uint32_t)((int64_t)16 * nchunks_all * p.coutp * WKC * 4));
  float4 bcur[4][2], bnext[4][2];
  auto fetch_b = [&](float4 (&dst)[4][2], int chunk) {
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) {
      const int pos = 4 * wave + pi;
      const uint32_t off = (n0 + l31 < p.coutp)
                               ? (uint32_t)(((((int64_t)pos * nchunks_all + chunk) * p.coutp + n0 + l31) * WKC + 8 * kh) * 4)
                               : LFDM_BUF_OOB;
      dst[pi][0] = lfdm_buf_load_f4(bufw, off);
      dst[pi][1] = lfdm_buf_load_f4(bufw, off == LFDM_BUF_OOB ? LFDM_BUF_OOB : off + 16);
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int pi = 0; pi < 4; ++pi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[pi][r] = 0.f;

  float* const Vs = smem;           // [16 pos][WT tiles][LDV]
  if (xform) fetch_patch(kc_begin);
  fetch_b(bcur, kc_begin);
  for (int kc = kc_begin; kc < kc_end; ++kc) {
    // ---- B^T d B on float4 (4 channels), rows then columns ----
    if (xform) {
      float4 r[16];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 d0 = patch[c], d1 = patch[4 + c], d2 = patch[8 + c], d3 = patch[12 + c];
        r[c] = make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);
        r[4 + c] = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
        r[8 + c] = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
        r[12 + c] = make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 a = r[4 * i], b = r[4 * i + 1], c = r[4 * i + 2], d = r[4 * i + 3];
        float* dst = Vs + ((4 * i) * WT + x_tile) * LDV + 4 * x_c4;
        *reinterpret_cast<float4*>(dst) = make_float4(a.x - c.x, a.y - c.y, a.z - c.z, a.w - c.w);
        *reinterpret_cast<float4*>(dst + WT * LDV) = make_float4(b.x + c.x, b.y + c.y, b.z + c.z, b.w + c.w);
        *reinterpret_cast<float4*>(dst + 2 * WT * LDV) = make_float4(c.x - b.x, c.y - b.y, c.z - b.z, c.w - b.w);
        *reinterpret_cast<float4*>(dst + 3 * WT * LDV) = make_float4(b.x - d.x, b.y - d.y, b.z - d.z, b.w - d.w);
      }
    }
    __syncthreads();
    {
      const int nxt = kc + 1 < kc_end ? kc + 1 : kc;           // clamped: harmless re-fetch after the last chunk
      if (xform) fetch_patch(nxt);
      fetch_b(bnext, nxt);
    }
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) {
      const float* va = Vs + ((4 * wave + pi) * WT + l31) * LDV + 8 * kh;
      const float4 a0 = *reinterpret_cast<const float4*>(va);
      const float4 a1 = *reinterpret_cast<const float4*>(va + 4);
      acc[pi] = mfma_32x32x2(a0.x, bcur[pi][0].x, acc[pi]);
      acc[pi] = mfma_32x32x2(a0.y, bcur[pi][0].y, acc[pi]);
      acc[pi] = mfma_32x32x2(a0.z, bcur[pi][0].z, acc[pi]);
      acc[pi] = mfma_32x32x2(a0.w, bcur[pi][0].w, acc[pi]);
      acc[pi] = mfma_32x32x2(a1.x, bcur[pi][1].x, acc[pi]);
      acc[pi] = mfma_32x32x2(a1.y, bcur[pi][1].y, acc[pi]);
      acc[pi] = mfma_32x32x2(a1.z, bcur[pi][1].z, acc[pi]);
      acc[pi] = mfma_32x32x2(a1.w, bcur[pi][1].w, acc[pi]);
    }
    __syncthreads();
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) {
      bcur[pi][0] = bnext[pi][0];
      bcur[pi][1] = bnext[pi][1];
    }
  }

  // ---- M[pos][tile][co] -> LDS, output transform A^T M A, epilogue ----
  float* const Ms = smem;           // [16][WT][LDM]
#pragma unroll
  for (int pi = 0; pi < 4; ++pi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tile = (r & 3) + 8 * (r >> 2) + 4 * kh;
      Ms[((4 * wave + pi) * WT + tile) * LDM + l31] = acc[pi][r];
    }
  __syncthreads();
  const int co = n0 + l31;
  float gs = 0.f, gq = 0.f;
  const float bb = (p.bias && ksplit == 1 && co < p.cout) ? p.bias[co] : 0.f;
#pragma unroll 1
  for (int it = 0; it < WT / 8; ++it) {
    const int tile = (tid >> 5) + 8 * it;
    const int n = s_n[tile];
    if (n < 0 || co >= p.coutp) continue;
    float m[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) m[q] = Ms[(q * WT + tile) * LDM + l31];
    float y[4];
    {
      float tt[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        tt[0][j] = m[j] + m[4 + j] + m[8 + j];
        tt[1][j] = m[4 + j] - m[8 + j] - m[12 + j];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        y[2 * i] = tt[i][0] + tt[i][1] + tt[i][2];
        y[2 * i + 1] = tt[i][1] - tt[i][2] - tt[i][3];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int oy = 2 * s_ty[tile] + (q >> 1), ox = 2 * s_tx[tile] + (q & 1);
      const int64_t orow = ((int64_t)n * p.hi + oy) * p.wi + ox;
      if (ksplit > 1) {
        p.partial[((int64_t)blockIdx.z * M + orow) * p.coutp + co] = y[q];
      } else if (co < p.cout) {
        float v = y[q] + bb;
        gs += v;
        gq += v * v;
        if (p.residual) v += p.residual[orow * p.ldr + co];
        if (ACT) v = apply_act(v, p.act);
        p.out[orow * p.ldo + co] = v;
      }
    }
  }
  if (p.gn_partial && ksplit == 1) {
    s_gn[0][tid >> 5][l31] = gs;
    s_gn[1][tid >> 5][l31] = gq;
    __syncthreads();
    const int cg = p.cout / p.gn_groups;
    const int gpt = WN / cg;                      // groups inside this column tile (cg divides 32: host check)
    if (tid < gpt && n0 + tid * cg < p.cout) {
      float s = 0.f, q = 0.f;
      for (int c = 0; c < cg; ++c)
        for (int w8 = 0; w8 < 8; ++w8) {
          s += s_gn[0][w8][tid * cg + c];
          q += s_gn[1][w8][tid * cg + c];
        }
      float* dst = p.gn_partial + ((int64_t)blockIdx.x * p.gn_groups + (n0 / cg + tid)) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

}  // namespace

// grid (tile blocks, column tiles, ksplit).  Called by lfdm_conv2d_cl_f32 (conv_igemm.hip).
int lfdm_conv_wino_launch(const lfdm_conv_params& p, hipStream_t stream) {
  const int64_t ntiles = (int64_t)p.n_img * (p.hi / 2) * (p.wi / 2);
  const dim3 grid((unsigned)((ntiles + WT - 1) / WT), (unsigned)((p.coutp + WN - 1) / WN), p.ksplit > 1 ? p.ksplit : 1);
  if (p.act != LFDM_ACT_NONE) LFDM_LAUNCH((conv_wino_kernel<true>), grid, dim3(256), 0, stream, p);
  else LFDM_LAUNCH((conv_wino_kernel<false>), grid, dim3(256), 0, stream, p);
  return lfdm_check_launch("conv_wino");
}
