// Winograd F(2x2,3x3), "wide" schedule - EXPERIMENT for the next round (opt-in LFDM_WINO_WIDE=1; parity-tested under the
// emulator, never timed).  Same mathematics, operand layouts and epilogue contract as conv_wino.hip; different traffic:
//   * 64 output tiles (256 pixels) x 32 output channels per workgroup, 8-channel chunks: every weight fragment (one float4
//     per position and chunk, straight from global memory) multiplies TWO A tiles -> half the weight bytes per MFMA;
//   * the input always goes through the unique-pixel staging of conv_wino.hip's STAGE variant (16-byte loads of each tile
//     row's 4-row band into LDS, patches built from there) -> ~2.3x fewer TA cycles for the patches.
// Per chunk and wave: 4 positions x 2 tile halves x 4 k-steps = 32 MFMAs (as in conv_wino.hip), 16 KB of weights and
// ~27 KB of staged pixels per 64 tiles, where the default schedule moves 32 KB + 32 KB per 32 tiles.  ~215 VGPRs, ~80 KB LDS:
// two workgroups per CU.  Needs >= 8 tiles per image row (the 16x16 level and up) so that the staged bands fit.
#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

constexpr int XT = 64;             // tiles per workgroup (two 32-row MFMA tiles)
constexpr int XN = 32;             // output channels per workgroup
constexpr int XKC = 8;             // input channels per chunk
constexpr int XLD = XKC + 4;       // LDS row stride of V (float4 aligned)
constexpr int XRS = XKC + 4;       // floats per staged pixel
constexpr int XSP = 576;           // staged pixels at most: 8 tiles per row -> 8 segments x 4 rows x 18 columns
constexpr int XSTG = 5;            // float4 stage loads per thread: ceil(XSP * 2 / 256)
constexpr int XLDM = XN + 1;

template <bool ACT>
__global__ __launch_bounds__(256, 2) void conv_wino_wide_kernel(lfdm_conv_params p) {
  constexpr int VSZ = 16 * XT * XLD;                       // 49 KB
  __shared__ __attribute__((aligned(16))) float smem[VSZ];   // V during the loop; 8*32*XLDM epilogue planes per tile half after
  static_assert(VSZ >= 8 * 32 * XLDM, "epilogue planes must fit in the V buffer");
  __shared__ __attribute__((aligned(16))) float raw[XSP * XRS];
  __shared__ int s_n[XT], s_ty[XT], s_tx[XT];
  __shared__ float s_gn[2][8][XN];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int up = p.upsample ? 1 : 0;
  const int th = p.hq >> 1, tw = p.wq >> 1;
  const unsigned ntiles = (unsigned)p.n_img * th * tw;
  const unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  const unsigned t0 = bx * XT;
  const int n0 = by * XN;
  const int cin = p.c0 + p.c1;
  const int nch16 = cin / 16;                              // chunks of the packed filter layout [16][cin/16][coutp][16]
  const int nch8 = cin / XKC;
  const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
  const int kc_begin = (int)((int64_t)nch8 * bz / ksplit);
  const int kc_end = (int)((int64_t)nch8 * (bz + 1) / ksplit);
  const int64_t M = (int64_t)p.n_img * p.hq * p.wq;

  if (tid < XT) {
    const unsigned t = t0 + tid;
    int n = -1, ty = 0, tx = 0;
    if (t < ntiles) {
      n = (int)(t / (unsigned)(th * tw));
      const unsigned rem = t - (unsigned)n * (th * tw);
      ty = (int)(rem / (unsigned)tw);
      tx = (int)(rem - (unsigned)ty * tw);
    }
    s_n[tid] = n;
    s_ty[tid] = ty;
    s_tx[tid] = tx;
  }
  __syncthreads();

  const int64_t in_rows = (int64_t)p.n_img * p.hi * p.wi;
  const lfdm_buf buf0 = lfdm_make_buf(p.src0, (uint32_t)(((in_rows - 1) * p.ld0 + p.c0) * 4));
  const lfdm_buf buf1 = p.c1 > 0 ? lfdm_make_buf(p.src1, (uint32_t)(((in_rows - 1) * p.ld1 + p.c1) * 4)) : buf0;
  // ---- staging: segment r = tiles [r*TPR, (r+1)*TPR) (one image row of tiles, or 64 of one); band = logical rows
  // 2ty-1..2ty+2 x columns 2tx0-1..2(tx0+TPR); staged pixel q = (r*4 + py)*BW + bx; two float4 items per pixel ----
  const int TPR = tw < XT ? tw : XT, BW = 2 * TPR + 2, SP2 = (XT / TPR) * 4 * BW * 2;
  uint32_t spix[XSTG];
#pragma unroll
  for (int j = 0; j < XSTG; ++j) {
    const int item = tid + 256 * j;
    spix[j] = 0xFFFFFFFFu;
    if (item < SP2) {
      const int q = item >> 1;
      const int r = q / (4 * BW), rem = q - r * 4 * BW;
      const int py = rem / BW, bxx = rem - py * BW;
      const int n = s_n[r * TPR];
      const int iy = 2 * s_ty[r * TPR] - 1 + py, ix = 2 * s_tx[r * TPR] - 1 + bxx;
      if (n >= 0 && iy >= 0 && iy < p.hq && ix >= 0 && ix < p.wq)
        spix[j] = (uint32_t)((n * p.hi + (iy >> up)) * p.wi + (ix >> up));
    }
  }
  float4 stg[XSTG];
  auto fetch_stage = [&](int chunk) {
    int cc = chunk * XKC;
    const bool second = cc >= p.c0;
    if (second) cc -= p.c0;
    const lfdm_buf buf = second ? buf1 : buf0;
    const uint32_t ld4 = (uint32_t)(second ? p.ld1 : p.ld0) * 4u;
#pragma unroll
    for (int j = 0; j < XSTG; ++j) {
      const uint32_t c4 = (uint32_t)((tid + 256 * j) & 1);
      stg[j] = lfdm_buf_load_f4(buf, spix[j] != 0xFFFFFFFFu ? spix[j] * ld4 + ((uint32_t)cc + 4u * c4) * 4u : LFDM_BUF_OOB);
    }
  };
  auto write_stage = [&]() {
#pragma unroll
    for (int j = 0; j < XSTG; ++j) {
      const int item = tid + 256 * j;
      if (item < SP2) *reinterpret_cast<float4*>(raw + (item >> 1) * XRS + 4 * (item & 1)) = stg[j];
    }
  };
  // ---- input transform: one (tile, channel pair) patch per thread: 64 tiles x 4 pairs ----
  const int x_tile = tid >> 2, x_c2 = tid & 3;
  const int seg = x_tile / TPR;
  const float* const raw_patch = raw + ((seg * 4) * BW + 2 * (x_tile - seg * TPR)) * XRS + 2 * x_c2;
  float2 patch[16];
  auto xform_part = [&](int i, float* V) {
    float2 r[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float2 d0 = patch[c], d1 = patch[4 + c], d2 = patch[8 + c], d3 = patch[12 + c];
      r[c] = i == 0 ? make_float2(d0.x - d2.x, d0.y - d2.y)
           : i == 1 ? make_float2(d1.x + d2.x, d1.y + d2.y)
           : i == 2 ? make_float2(d2.x - d1.x, d2.y - d1.y)
                    : make_float2(d1.x - d3.x, d1.y - d3.y);
    }
    float* dst = V + ((4 * i) * XT + x_tile) * XLD + 2 * x_c2;
    *reinterpret_cast<float2*>(dst) = make_float2(r[0].x - r[2].x, r[0].y - r[2].y);
    *reinterpret_cast<float2*>(dst + XT * XLD) = make_float2(r[1].x + r[2].x, r[1].y + r[2].y);
    *reinterpret_cast<float2*>(dst + 2 * XT * XLD) = make_float2(r[2].x - r[1].x, r[2].y - r[1].y);
    *reinterpret_cast<float2*>(dst + 3 * XT * XLD) = make_float2(r[1].x - r[3].x, r[1].y - r[3].y);
  };
  // ---- weight fragments: lane (co = n0 + l31, k-slot kh) holds U[pos][8*chunk + 4*kh + s][co], s = 0..3 ----
  const lfdm_buf bufw = lfdm_make_buf(p.weight_wino, (uint32_t)((int64_t)16 * nch16 * p.coutp * 16 * 4));
  float4 bfrag[4];
  auto fetch_b = [&](int pi, int chunk) {
    const int pos = 4 * wave + pi;
    const int n = n0 + l31;
    bfrag[pi] = lfdm_buf_load_f4(bufw, n < p.coutp
        ? (uint32_t)(((((int64_t)pos * nch16 + (chunk >> 1)) * p.coutp + n) * 16 + 8 * (chunk & 1) + 4 * kh) * 4)
        : LFDM_BUF_OOB);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int pi = 0; pi < 4; ++pi)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pi][mt][r] = 0.f;

  const int kc_last = kc_end - 1;
  auto clampc = [&](int c) { return c < kc_last ? c : kc_last; };
  float* const Vs = smem;           // [16 pos][XT tiles][XLD]
#pragma unroll
  for (int pi = 0; pi < 4; ++pi) fetch_b(pi, kc_begin);
  fetch_stage(kc_begin);
  for (int kc = kc_begin; kc < kc_end; ++kc) {
    const int nxt = clampc(kc + 1);
    write_stage();
    __syncthreads();                 // band in LDS; every wave has left the previous chunk's MFMA phase (V is free)
#pragma unroll
    for (int q = 0; q < 16; ++q) patch[q] = *reinterpret_cast<const float2*>(raw_patch + ((q >> 2) * BW + (q & 3)) * XRS);
#pragma unroll
    for (int i = 0; i < 4; ++i) xform_part(i, Vs);
    __syncthreads();                 // V complete; all reads of the staged band done
    fetch_stage(nxt);
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) {
      const float* va = Vs + ((4 * wave + pi) * XT + l31) * XLD + 4 * kh;
      const float4 a0 = *reinterpret_cast<const float4*>(va);               // tiles 0..31
      const float4 a1 = *reinterpret_cast<const float4*>(va + 32 * XLD);    // tiles 32..63
      const float4 b = bfrag[pi];
      acc[pi][0] = mfma_32x32x2(a0.x, b.x, acc[pi][0]);
      acc[pi][1] = mfma_32x32x2(a1.x, b.x, acc[pi][1]);
      acc[pi][0] = mfma_32x32x2(a0.y, b.y, acc[pi][0]);
      acc[pi][1] = mfma_32x32x2(a1.y, b.y, acc[pi][1]);
      acc[pi][0] = mfma_32x32x2(a0.z, b.z, acc[pi][0]);
      acc[pi][1] = mfma_32x32x2(a1.z, b.z, acc[pi][1]);
      acc[pi][0] = mfma_32x32x2(a0.w, b.w, acc[pi][0]);
      acc[pi][1] = mfma_32x32x2(a1.w, b.w, acc[pi][1]);
      fetch_b(pi, nxt);
    }
  }
  __syncthreads();                   // the epilogue reuses the V buffer

  // ---- output transform (column sum in registers, row sum across waves through LDS), one 32-tile half at a time; each half
  // is one 128-pixel GroupNorm-partial block, exactly as conv_wino.hip emits them ----
  float* const Ms = smem;            // [8 = 2*i + j'][32][XLDM]
  const int co = n0 + l31;
  const float bb = (p.bias && ksplit == 1 && co < p.cout) ? p.bias[co] : 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    if (mt > 0) __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tile = (r & 3) + 8 * (r >> 2) + 4 * kh;
      Ms[((2 * wave) * 32 + tile) * XLDM + l31] = acc[0][mt][r] + acc[1][mt][r] + acc[2][mt][r];
      Ms[((2 * wave + 1) * 32 + tile) * XLDM + l31] = acc[1][mt][r] - acc[2][mt][r] - acc[3][mt][r];
    }
    __syncthreads();
    float gs = 0.f, gq = 0.f;
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
      const int tl = (tid >> 5) + 8 * it;                   // tile inside this half
      const int tile = 32 * mt + tl;
      const int n = s_n[tile];
      if (n < 0 || co >= p.coutp) continue;
      float m[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) m[q] = Ms[(q * 32 + tl) * XLDM + l31];
      float y[4];
      y[0] = m[0] + m[2] + m[4];
      y[1] = m[1] + m[3] + m[5];
      y[2] = m[2] - m[4] - m[6];
      y[3] = m[3] - m[5] - m[7];
      const int64_t orow0 = ((int64_t)n * p.hq + 2 * s_ty[tile]) * p.wq + 2 * s_tx[tile];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t orow = orow0 + (q >> 1) * p.wq + (q & 1);
        if (ksplit > 1) {
          p.partial[((int64_t)bz * M + orow) * p.coutp + co] = y[q];
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
      const int gpt = XN / cg;
      const int64_t blk = (int64_t)bx * 2 + mt;             // 128-pixel block index (lfdm_conv2d_plan's tile_rows = 128)
      if (tid < gpt && n0 + tid * cg < p.cout && (unsigned)(t0 + 32 * mt) < ntiles) {
        float s = 0.f, q = 0.f;
        for (int c = 0; c < cg; ++c)
          for (int w8 = 0; w8 < 8; ++w8) {
            s += s_gn[0][w8][tid * cg + c];
            q += s_gn[1][w8][tid * cg + c];
          }
        float* dst = p.gn_partial + (blk * p.gn_groups + (n0 / cg + tid)) * 2;
        dst[0] = s;
        dst[1] = q;
      }
    }
  }
}

}  // namespace

// true if this geometry can run the wide schedule (conv_wino.hip's launcher asks when LFDM_WINO_WIDE=1)
bool lfdm_conv_wino_wide_ok(const lfdm_conv_params& p) {
  const int tw = p.wq / 2;
  return tw >= 8 && ((tw <= XT && XT % tw == 0) || tw % XT == 0) && (p.c0 % XKC == 0) && (p.c1 % XKC == 0);
}

int lfdm_conv_wino_wide_launch(const lfdm_conv_params& p, hipStream_t stream) {
  const int64_t ntiles = (int64_t)p.n_img * (p.hq / 2) * (p.wq / 2);
  const dim3 grid((unsigned)((ntiles + XT - 1) / XT), (unsigned)((p.coutp + XN - 1) / XN), p.ksplit > 1 ? p.ksplit : 1);
  if (p.act != LFDM_ACT_NONE) LFDM_LAUNCH((conv_wino_wide_kernel<true>), grid, dim3(256), 0, stream, p);
  else LFDM_LAUNCH((conv_wino_wide_kernel<false>), grid, dim3(256), 0, stream, p);
  return lfdm_check_launch("conv_wino_wide");
}
