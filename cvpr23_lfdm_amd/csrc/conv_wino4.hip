// Winograd F(4x4, 3x3) schedule of the 3x3 / stride-1 / zero-padded convolution on fp32 MFMA, for the BATCHED shapes (training's
// frozen-LFAE decode, throughput mode): include/lfdm_hip.h lfdm_conv_params.weight_wino4 (schedule 4).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A   per 6x6 input patch d / 4x4 output tile Y: 36 "frequency positions", each an ordinary GEMM
// over c_in -> 36/144 = 1/4 of the multiplications of the direct form, 16/9 fewer than F(2x2, 3x3) (conv_wino.hip).  That kernel's K
// loop already runs at the matrix-pipe bound on these shapes (profiles/r02_g_wino_phases.txt), so only a cheaper algorithm moves it.
// The price: 36 accumulator tiles instead of 16, and a transform of ~12 packed operations per patch element - a workgroup that did
// both like conv_wino.hip would need > 256 registers per lane or leave the matrix pipe idle during its transforms.  So the roles are
// split (one workgroup per CU, 512 threads, every SIMD holds one consumer and one producer wave):
//   * waves 0-3 = CONSUMERS: wave w owns positions 9w .. 9w+8 of 32 tiles x 32 output channels (9 x 16 accumulator registers), A
//     fragments = V from LDS (ds_read_b128), B fragments = the pre-transformed filters straight from global memory in operand order
//     ([36][C_in/8][coutp][8]), three positions at a time, requested two groups ahead;
//   * waves 4-7 = PRODUCERS: thread = (tile, channel pair of a 16-channel period) loads the pair's 6x6 patch (36 8-byte buffer loads;
//     the 8 pairs of a pixel are 8 adjacent lanes = one 64-byte segment, out-of-image taps = out-of-range offset = 0), applies B^T d B
//     with packed fp32 arithmetic (both channels per instruction) and writes V[36][32][16] of the NEXT period into the other LDS
//     buffer, then requests the patch after that (two register sets, two periods ahead);
//   * one barrier per 16-channel period; the matrix pipe of every SIMD sees a continuous MFMA stream while the VALU transforms.
//   (First version: 8-channel periods produced by two waves from 32-byte segments: the ablation builds of tools/probe_wino4_ablate.sh
//   showed the consumers alone at 803 us and the patch loads costing 520 us of a 1527 us launch - half-used 64-byte L2 requests.)
// Output transform: all accumulators go through LDS once (36 planes x 32 tiles x 32 channels = 144 KB, aliasing the V buffers),
// thread = (tile, 4 channels) applies A^T . A and writes the tile's 16 pixels as float4 with bias / residual / activation.
// Error against the direct form: ~4e-6 of the output scale in fp32 (tests/test_ops_parity.py::test_conv2d_winograd4), F(2x2): ~1e-6.
// Measured (MI355X, profiles/r03_p_*): 3x3 256 -> 256 on 320 frames of 32x32 1838 -> 1419 us (1.30x F(2x2)); the consumers alone would
// run it in 795 us - the launch is bound by 12 GB of L2 -> L1 traffic (patches + filter fragments) at this 32-tile x 32-channel
// workgroup tile, which is what 144 accumulator registers per consumer lane allow.
#include <cstdio>
#include <cstdlib>

#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int W4T = 32;            // tiles per workgroup (4x4 outputs each: 512 pixels)
constexpr int W4N = 32;            // output channels per workgroup
constexpr int W4K = 8;             // input channels per chunk
constexpr int W4C = 16;            // input channels per barrier period (two 8-channel chunks of the filter layout)
constexpr int LDV4 = W4C;          // LDS row of V: 16 floats = four 16-byte slots, slot s of row r stored at s ^ ((r >> 2) & 3) (no padding:
                                   // the swizzle makes the b128 fragment reads of 16 consecutive rows conflict-free)
constexpr int V4SZ = 36 * W4T * LDV4;      // floats per V buffer (73 728 B)

// B^T x for one 6-vector (Lavin & Gray, F(4x4,3x3)): 12 operations
__device__ __forceinline__ void bt6(const f32x2 d0, const f32x2 d1, const f32x2 d2, const f32x2 d3, const f32x2 d4, const f32x2 d5,
                                    f32x2& o0, f32x2& o1, f32x2& o2, f32x2& o3, f32x2& o4, f32x2& o5) {
  const f32x2 a = d4 - 4.0f * d2, b = d3 - 4.0f * d1, c = d4 - d2, e = d3 - d1;
  o0 = 4.0f * d0 - 5.0f * d2 + d4;
  o1 = a + b;
  o2 = a - b;
  o3 = c + 2.0f * e;
  o4 = c - 2.0f * e;
  o5 = 4.0f * d1 - 5.0f * d3 + d5;
}

template <bool ACT>
__global__ __launch_bounds__(512) void conv_wino4_kernel(lfdm_conv_params p, int gx, int ny, int ablate) {
  // ablate: bit 0 = raised instruction-issue priority for the producer waves (LFDM_W4_PRIO; probe builds -DLFDM_W4_PROBE=<mask> leave pipeline stages out at compile time: tools/probe_wino4_ablate.sh)
  constexpr int SMEM4 = 2 * V4SZ > 36 * W4T * W4N ? 2 * V4SZ : 36 * W4T * W4N;
  __shared__ __attribute__((aligned(16))) float smem[SMEM4];      // V double buffer (144 KB); the epilogue's accumulator planes (144 KB) alias it

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = lfdm_uniform(tid >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  const int up = p.upsample ? 1 : 0;
  const int th = p.hq >> 2, tw = p.wq >> 2;
  const unsigned ntiles = (unsigned)p.n_img * th * tw;
  // Workgroup order (1-D grid): ids are dealt round-robin to the 8 XCDs; inside an XCD consecutive workgroups walk the COLUMN tiles of
  // one tile block, so the workgroups that need the same input patches run at the same time behind the same L2 (each reads its
  // 1.2 MB of patches once from HBM instead of once per column tile), and the tile blocks that are in flight together on an XCD walk
  // the filter chunks roughly in step (the 9.4 MB of F(4x4) filters come through L2 / the Infinity Cache, not HBM).
  // First version (grid = tile blocks x column tiles, tile blocks fastest): HBM-bound - 6 GB of input re-reads per 256 -> 256 launch.
  const int id = blockIdx.x, slot = id >> 3;
  const int bx = (id & 7) + 8 * (slot / ny), by = slot % ny;
  if (bx >= gx) return;          // (whole workgroups: no barrier has been reached)
  const unsigned t0 = (unsigned)bx * W4T;
  const int n0 = by * W4N;
  const int cin = p.c0;
  const int nch = cin / W4K;           // 8-channel chunks of the filter layout
  const int nper = cin / W4C;          // 16-channel periods (even: C_in % 32 == 0 is the plan's condition)

  // nper is even: the period loops below run two periods per trip as ONE basic block - with a break between the halves hipcc renamed
  // the accumulators from trip to trip (out-of-place MFMAs: both tuples live, spills)
  const int last = nch - 1, lastp = nper - 1;
  auto clampc = [&](int c) { return c < last ? c : last; };      // re-fetching the last chunk / period is harmless
  auto clampp = [&](int c) { return c < lastp ? c : lastp; };
  float* const V0 = smem;
  float* const V1 = smem + V4SZ;

  // The two roles live in separate branches (the wave index is scalar: a uniform branch), each with its own copy of the chunk loop
  // and the SAME number of barriers - a common loop would keep the producers' 144 patch registers and the consumers' 144 accumulator
  // registers alive together.
  if (wave >= 4) {
    // ---------------------------------------------------------------- PRODUCERS (waves 4-7): thread = (tile, channel pair of 8)
    // Instruction-issue priority for the producers (round 4).  Cycle stamps (-DLFDM_W4_STAMP, tools/probe_wino4_stamp.py) showed the
    // CONSUMERS waiting 3 900 of every 9 000 cycles at the period barrier while each producer instruction took ~38 cycles: a SIMD's matrix
    // pipe and vector ALU do not run side by side here, and at equal priority the consumer wave's next MFMA (64 cycles) wins the
    // issue slot whenever the producer wave's dependent transform chain is not ready in that very cycle.  With priority the producers
    // finish a period in ~6 600 cycles instead of 9 000: 1 411 -> 1 263 us on 256 -> 256 @32x32 x 320 frames.
#if !defined(LFDM_EMU_BUILD)
    if (ablate & 1) __builtin_amdgcn_s_setprio(1);
#endif
    const int pt = tid - 256;
    const int x_tile = pt >> 3, x_pair = pt & 7;
    const int64_t in_rows = (int64_t)p.n_img * p.hi * p.wi;
    const lfdm_buf buf0 = lfdm_make_buf(p.src0, (uint32_t)(((in_rows - 1) * p.ld0 + p.c0) * 4));
    uint32_t base0 = 0;
    uint64_t valid = 0;                             // bit (6*r + c): patch pixel inside the image
    {
      const unsigned t = t0 + x_tile;
      if (t < ntiles) {
        const int n = (int)(t / (unsigned)(th * tw));
        const unsigned rem = t - (unsigned)n * (th * tw);
        const int ty = (int)(rem / (unsigned)tw), tx = (int)(rem - (unsigned)ty * tw);
        // physical pixel of the patch corner: logical (4ty-1, 4tx-1); through the upsample that is (2ty-1, 2tx-1)
        const uint32_t pix = up ? (uint32_t)((n * p.hi + 2 * ty - 1) * p.wi + 2 * tx - 1)
                                : (uint32_t)((n * p.hi + 4 * ty - 1) * p.wi + 4 * tx - 1);
        base0 = (pix * (uint32_t)p.ld0 + 2u * x_pair) * 4u;
        const unsigned rows = 0x3Fu & ~(ty == 0 ? 1u : 0u) & ~(ty == th - 1 ? 32u : 0u);
        const unsigned cols = 0x3Fu & ~(tx == 0 ? 1u : 0u) & ~(tx == tw - 1 ? 32u : 0u);
#pragma unroll
        for (int r = 0; r < 6; ++r)
          if ((rows >> r) & 1u) valid |= (uint64_t)cols << (6 * r);
      }
    }
    const uint32_t ld4 = (uint32_t)p.ld0 * 4u;
    f32x2 pa[36], pb[36];                           // two patch register sets (two chunks in flight)
    // (Round 4, measured and dropped: per-lane byte offsets of the 36 patch pixels kept in registers + the period as the SGPR offset of
    // the load - no vector arithmetic per fetch - and the same for the consumers' filter loads: 3-7 % SLOWER; the loads then leave in
    // one burst and the launch sits on L2 -> L1 delivery instead, profiles/r04_w_wino4_stamps.txt.)
    auto fetch_patch = [&](f32x2 (&d)[36], int chunk) {
#ifdef LFDM_W4_PROBE
      if constexpr ((LFDM_W4_PROBE & 1) != 0) return;
#endif
      const uint32_t base = base0 + (uint32_t)chunk * (W4C * 4u);
      // issue order inside a patch row: columns 0, 4, 1, 5, 2, 3 - a wave's 8 lanes-of-8 are 8 horizontally adjacent tiles, and column
      // c + 4 of tile i is column c of tile i + 1: requested back to back, 7/8 of the second load's 64-byte segments hit L1
#pragma unroll
      for (int qq = 0; qq < 36; ++qq) {
        constexpr int corder[6] = {0, 4, 1, 5, 2, 3};
        const int r = qq / 6, c = corder[qq % 6], q = 6 * r + c;
        // upsampled: logical rows 4ty-1 .. 4ty+4 are physical rows 2ty-1 + ((r+1)>>1)
        const int py = up ? (r + 1) >> 1 : r, px = up ? (c + 1) >> 1 : c;
        const uint32_t delta = (uint32_t)(py * p.wi + px) * ld4;
        const float2 v = lfdm_buf_load_f2(buf0, ((valid >> q) & 1ull) ? base + delta : LFDM_BUF_OOB);
        d[q].x = v.x;
        d[q].y = v.y;
      }
    };
    auto transform_store = [&](f32x2 (&d)[36], float* V) {
#ifdef LFDM_W4_PROBE
      if constexpr ((LFDM_W4_PROBE & 16) != 0) return;
      if constexpr ((LFDM_W4_PROBE & 2) != 0) {
        float* dst = V + x_tile * LDV4 + 4 * ((x_pair >> 1) ^ ((x_tile >> 2) & 3)) + 2 * (x_pair & 1);
#pragma unroll
        for (int q = 0; q < 36; ++q) *reinterpret_cast<f32x2*>(dst + q * (W4T * LDV4)) = d[q];
        return;
      }
#endif
      // columns first (t = B^T d, in place), then rows (V = t B)
#pragma unroll
      for (int c = 0; c < 6; ++c)
        bt6(d[c], d[6 + c], d[12 + c], d[18 + c], d[24 + c], d[30 + c], d[c], d[6 + c], d[12 + c], d[18 + c], d[24 + c], d[30 + c]);
      float* dst = V + x_tile * LDV4 + 4 * ((x_pair >> 1) ^ ((x_tile >> 2) & 3)) + 2 * (x_pair & 1);      // swizzled 16-byte slot
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        f32x2 o[6];
        bt6(d[6 * i], d[6 * i + 1], d[6 * i + 2], d[6 * i + 3], d[6 * i + 4], d[6 * i + 5], o[0], o[1], o[2], o[3], o[4], o[5]);
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x2*>(dst + (6 * i + j) * (W4T * LDV4)) = o[j];
      }
    };
    fetch_patch(pa, 0);
    fetch_patch(pb, clampp(1));
    transform_store(pa, V0);
    fetch_patch(pa, clampp(2));
    __syncthreads();
#ifdef LFDM_W4_STAMP
    long long st_work = 0, st_bar = 0, st_t = clock64();
#define LFDM_W4_LAP(acc) { const long long now_ = clock64(); acc += now_ - st_t; st_t = now_; }
#else
#define LFDM_W4_LAP(acc)
#endif
#ifdef LFDM_W4_STAMP
    long long st_wait = 0, st_xf = 0;
#define LFDM_W4_WAITSET() { asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); LFDM_W4_LAP(st_wait) }
#else
#define LFDM_W4_WAITSET()
#endif
    for (int kp = 0; kp < nper; kp += 2) {
      LFDM_W4_WAITSET()
      transform_store(pb, V1);                      // period kp+1 while the consumers multiply period kp from V0
      LFDM_W4_LAP(st_xf)
      fetch_patch(pb, clampp(kp + 3));
      LFDM_W4_LAP(st_work)
      __syncthreads();
      LFDM_W4_LAP(st_bar)
      LFDM_W4_WAITSET()
      transform_store(pa, V0);                      // period kp+2 while the consumers multiply period kp+1 from V1
      LFDM_W4_LAP(st_xf)
      fetch_patch(pa, clampp(kp + 4));
      LFDM_W4_LAP(st_work)
      __syncthreads();
      LFDM_W4_LAP(st_bar)
    }
#ifdef LFDM_W4_STAMP
    if (lane == 0 && blockIdx.x < 256) {
      long long* st = (long long*)p.gn_in_gamma + ((int64_t)blockIdx.x * 8 + wave) * 2;
      st[0] = st_work; st[1] = st_bar;
      long long* st2 = (long long*)p.gn_in_gamma + 256 * 8 * 2 + ((int64_t)blockIdx.x * 8 + wave) * 2;
      st2[0] = st_wait; st2[1] = st_xf;
    }
#endif
    __syncthreads();                                // the consumers' epilogue: one more barrier
    return;
  }

  // ---------------------------------------------------------------- CONSUMERS (waves 0-3, one per SIMD)
  const lfdm_buf bufw = lfdm_make_buf(p.weight_wino4, (uint32_t)((int64_t)36 * nch * p.coutp * W4K * 4));
  const int cw = wave;
  const int ncol = n0 + l31;
  // B fragments: group g (three positions) of a chunk lives in bfr[g]; while group g is multiplied the fragments of the group after
  // next are requested into the buffer that became free one group ago (two groups = 24 MFMAs = ~1500 cycles of flight time; one group
  // ahead - 768 cycles - left the matrix pipe waiting for L2 at every group, 4300 cycles per chunk for 2304 of MFMA)
  float4 bfr[3][3];
  // byte offset = position * (nch * coutp * 32) + chunk * (coutp * 32) [scalar] + this lane's column / k-half [vector]
  const uint32_t chunk_bytes = (uint32_t)p.coutp * (W4K * 4u);
  const uint32_t pos_bytes = (uint32_t)nch * chunk_bytes;
  // (coutp % 32 == 0 and gridDim.y = coutp / 32 - the plan's condition - so every lane's column exists)
  const uint32_t b_lane = ((uint32_t)ncol * W4K + 4u * kh) * 4u;
  auto fetch_bg = [&](float4 (&dst)[3], int g, int chunk) {
#ifdef LFDM_W4_PROBE
    if constexpr ((LFDM_W4_PROBE & 4) != 0) { if (chunk > 0) return; }
#endif
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const uint32_t uni = (uint32_t)(9 * cw + 3 * g + u) * pos_bytes + (uint32_t)chunk * chunk_bytes;
      dst[u] = lfdm_buf_load_f4(bufw, uni + b_lane);
    }
  };
  f32x16 acc[9];
#pragma unroll
  for (int q = 0; q < 9; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const int a_swz = (l31 >> 2) & 3;
  auto load_a = [&](float4 (&a)[3], const float* V, int half, int g) {      // channels [8*half + 4*kh, +4) = 16-byte slot 2*half + kh
#pragma unroll
    for (int u = 0; u < 3; ++u)
      a[u] = *reinterpret_cast<const float4*>(V + ((9 * cw + 3 * g + u) * W4T + l31) * LDV4 + 4 * ((2 * half + kh) ^ a_swz));
  };
  // one 8-channel chunk (half `half` of the period in V) against filter chunk `chunk`
  auto consume = [&](const float* V, int half, int chunk, int next_chunk) {
    float4 a[2][3];                                // A fragments of the current and of the next group (LDS latency under the MFMAs)
    load_a(a[0], V, half, 0);
#pragma unroll
    for (int g = 0; g < 3; ++g) {                 // three positions at a time: independent accumulator chains
      if (g < 2) load_a(a[(g + 1) & 1], V, half, g + 1);
      // the buffer of the PREVIOUS group is free (its MFMAs have been issued): it takes the group after next - group 2 of this chunk
      // (g = 0) or group g - 1 of the next chunk - which then has two groups of MFMAs to arrive
      if (g == 0) fetch_bg(bfr[2], 2, chunk);
      else fetch_bg(bfr[g - 1], g - 1, next_chunk);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const float4 bq = bfr[g][u], aq = a[g & 1][u];
          const float av = e == 0 ? aq.x : e == 1 ? aq.y : e == 2 ? aq.z : aq.w;
          const float bv = e == 0 ? bq.x : e == 1 ? bq.y : e == 2 ? bq.z : bq.w;
#ifdef LFDM_W4_PROBE
          if constexpr ((LFDM_W4_PROBE & 8) != 0) acc[3 * g + u][e] += av * bv;
          else
#endif
          acc[3 * g + u] = mfma_32x32x2(av, bv, acc[3 * g + u]);
        }
#if !defined(LFDM_EMU_BUILD)
      __builtin_amdgcn_sched_barrier(0);          // keep the groups apart: loads hoisted further would need fresh registers
#endif
    }
  };
  fetch_bg(bfr[0], 0, 0);
  fetch_bg(bfr[1], 1, 0);
  __syncthreads();
#ifdef LFDM_W4_STAMP
  long long st_work = 0, st_bar = 0, st_t = clock64();
#endif
  for (int kp = 0; kp < nper; kp += 2) {
    consume(V0, 0, 2 * kp, 2 * kp + 1);
    consume(V0, 1, 2 * kp + 1, 2 * kp + 2);
    LFDM_W4_LAP(st_work)
    __syncthreads();
    LFDM_W4_LAP(st_bar)
    consume(V1, 0, 2 * kp + 2, 2 * kp + 3);
    consume(V1, 1, 2 * kp + 3, clampc(2 * kp + 4));
    LFDM_W4_LAP(st_work)
    __syncthreads();
    LFDM_W4_LAP(st_bar)
  }
#ifdef LFDM_W4_STAMP
  if (lane == 0 && blockIdx.x < 256) {
    long long* st = (long long*)p.gn_in_gamma + ((int64_t)blockIdx.x * 8 + wave) * 2;
    st[0] = st_work; st[1] = st_bar;
  }
#endif

  // ---------------------------------------------------------------- output transform A^T M A through LDS
  // every accumulator goes to its plane first ([36 positions][32 tiles][32 channels]: 147 KB, the accumulators are dead afterwards),
  // then thread = (tile, 4 channels) reads its 36 float4 and produces the tile's 4x4 output pixels
  float* const Ms = smem;
#pragma unroll
  for (int q = 0; q < 9; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tile = (r & 3) + 8 * (r >> 2) + 4 * kh;
      Ms[((9 * cw + q) * W4T + tile) * W4N + l31] = acc[q][r];
    }
  const int e_quad = tid & 7, e_tile = tid >> 3;      // the 256 consumer threads
  int my_n = -1, my_ty = 0, my_tx = 0;
  {
    const unsigned t = t0 + e_tile;
    if (t < ntiles) {
      my_n = (int)(t / (unsigned)(th * tw));
      const unsigned rem = t - (unsigned)my_n * (th * tw);
      my_ty = (int)(rem / (unsigned)tw);
      my_tx = (int)(rem - (unsigned)my_ty * tw);
    }
  }
  const int co = n0 + 4 * e_quad;
  const bool live = my_n >= 0 && co < p.cout;
  const int64_t orow0 = ((int64_t)my_n * p.hq + 4 * my_ty) * p.wq + 4 * my_tx;
  float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live && p.bias) bb = *reinterpret_cast<const float4*>(p.bias + co);
  // the 16 residual rows of the tile are requested before the barrier (the accumulator registers are free now); loaded inside the store
  // loop each was a load -> wait -> store round trip, because `out` may alias `residual` (ResBlock2d: in place) and nothing can be hoisted
  float4 res[16];
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    res[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && p.residual) res[o] = *reinterpret_cast<const float4*>(p.residual + (orow0 + (o >> 2) * p.wq + (o & 3)) * p.ldr + co);
  }
  __syncthreads();
  if (!live) return;
  // T[i][b] = sum_j M[i][j] A[j][b]
  float4 T[6][4];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float4 m[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) m[j] = *reinterpret_cast<const float4*>(Ms + ((6 * i + j) * W4T + e_tile) * W4N + 4 * e_quad);
#define LFDM_W4_COL(f)                                                                                          \
    {                                                                                                           \
      const float s12 = m[1].f + m[2].f, d12 = m[1].f - m[2].f, s34 = m[3].f + m[4].f, d34 = m[3].f - m[4].f;  \
      T[i][0].f = m[0].f + s12 + s34;                                                                           \
      T[i][1].f = d12 + 2.0f * d34;                                                                             \
      T[i][2].f = s12 + 4.0f * s34;                                                                             \
      T[i][3].f = d12 + 8.0f * d34 + m[5].f;                                                                    \
    }
    LFDM_W4_COL(x) LFDM_W4_COL(y) LFDM_W4_COL(z) LFDM_W4_COL(w)
#undef LFDM_W4_COL
  }
  // Y[a][b] = sum_i A^T[a][i] T[i][b], one output column b at a time (4 pixels of 4 channels each)
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    float4 y[4];
#define LFDM_W4_ROW(f)                                                                                                            \
    {                                                                                                                             \
      const float s12 = T[1][b].f + T[2][b].f, d12 = T[1][b].f - T[2][b].f, s34 = T[3][b].f + T[4][b].f, d34 = T[3][b].f - T[4][b].f; \
      y[0].f = T[0][b].f + s12 + s34;                                                                                             \
      y[1].f = d12 + 2.0f * d34;                                                                                                  \
      y[2].f = s12 + 4.0f * s34;                                                                                                  \
      y[3].f = d12 + 8.0f * d34 + T[5][b].f;                                                                                      \
    }
    LFDM_W4_ROW(x) LFDM_W4_ROW(y) LFDM_W4_ROW(z) LFDM_W4_ROW(w)
#undef LFDM_W4_ROW
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int64_t orow = orow0 + a * p.wq + b;
      const float4 rr = res[4 * a + b];
      float4 v = make_float4(y[a].x + bb.x + rr.x, y[a].y + bb.y + rr.y, y[a].z + bb.z + rr.z, y[a].w + bb.w + rr.w);
      if (ACT) {
        v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act);
        v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
      }
      *reinterpret_cast<float4*>(p.out + orow * p.ldo + co) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// STAGED flavour (round 4): the workgroup's 32 tiles form an 8 x 4 block of the image, and the block's UNIQUE input pixels of a
// 16-channel period - 34 x 18 (10 x 18 physical pixels behind the virtual x2 upsample) instead of 32 overlapping 6 x 6 patches, 39 KB
// instead of 74 KB - are copied global -> registers -> LDS once (16-byte coalesced loads, out-of-image pixels = out-of-range offsets =
// zeros: no validity masks anywhere) and the producers cut their patches out of LDS.  After the priority fix the launch sat on L2 -> L1
// delivery (147 KB per period and workgroup, profiles/r04_w_wino4_stamps.txt); this takes a quarter of it away.  LDS: the raw tile is
// double-buffered (2 x 40 KB, quad-major [4 channel quads][625 pixels][16 B]: quad stride = 1 pixel mod 16, so the 8-byte patch reads
// of a half-wave - 4 tiles x 4 quads x 2 halves - hit 64 different banks), which leaves room for ONE V buffer (73.7 KB): a period is
//   phase A  consumers multiply V(k)  ||  producers: patches of raw(k+1) from LDS -> B^T d B in registers; raw(k+2) registers -> LDS;
//                                                    loads of raw(k+3) issued
//   barrier; phase B  producers store V(k+1) (the matrix pipe idles for these ~1 000 cycles);  barrier.
// Conditions (lfdm_conv_wino4_launch): tiles per row % 8 == 0, tile rows % 4 == 0 - every LFAE decode shape.
constexpr int W4S_NPXP = 625;                           // padded pixel count of a raw tile (612 used; 625 = 1 mod 16)
constexpr int W4S_QS = W4S_NPXP * 4;                    // floats per channel-quad plane
constexpr int W4S_RAW = 4 * W4S_QS;                     // floats per raw buffer (40 000 B)

template <bool ACT, bool UP>
__global__ __launch_bounds__(512) void conv_wino4s_kernel(lfdm_conv_params p, int gx, int ny, int ablate) {
  constexpr int SMEM4S = V4SZ + 2 * W4S_RAW > 36 * W4T * W4N ? V4SZ + 2 * W4S_RAW : 36 * W4T * W4N;
  __shared__ __attribute__((aligned(16))) float smem[SMEM4S];      // V (73.7 KB) | raw0 | raw1 (40 KB each); the epilogue's planes (147 KB) alias all of it
  constexpr int RW = UP ? 18 : 34, RH = UP ? 10 : 18;              // raw tile in (physical) pixels
  constexpr int NPIX = RW * RH, NPIECE = NPIX * 4, NLD = (NPIECE + 255) / 256;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = lfdm_uniform(tid >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  const int th = p.hq >> 2, tw = p.wq >> 2;
  const int bw = tw >> 3, bpi = bw * (th >> 2);                    // tile blocks per row / per image
  const int id = blockIdx.x, slot = id >> 3;
  const int bx = (id & 7) + 8 * (slot / ny), by = slot % ny;       // (same XCD-aware order as conv_wino4_kernel)
  if (bx >= gx) return;
  const int blk_n = bx / bpi, blk_r = bx - blk_n * bpi;
  const int blk_y = blk_r / bw, blk_x = blk_r - blk_y * bw;
  const int n0 = by * W4N;
  const int cin = p.c0;
  const int nch = cin / W4K, nper = cin / W4C;
  const int last = nch - 1, lastp = nper - 1;
  auto clampc = [&](int c) { return c < last ? c : last; };
  auto clampp = [&](int c) { return c < lastp ? c : lastp; };
  float* const V = smem;
  float* const R0 = smem + V4SZ;
  float* const R1 = R0 + W4S_RAW;

  if (wave >= 4) {
    // ---------------------------------------------------------------- PRODUCERS
#if !defined(LFDM_EMU_BUILD)
    if (ablate & 1) __builtin_amdgcn_s_setprio(1);
#endif
    const int pt = tid - 256;
    const int x_tile = pt >> 3, x_pair = pt & 7;
    const int ltx = x_tile & 7, lty = x_tile >> 3;
    const int64_t in_rows = (int64_t)p.n_img * p.hi * p.wi;
    const lfdm_buf buf0 = lfdm_make_buf(p.src0, (uint32_t)(((in_rows - 1) * p.ld0 + p.c0) * 4));
    // raw pieces of this thread: piece j = pt + 256 i = (pixel j >> 2, channel quad j & 3); physical pixel of the block's corner
    const int py0 = (UP ? 8 : 16) * blk_y - 1, px0 = (UP ? 16 : 32) * blk_x - 1;
    uint32_t rvoff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int j = pt + 256 * i, pix = j >> 2, quad = j & 3;
      const int pr = pix / RW, pc = pix - pr * RW;
      const int gy = py0 + pr, gx_ = px0 + pc;
      const bool ok = j < NPIECE && gy >= 0 && gy < p.hi && gx_ >= 0 && gx_ < p.wi;
      rvoff[i] = ok ? (uint32_t)(((((int64_t)blk_n * p.hi + gy) * p.wi + gx_) * p.ld0 + 4 * quad) * 4) : LFDM_BUF_OOB;
    }
    float4 rreg[NLD];
    auto load_raw = [&](int per) {
      const uint32_t cb = (uint32_t)per * (W4C * 4u);
#pragma unroll
      for (int i = 0; i < NLD; ++i) rreg[i] = lfdm_buf_load_f4(buf0, rvoff[i] == LFDM_BUF_OOB ? LFDM_BUF_OOB : rvoff[i] + cb);
    };
    auto store_raw = [&](float* R) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int j = pt + 256 * i;
        if (j < NPIECE) *reinterpret_cast<float4*>(R + (j & 3) * W4S_QS + (j >> 2) * 4) = rreg[i];
      }
    };
    // this thread's patch inside the raw tile: pixel (row0 + dr(r), col0 + dc(c)), its channel pair = quad x_pair >> 1, half x_pair & 1
    const int row0 = UP ? 2 * lty : 4 * lty, col0 = UP ? 2 * ltx : 4 * ltx;
    const int pbase = (x_pair >> 1) * W4S_QS + (row0 * RW + col0) * 4 + 2 * (x_pair & 1);
    f32x2 d[36];
    auto read_patch = [&](const float* R) {
#pragma unroll
      for (int q = 0; q < 36; ++q) {
        const int r = q / 6, c = q % 6;
        const int dr = UP ? (r + 1) >> 1 : r, dc = UP ? (c + 1) >> 1 : c;
        d[q] = *reinterpret_cast<const f32x2*>(R + pbase + (dr * RW + dc) * 4);
      }
    };
    auto transform = [&]() {      // B^T d B in place: columns, then rows
#pragma unroll
      for (int c = 0; c < 6; ++c)
        bt6(d[c], d[6 + c], d[12 + c], d[18 + c], d[24 + c], d[30 + c], d[c], d[6 + c], d[12 + c], d[18 + c], d[24 + c], d[30 + c]);
#pragma unroll
      for (int i = 0; i < 6; ++i)
        bt6(d[6 * i], d[6 * i + 1], d[6 * i + 2], d[6 * i + 3], d[6 * i + 4], d[6 * i + 5], d[6 * i], d[6 * i + 1], d[6 * i + 2], d[6 * i + 3],
            d[6 * i + 4], d[6 * i + 5]);
    };
    auto store_v = [&]() {
      float* dst = V + x_tile * LDV4 + 4 * ((x_pair >> 1) ^ ((x_tile >> 2) & 3)) + 2 * (x_pair & 1);      // swizzled 16-byte slot
#pragma unroll
      for (int q = 0; q < 36; ++q) *reinterpret_cast<f32x2*>(dst + q * (W4T * LDV4)) = d[q];
    };
    load_raw(0);
    store_raw(R0);
    load_raw(clampp(1));
    __syncthreads();                                  // raw(0) complete
    read_patch(R0);
    transform();
    store_v();                                        // V(0)
    store_raw(R1);                                    // raw(1)
    load_raw(clampp(2));
    __syncthreads();
#ifdef LFDM_W4_STAMP
    long long s_a1 = 0, s_a2 = 0, s_w1 = 0, s_b = 0, s_w2 = 0, s_t = clock64();
#define LFDM_W4S_LAP(acc) { const long long now_ = clock64(); acc += now_ - s_t; s_t = now_; }
#else
#define LFDM_W4S_LAP(acc)
#endif
    for (int k = 0; k < nper; ++k) {
      // phase A (the consumers multiply V(k))
      read_patch((k & 1) ? R0 : R1);                  // raw(k+1)
      transform();
      LFDM_W4S_LAP(s_a1)
      store_raw((k & 1) ? R1 : R0);                   // raw(k+2) -> the buffer raw(k) has left
      load_raw(clampp(k + 3));
      LFDM_W4S_LAP(s_a2)
      __syncthreads();
      LFDM_W4S_LAP(s_w1)
      store_v();                                      // phase B: V(k+1)
      LFDM_W4S_LAP(s_b)
      __syncthreads();
      LFDM_W4S_LAP(s_w2)
    }
#ifdef LFDM_W4_STAMP
    if (lane == 0 && blockIdx.x < 256) {
      long long* st = (long long*)p.gn_in_gamma + ((int64_t)blockIdx.x * 8 + wave) * 8;
      st[0] = s_a1; st[1] = s_a2; st[2] = s_w1; st[3] = s_b; st[4] = s_w2;
    }
#endif
    __syncthreads();                                  // the consumers' epilogue: one more barrier
    return;
  }

  // ---------------------------------------------------------------- CONSUMERS (as in conv_wino4_kernel, one V buffer)
  const lfdm_buf bufw = lfdm_make_buf(p.weight_wino4, (uint32_t)((int64_t)36 * nch * p.coutp * W4K * 4));
  const int cw = wave;
  const int ncol = n0 + l31;
  float4 bfr[3][3];
  const uint32_t chunk_bytes = (uint32_t)p.coutp * (W4K * 4u);
  const uint32_t pos_bytes = (uint32_t)nch * chunk_bytes;
  const uint32_t b_lane = ((uint32_t)ncol * W4K + 4u * kh) * 4u;
  auto fetch_bg = [&](float4 (&dst)[3], int g, int chunk) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const uint32_t uni = (uint32_t)(9 * cw + 3 * g + u) * pos_bytes + (uint32_t)chunk * chunk_bytes;
      dst[u] = lfdm_buf_load_f4(bufw, uni + b_lane);
    }
  };
  f32x16 acc[9];
#pragma unroll
  for (int q = 0; q < 9; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const int a_swz = (l31 >> 2) & 3;
  auto load_a = [&](float4 (&a)[3], int half, int g) {
#pragma unroll
    for (int u = 0; u < 3; ++u)
      a[u] = *reinterpret_cast<const float4*>(V + ((9 * cw + 3 * g + u) * W4T + l31) * LDV4 + 4 * ((2 * half + kh) ^ a_swz));
  };
  auto consume = [&](int half, int chunk, int next_chunk) {
    float4 a[2][3];
    load_a(a[0], half, 0);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      if (g < 2) load_a(a[(g + 1) & 1], half, g + 1);
      if (g == 0) fetch_bg(bfr[2], 2, chunk);
      else fetch_bg(bfr[g - 1], g - 1, next_chunk);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const float4 bq = bfr[g][u], aq = a[g & 1][u];
          const float av = e == 0 ? aq.x : e == 1 ? aq.y : e == 2 ? aq.z : aq.w;
          const float bv = e == 0 ? bq.x : e == 1 ? bq.y : e == 2 ? bq.z : bq.w;
          acc[3 * g + u] = mfma_32x32x2(av, bv, acc[3 * g + u]);
        }
#if !defined(LFDM_EMU_BUILD)
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };
  fetch_bg(bfr[0], 0, 0);
  fetch_bg(bfr[1], 1, 0);
  __syncthreads();
  __syncthreads();
#ifdef LFDM_W4_STAMP
  long long s_a1 = 0, s_w1 = 0, s_w2 = 0, s_t = clock64();
#endif
  for (int k = 0; k < nper; ++k) {
    consume(0, 2 * k, 2 * k + 1);
    consume(1, 2 * k + 1, clampc(2 * k + 2));
    LFDM_W4S_LAP(s_a1)
    __syncthreads();
    LFDM_W4S_LAP(s_w1)
    __syncthreads();
    LFDM_W4S_LAP(s_w2)
  }
#ifdef LFDM_W4_STAMP
  if (lane == 0 && blockIdx.x < 256) {
    long long* st = (long long*)p.gn_in_gamma + ((int64_t)blockIdx.x * 8 + wave) * 8;
    st[0] = s_a1; st[2] = s_w1; st[4] = s_w2;
  }
#endif

  // ---------------------------------------------------------------- output transform A^T M A through LDS (as in conv_wino4_kernel)
  float* const Ms = smem;
#pragma unroll
  for (int q = 0; q < 9; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tile = (r & 3) + 8 * (r >> 2) + 4 * kh;
      Ms[((9 * cw + q) * W4T + tile) * W4N + l31] = acc[q][r];
    }
  const int e_quad = tid & 7, e_tile = tid >> 3;
  const int my_n = blk_n, my_ty = 4 * blk_y + (e_tile >> 3), my_tx = 8 * blk_x + (e_tile & 7);
  const int co = n0 + 4 * e_quad;
  const bool live = co < p.cout;
  const int64_t orow0 = ((int64_t)my_n * p.hq + 4 * my_ty) * p.wq + 4 * my_tx;
  float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live && p.bias) bb = *reinterpret_cast<const float4*>(p.bias + co);
  float4 res[16];
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    res[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && p.residual) res[o] = *reinterpret_cast<const float4*>(p.residual + (orow0 + (o >> 2) * p.wq + (o & 3)) * p.ldr + co);
  }
  __syncthreads();
  if (!live) return;
  float4 T[6][4];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float4 m[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) m[j] = *reinterpret_cast<const float4*>(Ms + ((6 * i + j) * W4T + e_tile) * W4N + 4 * e_quad);
#define LFDM_W4_COL(f)                                                                                          \
    {                                                                                                           \
      const float s12 = m[1].f + m[2].f, d12 = m[1].f - m[2].f, s34 = m[3].f + m[4].f, d34 = m[3].f - m[4].f;  \
      T[i][0].f = m[0].f + s12 + s34;                                                                           \
      T[i][1].f = d12 + 2.0f * d34;                                                                             \
      T[i][2].f = s12 + 4.0f * s34;                                                                             \
      T[i][3].f = d12 + 8.0f * d34 + m[5].f;                                                                    \
    }
    LFDM_W4_COL(x) LFDM_W4_COL(y) LFDM_W4_COL(z) LFDM_W4_COL(w)
#undef LFDM_W4_COL
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    float4 y[4];
#define LFDM_W4_ROW(f)                                                                                                            \
    {                                                                                                                             \
      const float s12 = T[1][b].f + T[2][b].f, d12 = T[1][b].f - T[2][b].f, s34 = T[3][b].f + T[4][b].f, d34 = T[3][b].f - T[4][b].f; \
      y[0].f = T[0][b].f + s12 + s34;                                                                                             \
      y[1].f = d12 + 2.0f * d34;                                                                                                  \
      y[2].f = s12 + 4.0f * s34;                                                                                                  \
      y[3].f = d12 + 8.0f * d34 + T[5][b].f;                                                                                      \
    }
    LFDM_W4_ROW(x) LFDM_W4_ROW(y) LFDM_W4_ROW(z) LFDM_W4_ROW(w)
#undef LFDM_W4_ROW
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int64_t orow = orow0 + a * p.wq + b;
      const float4 rr = res[4 * a + b];
      float4 v = make_float4(y[a].x + bb.x + rr.x, y[a].y + bb.y + rr.y, y[a].z + bb.z + rr.z, y[a].w + bb.w + rr.w);
      if (ACT) {
        v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act);
        v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
      }
      *reinterpret_cast<float4*>(p.out + orow * p.ldo + co) = v;
    }
  }
}

// U = G g G^T (6x6 per filter), one thread per (input channel k, output channel n)
__global__ __launch_bounds__(256) void pack_wino4_kernel(const float* __restrict__ w, int ld_o, int cout, int cin, int coutp,
                                                         float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)cin * coutp) return;
  const int n = (int)(idx % coutp), k = (int)(idx / coutp);
  float g[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t] = n < cout ? w[(int64_t)n * ld_o + (int64_t)k * 9 + t] : 0.f;
  // rows of G: [1/4 0 0], [-1/6 -1/6 -1/6], [-1/6 1/6 -1/6], [1/24 1/12 1/6], [1/24 -1/12 1/6], [0 0 1]
  auto g6 = [](float a, float b, float c, float (&o)[6]) {
    o[0] = 0.25f * a;
    o[1] = (-1.0f / 6.0f) * (a + b + c);
    o[2] = (-1.0f / 6.0f) * (a - b + c);
    o[3] = (1.0f / 24.0f) * a + (1.0f / 12.0f) * b + (1.0f / 6.0f) * c;
    o[4] = (1.0f / 24.0f) * a - (1.0f / 12.0f) * b + (1.0f / 6.0f) * c;
    o[5] = c;
  };
  float r[6][3];                                        // G g
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    float o[6];
    g6(g[b], g[3 + b], g[6 + b], o);
#pragma unroll
    for (int i = 0; i < 6; ++i) r[i][b] = o[i];
  }
  const int nch = cin / W4K;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float u[6];
    g6(r[i][0], r[i][1], r[i][2], u);
#pragma unroll
    for (int j = 0; j < 6; ++j) out[((((int64_t)(6 * i + j)) * nch + k / W4K) * coutp + n) * W4K + (k % W4K)] = u[j];
  }
}

}  // namespace

extern "C" int lfdm_pack_wino4_weight_f32(const float* w, int ld_o, int cout, int cin, int coutp, float* out, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!w || !out || cout <= 0 || cin <= 0 || cin % W4K != 0 || coutp < cout || coutp % 32 != 0 || ld_o < cin * 9) {
    lfdm_set_error("pack_wino4_weight: input channels must be a multiple of 8 and coutp a multiple of 32 >= the output channels");
    return LFDM_EINVAL;
  }
  const int64_t total = (int64_t)cin * coutp;
  LFDM_LAUNCH(pack_wino4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, ld_o, cout, cin, coutp, out);
  return lfdm_check_launch("pack_wino4_weight");
}

// 1-D grid of (tile blocks rounded up to 8) x column tiles workgroups.  Called by lfdm_conv2d_cl_f32 (conv_igemm.hip) for schedule 4.
int lfdm_conv_wino4_launch(const lfdm_conv_params& p, hipStream_t stream) {
  const int64_t ntiles = (int64_t)p.n_img * (p.hq / 4) * (p.wq / 4);
  const int gx = (int)((ntiles + W4T - 1) / W4T), ny = (p.coutp + W4N - 1) / W4N;
  const dim3 grid((unsigned)(((gx + 7) / 8) * 8 * ny));
  // producer-wave priority (kernel argument `ablate`, bit 0): on by default, LFDM_W4_PRIO=0 switches it off (A/B: profiles/r04_w_wino4_stamps.txt)
  static const int ablate = []() { const char* e = lfdm_knob("LFDM_W4_PRIO"); return e ? atoi(e) : 1; }();
  // staged flavour (unique pixels of an 8 x 4 tile block through LDS): whenever the tile grid allows; LFDM_W4_STAGED=0 keeps the direct one
  static const bool staged_on = []() { const char* e = lfdm_knob("LFDM_W4_STAGED"); return !e || atoi(e) != 0; }();
  const bool act = p.act != LFDM_ACT_NONE;
  if (staged_on && (p.wq / 4) % 8 == 0 && (p.hq / 4) % 4 == 0) {
    if (p.upsample) {
      if (act) LFDM_LAUNCH((conv_wino4s_kernel<true, true>), grid, dim3(512), 0, stream, p, gx, ny, ablate);
      else LFDM_LAUNCH((conv_wino4s_kernel<false, true>), grid, dim3(512), 0, stream, p, gx, ny, ablate);
    } else {
      if (act) LFDM_LAUNCH((conv_wino4s_kernel<true, false>), grid, dim3(512), 0, stream, p, gx, ny, ablate);
      else LFDM_LAUNCH((conv_wino4s_kernel<false, false>), grid, dim3(512), 0, stream, p, gx, ny, ablate);
    }
    return lfdm_check_launch("conv_wino4s");
  }
  if (act) LFDM_LAUNCH((conv_wino4_kernel<true>), grid, dim3(512), 0, stream, p, gx, ny, ablate);
  else LFDM_LAUNCH((conv_wino4_kernel<false>), grid, dim3(512), 0, stream, p, gx, ny, ablate);
  return lfdm_check_launch("conv_wino4");
}
