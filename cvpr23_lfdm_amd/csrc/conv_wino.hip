// Winograd F(2x2, 3x3) schedule of the 3x3 / stride-1 / zero-padded convolution on fp32 MFMA
// (include/lfdm_hip.h: lfdm_conv2d_cl_f32 with lfdm_conv_params.weight_wino; LFDM_WINO=0 forces the direct kernels).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A   per 4x4 input patch d / 2x2 output tile Y and (c_in, c_out) pair:
// 16 independent "frequency positions", each an ordinary GEMM over c_in: M[pos][tile][co] = sum_ci V[pos][tile][ci] U[pos][ci][co]
// -> 16/36 of the multiplications of the direct form.  One workgroup owns 32 output tiles (128 pixels) x 32 output channels;
// per 16-channel chunk every thread transforms one (tile, channel pair) patch (B^T d B) into LDS, then wave i runs the MFMAs
// of position row i (positions 4i..4i+3; A operand = V from LDS, B operand = the pre-transformed weights straight from global
// memory in operand order, each fragment re-loaded for the next chunk as soon as its MFMAs are issued).  The output
// transform A^T M A is split: the sum over the position column j happens in the wave's accumulator registers, the sum over
// the row i (across waves) through 8 LDS planes; it feeds the same epilogue as the direct kernels (bias, GroupNorm partial
// sums, residual, activation) or the split-K slabs.  ~150 VGPRs and 43 KB LDS: three workgroups per CU, so one
// workgroup's transform overlaps the others' MFMAs.  The same kernel reads its input through a virtual nearest x2 upsample
// (UpBlock2d) and, with the data-gradient form of the filters, computes dX of the training step.
#include <cstdio>
#include <cstdlib>

#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

constexpr int WT = 32;            // tiles per workgroup
constexpr int WN = 32;            // output channels per column tile (NT of them per workgroup)
constexpr int WKC = 16;           // input channels per chunk
constexpr int LDV = WKC + 4;      // LDS row stride of V
constexpr int LDM = WN + 4;       // LDS row stride of the half-transformed M planes in the epilogue (16-byte aligned rows: ds_read_b128)

// Operands are prefetched one chunk ahead; ~150 VGPRs / 43 KB LDS -> three workgroups per CU hide each other's phases.
// Per chunk and CU the vector-memory path moves 32 KB of patches + 32 KB of weight fragments for 2048 matrix-pipe cycles
// per SIMD; rocprofv3 (profiles/r01_n_wino_pmc.txt) shows ~2000-2500 L1 accesses per chunk and the busiest TA ~40 % busy:
// the kernel is co-limited by the L1 path, which is why the variants below changed nothing.  Measured and removed (no
// faster on any shape of tools/bench_conv.py): double-buffered LDS with the next chunk's transform placed between the
// positions' MFMAs; patches and weight fragments fetched two chunks ahead at two workgroups per CU; A-fragment LDS reads
// pinned one position ahead with sched_group_barrier.  What would help next: 32-channel chunks with 16-byte patch loads
// (half the TA cycles per patch byte), raw pixels staged once in LDS (the patches overlap 4x).
// NT = column tiles per workgroup: NT = 2 (64 columns; chosen by the plan for launches of >= 1536 such workgroups, i.e. the batched
// shapes of training / throughput mode - conv_igemm.hip make_plan, LFDM_WINO_BN64_MIN) halves the patch loads / transforms per MFMA at
// two workgroups per CU (244 VGPRs): 2-5 % faster on the large-M decoder shapes, slower wherever it leaves a CU fewer than
// ~3 workgroups (profiles/r01_o_conv_shapes_wino_bn64.txt).
// Measured and removed in round 2 (tools/sweep_conv.sh, profiles/r02_a_sweep_conv.txt): staging the unique pixels of each tile-row
// band in LDS with 16-byte loads (3-25 % SLOWER on every shape) and a 64-tile / 8-channel-chunk workgroup (conv_wino_wide.hip,
// 10-30 % slower): the L1 path is not what limits this kernel - per-workgroup phase stamps (tools/probe_wino_phases.py) show the
// K loop AT the matrix-pipe bound whenever three workgroups share a CU; the time is in the set-up, the first patch's latency
// and the epilogue, which all co-resident workgroups go through in lockstep.
// POOL: lfdm_conv_params.pool2 - an instantiation of its own, so that the plain kernel keeps its register allocation.
// Measured and removed in round 2: ResBlock2d's pre-activation BatchNorm + ReLU applied to the patches right before the transform
// (tables in LDS, no spills): the 256 -> 256 bottleneck convolution of a B = 8 training step went from 1650 to 1781 us, more
// than the 112 us streaming pass it replaced - this K loop has no idle VALU slots (profiles/r02_ab_*).  The same activation
// written as a SECOND OUTPUT of the producing convolution's epilogue (one more float4 store per pixel) cost 103 us per launch
// for the same 112 us pass: no gain either, removed (profiles/r02_ac_*).
// Built, measured and REMOVED in round 6 (records: HISTORY.md rounds 4-6): K groups inside a workgroup (G x 4 waves on interleaved chunks,
// accumulators merged through LDS: neutral, profiles/r04_b_bench_wino_kg.txt) and the input GroupNorm + SiLU applied to the patches on their
// way into the transform (one launch less per ResnetBlock, ~10 us more per convolution: profiles/r04_d_*).
// FUSE (round 5, lfdm_conv_params.tile_counters): split-K without a reduce launch.  Every slice's workgroup stores its 128 x 32 slab tile as
// 8-byte agent-scope words (written through to memory: no cache-wide release), takes a ticket of its output tile, and the workgroup that draws
// the tile's last ticket reads the ksplit slabs back past the non-coherent L2s (agent-scope loads: no acquire / invalidate), sums them in
// slice order - bit-identical to conv_splitk_reduce_vec_kernel, whoever arrives last - and runs the epilogue (bias, GroupNorm partial sums,
// residual, activation) itself.  The fence-based form of this hand-off (round 2, KSW schedule) measured neutral: its release wrote back the
// XCD's whole L2 from every workgroup's tail - the cost found in the BatchNorm reduce (profiles/r05_p_bench_bn.txt, r05_q_*).
template <bool ACT, int NT, bool POOL = false, bool FUSE = false>
__global__ __launch_bounds__(256, NT == 1 ? 3 : 2) void conv_wino_kernel(lfdm_conv_params p, int bal_whole) {
  static_assert(!FUSE || (NT == 1 && !POOL), "in-launch split-K reduction: the plain 32-column workgroup only");
  constexpr int WNB = WN * NT;      // output channels per workgroup
  constexpr int LD = LDV;
  constexpr int VSZ = 16 * WT * LD;
  __shared__ __attribute__((aligned(16))) float smem[VSZ];   // V during the loop; >= 8*WT*LDM for the epilogue planes
  static_assert(VSZ >= 8 * WT * LDM, "epilogue planes must fit in the V buffer");
  __shared__ float s_gn[2][4][WNB];

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
#ifdef LFDM_WINO_TIMING
  // probe build only (tools/probe_wino_phases.py): cycle stamps of every workgroup go behind the first 64 K words of p.tile_counters
  unsigned long long tstamp[8];
  tstamp[0] = __builtin_readcyclecounter();
  const unsigned long long wall0 = wall_clock64();
#endif
  // p.upsample: the input is read through a virtual nearest x2 upsample (UpBlock2d, LFAE util.py:120-133): the logical
  // image is (hq, wq) = (2*hi, 2*wi) and logical pixel (y, x) is physical (y >> 1, x >> 1)
  const int up = p.upsample ? 1 : 0;
  const int th = p.hq >> 1, tw = p.wq >> 1;                 // tiles per image (hq, wq even: host check)
  const unsigned ntiles = (unsigned)p.n_img * th * tw;       // < 2^31 / 4 (host check on the pixel count)
  // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in linear-id order.  The
  // default order puts the column tiles of one tile block on one XCD (gridDim.x % 8 == 0: they share the input patches).
  // Where the Winograd filters outweigh the input (16*coutp*cin vs pixels*cin floats: the 8x8 / 4x4 levels) it is the
  // filter slice that must not be fetched into all eight L2s: XCD k then owns the column tiles k, k+8, ...
  // BALANCED launch (round 6, FUSE only, bal_whole > 0): a launch of 640 (tile, K slice) jobs puts three workgroups on 128 CUs and two on the other 128
  // (the dispatcher deals ids i, i + 256, i + 512 to one CU: tools/probe_wino_phases.py --placement), and the CUs with three set the launch's time.  Here
  // the grid is ONE dimension of bal_whole + 2 * (jobs - bal_whole) = 768 ids: ids < bal_whole (512) run a whole job, the others HALF the K range of one
  // of the remaining 128 jobs - every CU gets two whole jobs and one half (10 chunk units instead of 12 / 8).  A halved job adds one slab to its tile; the
  // tile's ticket counts ksplit + (halved slices of the tile) arrivals.  v = the id the job would have had in the (gx, gy, ksplit) grid, and
  // v = id (mod 8): a job stays on the XCD the tile order below wants it on.
  unsigned gx = gridDim.x, gy = gridDim.y;
  unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  int half = -1;                                         // 0 / 1: this workgroup runs the first / second half of its job's chunks
  if (FUSE && bal_whole > 0) {
    gx = (unsigned)(((int64_t)p.n_img * (p.hq >> 1) * (p.wq >> 1) + WT - 1) / WT);
    gy = (unsigned)((p.coutp + WNB - 1) / WNB);
    if (L >= (unsigned)bal_whole) {
      const unsigned t = L - (unsigned)bal_whole, q = t >> 3;
      L = (unsigned)bal_whole + 8u * (q >> 1) + (t & 7u);
      half = (int)(q & 1u);
    }
  }
  const unsigned T = gx * gy;                            // tiles of one K slice
  unsigned bx = L % gx, by = (L / gx) % gy, bz = L / T;
  if ((gy & 7u) == 0 && 16ll * p.coutp > (int64_t)p.n_img * p.hi * p.wi) {     // (physical input pixels)
    const unsigned j = L >> 3, ny8 = gy >> 3;
    by = (L & 7u) + 8u * (j % ny8);
    const unsigned rest = j / ny8;
    bx = rest % gx;
    bz = rest / gx;
  }
  const unsigned t0 = bx * WT;
  const int n0 = by * WNB;
  // grouped convolution (lfdm_conv_params.groups): this workgroup's column tile lies inside ONE group (cout/groups % 32 == 0,
  // NT = 1): it reduces over that group's c0/groups input channels only, with the group's own filter pack
  const int ngroups = p.groups > 1 ? p.groups : 1;
  const int wcoutp = ngroups > 1 ? p.cout / ngroups : p.coutp;      // columns of one filter pack
  const int grp = ngroups > 1 ? n0 / wcoutp : 0;
  const int cin = (p.c0 + p.c1) / ngroups;
  const int cbase = grp * cin;                                       // first input channel of the group
  const int nchunks_all = cin / WKC;
  const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
  int kc_begin = (int)((int64_t)nchunks_all * bz / ksplit);
  int kc_end = (int)((int64_t)nchunks_all * (bz + 1) / ksplit);
  // slabs of this workgroup's tile and the one it writes: ksplit, or (balanced) one more per halved slice of the tile - slice z of the tile whose
  // position inside a slice layer is tl has id z * T + tl, and ids >= bal_whole are the halved ones
  int nslab = ksplit, my_slab = (int)bz;
  if (FUSE && bal_whole > 0) {
    const int tl = (int)(L % T);
    const int z_first = tl >= bal_whole ? 0 : (bal_whole - tl + (int)T - 1) / (int)T;      // the tile's first halved slice
    if (z_first < ksplit) nslab = ksplit + (ksplit - z_first);
    if (half >= 0) {
      const int mid = (kc_begin + kc_end) >> 1;
      if (half == 0) kc_end = mid;
      else { kc_begin = mid; my_slab = ksplit + ((int)bz - z_first); }
    }
  }
  const int64_t M = (int64_t)p.n_img * p.hq * p.wq;

  // every thread derives the coordinates of ITS tile (tid >> 3 - the same tile in the input transform and in the epilogue)
  // itself: no LDS table, no barrier before the first loads are issued
  int my_n = -1, my_ty = 0, my_tx = 0;
  {
    const unsigned t = t0 + (tid >> 3);
    if (t < ntiles) {
      my_n = (int)(t / (unsigned)(th * tw));
      const unsigned rem = t - (unsigned)my_n * (th * tw);
      my_ty = (int)(rem / (unsigned)tw);
      my_tx = (int)(rem - (unsigned)my_ty * tw);
    }
  }

  // ---- input transform: one (tile, channel pair) patch per thread ----
  const int x_tile = tid >> 3, x_c2 = tid & 7;              // 32 tiles x 8 channel pairs = 256 threads
  const int64_t in_rows = (int64_t)p.n_img * p.hi * p.wi;
  const lfdm_buf buf0 = lfdm_make_buf(p.src0, (uint32_t)(((in_rows - 1) * p.ld0 + p.c0) * 4));
  const lfdm_buf buf1 = p.c1 > 0 ? lfdm_make_buf(p.src1, (uint32_t)(((in_rows - 1) * p.ld1 + p.c1) * 4)) : buf0;
  // 32-bit byte offsets (the plan guarantees rows * ld * 4 < 2^32): off = base + uniform per-(py,px) delta + chunk; `base`
  // may wrap for the patch corner outside the image - those taps are masked to the out-of-range offset anyway
  uint32_t base0 = 0, base1 = 0;     // byte offset of the patch's (0,0) corner + this thread's channel pair, per source
  unsigned valid_mask = 0;           // bit (py*4+px): patch pixel inside the image
  if (my_n >= 0) {
    const int n = my_n, ty = my_ty, tx = my_tx;
    // physical pixel of the patch corner: logical (2ty-1, 2tx-1); through the upsample that is (ty-1, tx-1)
    const uint32_t pix = up ? (uint32_t)((n * p.hi + ty - 1) * p.wi + tx - 1)
                            : (uint32_t)((n * p.hi + 2 * ty - 1) * p.wi + 2 * tx - 1);
    base0 = (pix * (uint32_t)p.ld0 + 2u * x_c2) * 4u;
    base1 = (pix * (uint32_t)p.ld1 + 2u * x_c2) * 4u;
    const unsigned rows = 0xFu & ~(ty == 0 ? 1u : 0u) & ~(ty == th - 1 ? 8u : 0u);     // only the first / last patch row or
    const unsigned cols = 0xFu & ~(tx == 0 ? 1u : 0u) & ~(tx == tw - 1 ? 8u : 0u);     // column can fall outside
#pragma unroll
    for (int py = 0; py < 4; ++py)
      if ((rows >> py) & 1u) valid_mask |= cols << (4 * py);
  }
  // the epilogue's bias (unsplit tiles add it themselves) is requested HERE, under the whole K loop, not at the epilogue's start
  float4 bias_pre[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    const int co = n0 + WN * ct + 4 * (tid & 7);
    bias_pre[ct] = (NT == 1 && p.bias && nslab == 1 && co < p.cout) ? *reinterpret_cast<const float4*>(p.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  }                                    // (NT = 2 is at its register limit: it reads the bias in the epilogue as before)
  float2 patch[16];
  auto fetch_patch = [&](float2 (&patch)[16], int chunk, unsigned vmask) {
    int cc = chunk * WKC + cbase;
    const bool second = cc >= p.c0;
    if (second) cc -= p.c0;
    const lfdm_buf buf = second ? buf1 : buf0;
    const uint32_t ld4 = (uint32_t)(second ? p.ld1 : p.ld0) * 4u;
    const uint32_t base = (second ? base1 : base0) + (uint32_t)cc * 4u;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      // uniform (scalar registers); upsampled: logical rows 2ty-1..2ty+2 are physical rows ty-1, ty, ty, ty+1
      const int py = up ? ((q >> 2) + 1) >> 1 : (q >> 2), px = up ? ((q & 3) + 1) >> 1 : (q & 3);
      const uint32_t delta = (uint32_t)(py * p.wi + px) * ld4;
      patch[q] = lfdm_buf_load_f2(buf, ((vmask >> q) & 1u) ? base + delta : LFDM_BUF_OOB);
    }
  };
  // ---- weight fragments: lane (co = n0 + l31, k-slot kh) holds U[pos][16*chunk + 8*kh + s][co], s = 0..7 ----
  const lfdm_buf bufw = lfdm_make_buf(p.weight_wino + (int64_t)grp * 16 * nchunks_all * wcoutp * WKC,
                                      (uint32_t)((int64_t)16 * nchunks_all * wcoutp * WKC * 4));
  float4 bfrag[4][NT][2];
  auto fetch_b = [&](int pi, int chunk) {
    const int pos = 4 * wave + pi;
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      const int n = n0 + WN * ct + l31 - grp * wcoutp;          // column inside the (group's) filter pack
      // pack [pos][chunk][j = 0, 1][column][kh][4] (round 4): the lanes of ONE load instruction (32 columns x 2 k-slots x 16 bytes) read a
      // contiguous 1 KB - the previous [pos][chunk][column][16] order made each instruction touch half of every lane's 32-byte piece,
      // twice the L1 line look-ups per byte on the kernel's dominant stream
      const uint32_t off = (n < wcoutp)
                               ? (uint32_t)((((((int64_t)pos * nchunks_all + chunk) * 2) * wcoutp + n) * 2 + kh) * 16)
                               : LFDM_BUF_OOB;
      bfrag[pi][ct][0] = lfdm_buf_load_f4(bufw, off);
      bfrag[pi][ct][1] = lfdm_buf_load_f4(bufw, off == LFDM_BUF_OOB ? LFDM_BUF_OOB : off + (uint32_t)wcoutp * 32u);
    }
  };

  f32x16 acc[4][NT];
#pragma unroll
  for (int pi = 0; pi < 4; ++pi)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pi][ct][r] = 0.f;

  // B^T d B on a channel pair: part i = row i of the position grid (positions 4i..4i+3)
  auto xform_part = [&](const float2 (&d)[16], int i, float* V) {
    float2 r[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float2 d0 = d[c], d1 = d[4 + c], d2 = d[8 + c], d3 = d[12 + c];
      r[c] = i == 0 ? make_float2(d0.x - d2.x, d0.y - d2.y)
           : i == 1 ? make_float2(d1.x + d2.x, d1.y + d2.y)
           : i == 2 ? make_float2(d2.x - d1.x, d2.y - d1.y)
                    : make_float2(d1.x - d3.x, d1.y - d3.y);
    }
    float* dst = V + ((4 * i) * WT + x_tile) * LD + 2 * x_c2;
    *reinterpret_cast<float2*>(dst) = make_float2(r[0].x - r[2].x, r[0].y - r[2].y);
    *reinterpret_cast<float2*>(dst + WT * LD) = make_float2(r[1].x + r[2].x, r[1].y + r[2].y);
    *reinterpret_cast<float2*>(dst + 2 * WT * LD) = make_float2(r[2].x - r[1].x, r[2].y - r[1].y);
    *reinterpret_cast<float2*>(dst + 3 * WT * LD) = make_float2(r[1].x - r[3].x, r[1].y - r[3].y);
  };
  auto load_a = [&](const float* V, int pi, float4& a0, float4& a1) {        // A fragment of position 4*wave + pi
    const float* va = V + ((4 * wave + pi) * WT + l31) * LD + 8 * kh;
    a0 = *reinterpret_cast<const float4*>(va);
    a1 = *reinterpret_cast<const float4*>(va + 4);
  };
  auto mfma_pos = [&](const float4& a0, const float4& a1, int pi) {           // its 8 k-steps on this chunk (NT chains)
    const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        const float4 bq = bfrag[pi][ct][s8 >> 2];
        const float b = (s8 & 3) == 0 ? bq.x : (s8 & 3) == 1 ? bq.y : (s8 & 3) == 2 ? bq.z : bq.w;
        acc[pi][ct] = mfma_32x32x2(a[s8], b, acc[pi][ct]);
      }
  };
  const int kc_last = kc_end - 1;
  auto clampc = [&](int c) { return c < kc_last ? c : kc_last; };     // re-fetching the last chunk is harmless

  float* const Vs = smem;                        // [16 pos][WT tiles][LD]
#ifdef LFDM_WINO_TIMING
  tstamp[1] = __builtin_readcyclecounter();
#endif
  const int rounds = kc_end - kc_begin;
  const int kc0 = kc_begin;
  {
    // (patch BEFORE the filter fragments, as inside the loop: loads retire in order, and with the same order on both ways into the loop
    // header the wait for the patch there is vmcnt(8) - the eight fragment loads stay in flight under the input transform.  With the
    // fragments first the header got vmcnt(0): every round then waited for the fragment loads issued at the very end of the round before.)
    fetch_patch(patch, clampc(kc0), valid_mask);
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) fetch_b(pi, clampc(kc0));
    for (int r = 0; r < rounds; ++r) {
      const int nxt = clampc(kc0 + r + 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) xform_part(patch, i, Vs);
#ifdef LFDM_WINO_TIMING
      if (r == 0) tstamp[2] = __builtin_readcyclecounter();     // first patch arrived + transformed
#endif
      __syncthreads();
      fetch_patch(patch, nxt, valid_mask);     // in flight under this round's MFMAs
#pragma unroll
      for (int pi = 0; pi < 4; ++pi) {
        float4 a0, a1;
        load_a(Vs, pi, a0, a1);
        mfma_pos(a0, a1, pi);
        fetch_b(pi, nxt);                                        // refilled in place for the next round
      }
      __syncthreads();
    }
  }

#ifdef LFDM_WINO_TIMING
  tstamp[3] = __builtin_readcyclecounter();
#endif
  // ---- output transform A^T M A: the column sum (over j) in registers, the row sum (over i = wave) through LDS ----
  // Read side: thread = (tile, 4 consecutive output channels): 8 ds_read_b128, then the 2x2 output pixels as float4 stores
  // (8 lanes cover a pixel's 128-byte row segment) - 4x fewer LDS / global instructions than one channel per lane
  // (epilogue 4.7 -> measured in profiles/r02_*).  The plan only selects this schedule when float4 accesses are legal.
  float* const Ms = smem;                        // [8 = 2*i + j'][WT][LDM], one column tile at a time
  // split-K slabs [ksplit][M][coutp] through a buffer descriptor with 32-bit byte offsets (splitk_fused checks the size)
  const lfdm_buf slab_buf = lfdm_make_buf(FUSE && nslab > 1 ? p.partial : nullptr, FUSE && nslab > 1 ? (uint32_t)((int64_t)nslab * M * p.coutp * 4) : 0u);
  const int e_tile = tid >> 3, e_c4 = tid & 7;
  float gs[NT][4], gq[NT][4];
  float4 yown[4];                    // (FUSE) this workgroup's own contribution to its thread's four pixels: the reducer does not read its own slab back
#pragma unroll
  for (int q = 0; q < 4; ++q) yown[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    if (ct > 0) __syncthreads();    // the previous column tile's planes have been consumed
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tile = (r & 3) + 8 * (r >> 2) + 4 * kh;
      Ms[((2 * wave) * WT + tile) * LDM + l31] = acc[0][ct][r] + acc[1][ct][r] + acc[2][ct][r];
      Ms[((2 * wave + 1) * WT + tile) * LDM + l31] = acc[1][ct][r] - acc[2][ct][r] - acc[3][ct][r];
    }
    const int co = n0 + WN * ct + 4 * e_c4;
#pragma unroll
    for (int e = 0; e < 4; ++e) gs[ct][e] = gq[ct][e] = 0.f;
    float4 bb = bias_pre[ct];
    if (NT > 1 && p.bias && nslab == 1 && co < p.cout) bb = *reinterpret_cast<const float4*>(p.bias + co);
    const int n = my_n;                                 // e_tile == x_tile == tid >> 3
    const int64_t orow0 = ((int64_t)n * p.hq + 2 * my_ty) * p.wq + 2 * my_tx;
    const bool live = n >= 0 && co < p.coutp;
    float4 res[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                        // residual rows requested before the barrier
      res[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live && nslab == 1 && p.residual && co < p.cout)
        res[q] = *reinterpret_cast<const float4*>(p.residual + (orow0 + (q >> 1) * p.wq + (q & 1)) * p.ldr + co);
    }
    __syncthreads();
    if (live) {
      float4 m[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) m[q] = *reinterpret_cast<const float4*>(Ms + (q * WT + e_tile) * LDM + 4 * e_c4);
      float4 y[4];                   // y[i'][j'] = sum_i A^T[i'][i] T[i][j'],  m[2*i + j'] = T[i][j']
      y[0] = make_float4(m[0].x + m[2].x + m[4].x, m[0].y + m[2].y + m[4].y, m[0].z + m[2].z + m[4].z, m[0].w + m[2].w + m[4].w);
      y[1] = make_float4(m[1].x + m[3].x + m[5].x, m[1].y + m[3].y + m[5].y, m[1].z + m[3].z + m[5].z, m[1].w + m[3].w + m[5].w);
      y[2] = make_float4(m[2].x - m[4].x - m[6].x, m[2].y - m[4].y - m[6].y, m[2].z - m[4].z - m[6].z, m[2].w - m[4].w - m[6].w);
      y[3] = make_float4(m[3].x - m[5].x - m[7].x, m[3].y - m[5].y - m[7].y, m[3].z - m[5].z - m[7].z, m[3].w - m[5].w - m[7].w);
      if (POOL) {
        // DownBlock2d: conv -> act -> 2x2 average pool = the mean of this tile's four outputs: one row of the half-size image
        if (co < p.cout) {
          float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float4 v = make_float4(y[q].x + bb.x, y[q].y + bb.y, y[q].z + bb.z, y[q].w + bb.w);
            if (ACT) {
              v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act);
              v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
            }
            acc4.x += v.x; acc4.y += v.y; acc4.z += v.z; acc4.w += v.w;
          }
          const int64_t prow = ((int64_t)n * th + my_ty) * tw + my_tx;
          *reinterpret_cast<float4*>(p.out + prow * p.ldo + co) =
              make_float4(0.25f * acc4.x, 0.25f * acc4.y, 0.25f * acc4.z, 0.25f * acc4.w);
        }
        continue;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t orow = orow0 + (q >> 1) * p.wq + (q & 1);
        if (nslab > 1) {
          if (FUSE) {      // ONE 16-byte write-through store per float4 (round 6; two 8-byte agent-scope stores before: 2.7x the fabric time per byte)
            lfdm_buf_store_f4_sc1(slab_buf, (uint32_t)((((int64_t)my_slab * M + orow) * p.coutp + co) * 4), y[q]);
            yown[q] = y[q];
          } else {
            *reinterpret_cast<float4*>(p.partial + ((int64_t)bz * M + orow) * p.coutp + co) = y[q];
          }
        } else if (co < p.cout) {
          float4 v = make_float4(y[q].x + bb.x, y[q].y + bb.y, y[q].z + bb.z, y[q].w + bb.w);
          gs[ct][0] += v.x; gs[ct][1] += v.y; gs[ct][2] += v.z; gs[ct][3] += v.w;
          gq[ct][0] += v.x * v.x; gq[ct][1] += v.y * v.y; gq[ct][2] += v.z * v.z; gq[ct][3] += v.w * v.w;
          v.x += res[q].x; v.y += res[q].y; v.z += res[q].z; v.w += res[q].w;
          if (ACT) {
            v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act);
            v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
          }
          *reinterpret_cast<float4*>(p.out + orow * p.ldo + co) = v;
        }
      }
    }
  }
#ifdef LFDM_WINO_TIMING
  tstamp[4] = __builtin_readcyclecounter();
  tstamp[5] = wall_clock64() - wall0;
  tstamp[6] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);   // HW_ID | XCC_ID
  tstamp[7] = wall0;                                                   // (100 MHz wall clock at entry: which workgroups started together)
  if (threadIdx.x == 0 && p.tile_counters) {
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.tile_counters + 65536) +      // (behind the 64 K ticket words)
                              ((int64_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;      // (hardware id order)
    for (int i = 0; i < 8; ++i) dst[i] = tstamp[i];
  }
#endif
  bool reduced_here = false;
#if defined(LFDM_PROBE_NOTAIL) && LFDM_PROBE_NOTAIL == 1
  if (FUSE && nslab > 1) return;      // probe build only (tools/probe_wino_tail.sh; results WRONG): the launch without drain, ticket and reduce
#endif
  if (FUSE && nslab > 1) {
    __shared__ int s_last;
    LFDM_DRAIN_STORES();                                   // every storing wave: its slab words have left for memory
    __syncthreads();
    if (tid == 0) {
      unsigned* cnt = p.tile_counters + ((int64_t)by * gx + bx);
      const bool last = lfdm_ticket_take(cnt) == (unsigned)(nslab - 1);
      if (last) lfdm_ticket_reset(cnt);                   // ready for the next launch
      s_last = last ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
#if defined(LFDM_PROBE_NOTAIL) && LFDM_PROBE_NOTAIL == 2
    return;                            // probe build only: drain + ticket, but nobody reads the slabs back
#endif
    reduced_here = true;
    const int co = n0 + 4 * e_c4;
    const int n = my_n;
    const bool live = n >= 0 && co < p.cout;
#pragma unroll
    for (int e = 0; e < 4; ++e) gs[0][e] = gq[0][e] = 0.f;
    if (live) {
      const int64_t orow0 = ((int64_t)n * p.hq + 2 * my_ty) * p.wq + 2 * my_tx;
      float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) bb = *reinterpret_cast<const float4*>(p.bias + co);
      float4 res[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        res[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.residual) res[q] = *reinterpret_cast<const float4*>(p.residual + (orow0 + (q >> 1) * p.wq + (q & 1)) * p.ldr + co);
      }
      const uint32_t zs = (uint32_t)(M * p.coutp * 4);      // slab stride in bytes
      uint32_t src[4];
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) src[q] = (uint32_t)(((orow0 + (q >> 1) * p.wq + (q & 1)) * p.coutp + co) * 4);
      // four slabs x the thread's four pixels in flight per round (16 loads of 16 bytes, read past the L1: one memory round trip for
      // ksplit <= 4, two up to 8); summed z = 0, 1, ... per pixel (fixed order)
      for (int z0 = 0; z0 < nslab; z0 += 4) {
        float4 w[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            w[u][q] = lfdm_buf_load_f4_sc1(slab_buf, (z0 + u < nslab && z0 + u != my_slab) ? src[q] + (uint32_t)(z0 + u) * zs : LFDM_BUF_OOB);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (z0 + u < nslab) {
            const bool mine = z0 + u == my_slab;          // (the same bits this workgroup stored: the sum is the one every other arrival order gives)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 t = mine ? yown[q] : w[u][q];
              if (z0 + u == 0) v[q] = t;
              else { v[q].x += t.x; v[q].y += t.y; v[q].z += t.z; v[q].w += t.w; }
            }
          }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t orow = orow0 + (q >> 1) * p.wq + (q & 1);
        float4 t = v[q];
        t.x += bb.x; t.y += bb.y; t.z += bb.z; t.w += bb.w;
        gs[0][0] += t.x; gs[0][1] += t.y; gs[0][2] += t.z; gs[0][3] += t.w;
        gq[0][0] += t.x * t.x; gq[0][1] += t.y * t.y; gq[0][2] += t.z * t.z; gq[0][3] += t.w * t.w;
        t.x = apply_act(t.x + res[q].x, p.act); t.y = apply_act(t.y + res[q].y, p.act);
        t.z = apply_act(t.z + res[q].z, p.act); t.w = apply_act(t.w + res[q].w, p.act);
        *reinterpret_cast<float4*>(p.out + orow * p.ldo + co) = t;
      }
    }
  }
  if (p.gn_partial && (nslab == 1 || reduced_here)) {
    // per-channel sums over the workgroup's 32 tiles: lanes with equal e_c4 (stride 8) inside the wave, then the four waves
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float sv = gs[ct][e], qv = gq[ct][e];
#pragma unroll
        for (int msk = 8; msk <= 32; msk <<= 1) {
          sv += __shfl_xor(sv, msk);
          qv += __shfl_xor(qv, msk);
        }
        if (lane < 8) {
          s_gn[0][wave][WN * ct + 4 * e_c4 + e] = sv;
          s_gn[1][wave][WN * ct + 4 * e_c4 + e] = qv;
        }
      }
    __syncthreads();
    const int cg = p.cout / p.gn_groups;
    if (cg > WNB) {
      // a group wider than the workgroup's columns (512 channels / 8 groups at the 4x4 level): the workgroup's sums are ONE of the
      // cg / WNB column parts of its group and go to their own chunk slot - chunk = tile block * parts + part; the consumer merges
      // (pixels / tile rows) * parts chunks per sample (host: lfdm_conv2d_plan's tile_rows, unet.py)
      if (tid == 0) {
        float sv = 0.f, qv = 0.f;
        for (int c = 0; c < WNB; ++c)
          for (int w4 = 0; w4 < 4; ++w4) {
            sv += s_gn[0][w4][c];
            qv += s_gn[1][w4][c];
          }
        const int parts = cg / WNB;
        float* dst = p.gn_partial + (((int64_t)bx * parts + (n0 % cg) / WNB) * p.gn_groups + n0 / cg) * 2;
        dst[0] = sv;
        dst[1] = qv;
      }
      return;
    }
    const int gpt = WNB / cg;                     // groups inside this workgroup's columns (cg divides 32 or is a multiple of it: host check)
    if (tid < gpt && n0 + tid * cg < p.cout) {
      float sv = 0.f, qv = 0.f;
      for (int c = 0; c < cg; ++c)
        for (int w4 = 0; w4 < 4; ++w4) {
          sv += s_gn[0][w4][tid * cg + c];
          qv += s_gn[1][w4][tid * cg + c];
        }
      float* dst = p.gn_partial + ((int64_t)bx * p.gn_groups + (n0 / cg + tid)) * 2;
      dst[0] = sv;
      dst[1] = qv;
    }
  }
}

// U = G g G^T.  One thread per (16-channel chunk, half, output channel n, kh): the FOUR reduction channels k = 16 chunk + 8 kh + 4 half + e
// whose 16 transformed values lie side by side in the packed layout [pos][chunk][half][n][kh][e] - one 16-byte store per position, and
// adjacent threads (kh, then n) store adjacent 16 bytes: a wavefront writes 1 KB contiguous per position.  (One thread per (k, n) pair wrote
// 4 bytes at a 32-byte stride, 16 times: the multi-filter re-pack of a training step ran at 0.58 TB/s, profiles/r05_n_lfae_census.txt.)
__device__ __forceinline__ void pack_wino_item(const float* __restrict__ w, int ld_o, int cout, int cin, int coutp, int dgrad,
                                               float* __restrict__ out, int64_t idx) {
  const int K = dgrad ? cout : cin;                     // reduction channels of the target convolution
  const int N = dgrad ? cin : cout;                     // its output channels
  if (idx >= (int64_t)(K / 4) * coutp) return;
  const int kh2 = (int)(idx & 1);
  const int n = (int)((idx >> 1) % coutp);
  const int rest = (int)((idx >> 1) / coutp);           // chunk * 2 + half
  const int half = rest & 1, chunk = rest >> 1;
  const int k0 = chunk * WKC + 8 * kh2 + 4 * half;
  float u[16][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float g[9];
    if (n < N) {
      const int k = k0 + e;
      const float* src = dgrad ? w + (int64_t)k * ld_o + (int64_t)n * 9 : w + (int64_t)n * ld_o + (int64_t)k * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) g[t] = dgrad ? src[8 - t] : src[t];
    } else {
#pragma unroll
      for (int t = 0; t < 9; ++t) g[t] = 0.f;
    }
    float r[4][3];                                      // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      r[0][b] = g[b];
      r[1][b] = 0.5f * (g[b] + g[3 + b] + g[6 + b]);
      r[2][b] = 0.5f * (g[b] - g[3 + b] + g[6 + b]);
      r[3][b] = g[6 + b];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u[4 * i + 0][e] = r[i][0];
      u[4 * i + 1][e] = 0.5f * (r[i][0] + r[i][1] + r[i][2]);
      u[4 * i + 2][e] = 0.5f * (r[i][0] - r[i][1] + r[i][2]);
      u[4 * i + 3][e] = r[i][2];
    }
  }
  const int nch = K / WKC;
#pragma unroll
  for (int pos = 0; pos < 16; ++pos)
    *reinterpret_cast<float4*>(out + ((((((int64_t)pos) * nch + chunk) * 2 + half) * coutp + n) * 2 + kh2) * 4) =
        make_float4(u[pos][0], u[pos][1], u[pos][2], u[pos][3]);
}

__global__ __launch_bounds__(256) void pack_wino_kernel(const float* __restrict__ w, int ld_o, int cout, int cin, int coutp,
                                                        int dgrad, float* __restrict__ out) {
  pack_wino_item(w, ld_o, cout, cin, coutp, dgrad, out, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// Many filters in ONE launch (training re-packs every 3x3 filter - forward and data-gradient form - after each optimizer step: ~100 launches of
// 5-130 us per DM step, ~80 per LFAE step).  jobs: n_jobs records in device memory, sorted by block0 (the first workgroup of the job).
__global__ __launch_bounds__(256) void pack_wino_multi_kernel(const lfdm_pack_wino_job* __restrict__ jobs, int n_jobs) {
  __shared__ int s_job;
  if (threadIdx.x == 0) {
    int lo = 0, hi = n_jobs - 1;                        // last job with block0 <= blockIdx.x
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid;
      else hi = mid - 1;
    }
    s_job = lo;
  }
  __syncthreads();
  const lfdm_pack_wino_job j = jobs[s_job];
  pack_wino_item(j.w, j.ld_o, j.cout, j.cin, j.coutp, j.dgrad, j.out, (int64_t)((int)blockIdx.x - j.block0) * 256 + threadIdx.x);
}

}  // namespace

extern "C" int lfdm_pack_wino_weight_f32(const float* w, int ld_o, int cout, int cin, int coutp, int dgrad, float* out,
                                         lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int K = dgrad ? cout : cin, N = dgrad ? cin : cout;
  if (!w || !out || cout <= 0 || cin <= 0 || K % WKC != 0 || coutp < N || coutp % 32 != 0 || ld_o < cin * 9) {
    lfdm_set_error("pack_wino_weight: reduction channels must be a multiple of 16 and coutp a multiple of 32 >= the output channels");
    return LFDM_EINVAL;
  }
  const int64_t total = (int64_t)(K / 4) * coutp;
  LFDM_LAUNCH(pack_wino_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, ld_o, cout, cin, coutp, dgrad, out);
  return lfdm_check_launch("pack_wino_weight");
}

extern "C" int lfdm_pack_wino_weights_multi_f32(const lfdm_pack_wino_job* jobs, int n_jobs, int total_blocks, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!jobs || n_jobs <= 0 || total_blocks <= 0) { lfdm_set_error("pack_wino_weights_multi: needs a device job table and its workgroup count"); return LFDM_EINVAL; }
  LFDM_LAUNCH(pack_wino_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream, jobs, n_jobs);
  return lfdm_check_launch("pack_wino_weights_multi");
}

// grid (tile blocks, column tiles, ksplit).  Called by lfdm_conv2d_cl_f32 (conv_igemm.hip).
int lfdm_conv_wino_launch(const lfdm_conv_params& p, int bn, bool fuse_reduce, int bal_whole, hipStream_t stream) {
  const int64_t ntiles = (int64_t)p.n_img * (p.hq / 2) * (p.wq / 2);
  const dim3 grid((unsigned)((ntiles + WT - 1) / WT), (unsigned)((p.coutp + bn - 1) / bn), p.ksplit > 1 ? p.ksplit : 1);
  const bool act = p.act != LFDM_ACT_NONE;
  if (fuse_reduce) {                      // (splitk_fused / wino_balance, conv_igemm.hip: 32-column tiles; any activation at run time)
    if (bal_whole > 0) {                  // balanced: whole jobs first, then the two halves of each remaining job (see the kernel)
      const unsigned jobs = grid.x * grid.y * grid.z;
      LFDM_LAUNCH((conv_wino_kernel<false, 1, false, true>), dim3((unsigned)bal_whole + 2u * (jobs - (unsigned)bal_whole)), dim3(256), 0, stream, p, bal_whole);
    } else {
      LFDM_LAUNCH((conv_wino_kernel<false, 1, false, true>), grid, dim3(256), 0, stream, p, 0);
    }
    return lfdm_check_launch("conv_wino");
  }
  if (p.pool2 && act) {                   // (the pooled form follows an output activation in every caller: lfdm_conv2d_cl_f32 checks)
    if (bn == 64) LFDM_LAUNCH((conv_wino_kernel<true, 2, true>), grid, dim3(256), 0, stream, p, 0);
    else LFDM_LAUNCH((conv_wino_kernel<true, 1, true>), grid, dim3(256), 0, stream, p, 0);
  } else if (bn == 64 && act) LFDM_LAUNCH((conv_wino_kernel<true, 2>), grid, dim3(256), 0, stream, p, 0);
  else if (bn == 64) LFDM_LAUNCH((conv_wino_kernel<false, 2>), grid, dim3(256), 0, stream, p, 0);
  else if (act) LFDM_LAUNCH((conv_wino_kernel<true, 1>), grid, dim3(256), 0, stream, p, 0);
  else LFDM_LAUNCH((conv_wino_kernel<false, 1>), grid, dim3(256), 0, stream, p, 0);
  return lfdm_check_launch("conv_wino");
}
