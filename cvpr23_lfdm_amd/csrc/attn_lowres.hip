// PreNorm LayerNorm + to_qkv + attention core in ONE launch for the low-resolution UNet levels (C = 128 ... 512, a few hundred
// to a few thousand rows): include/lfdm_hip.h lfdm_linear_attention_lowres_cl_f32 (SpatialLinearAttention,
// DM/modules/video_flow_diffusion.py:240-265) and lfdm_attention_lowres_cl_f32 (Attention over the frame axis incl. rotary and
// the relative-position bias, or over the pixels of a frame in the mid block, :286-363), both without their to_out projection.
//
// Why: at these levels the separate launches (LN-folded to_qkv GEMM [+ split-K reduce], attention core [x2 for the linear
// form]) cost 25-50 us for 0.3-1 GFLOP - each sits at its launch + cold-input latency, and the 768-wide qkv rows make a
// round trip through memory.  Here one workgroup owns one (frame, head) resp. (pixel sequence, head):
//  * projection: the unit's rows of x (16 ... 256 of them) times the head's 96 rows of W' = W * gamma, on
//    v_mfma_f32_16x16x4_f32 with operand fragments straight from global memory (lane = (row | output column, k-slot), 16
//    contiguous bytes of a row per load, the contraction order is free).  The four wavefronts split the channels (KSPLIT) and
//    sum their partial tiles through LDS, or - 256-row frames - split the rows;
//  * the channel LayerNorm is folded algebraically (y = rstd * (x.W' - mean * sum_c W')): the row sums come from the same
//    fragments, so x is read exactly once;
//  * q | k | v of the unit stay in LDS ([row][96]); the linear form (k-softmax over the pixels, 32x32 context, q-softmax, output)
//    resp. the softmax attention (scores transposed in registers as in attention.hip) run on them and only the 32 output
//    columns of the head are written.
#include <stdlib.h>

#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

constexpr int HEADS = 8;
constexpr int DH = 32;
constexpr int OUT_LD = HEADS * DH;      // 256
constexpr int NQ = 3 * DH;              // q | k | v columns of one head
constexpr int LDQ = NQ + 4;             // LDS row stride (16-byte aligned rows)
constexpr float ATT_SCALE = 0.17677669529663687f;   // 32^-0.5

// One wavefront's share of the projection: rows r = 0 .. 16*NT-1 of the unit (global row row0 + r * rstride, rows >= nrows are
// clamped duplicates that nobody reads), channels [c_begin, c_end) in steps of 16.  acc[ti][nj]: D layout of v_mfma_f32_16x16x4
// (column 16*nj + l15 of the head's 96, row 16*ti + 4*ks + r).  s1 / s2: this lane's partial row sums (row 16*ti + l15).
template <int NT, int D>
__device__ __forceinline__ void project_wave_d(const float* __restrict__ x, int ldx, int64_t row0, int64_t rstride, int nrows,
                                               int c_begin, int c_end, const float* __restrict__ wf, int C, int head,
                                               f32x4 (&acc)[NT][6], float (&s1)[NT], float (&s2)[NT]) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, ks = lane >> 4;
  const float* arow[NT];
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    int r = 16 * ti + l15;
    if (r >= nrows) r = nrows - 1;
    arow[ti] = x + (row0 + (int64_t)r * rstride) * ldx + 4 * ks;
    s1[ti] = s2[ti] = 0.f;
  }
  const float* brow[6];
#pragma unroll
  for (int nj = 0; nj < 6; ++nj) {
    const int gcol = (nj >> 1) * OUT_LD + head * DH + (nj & 1) * 16 + l15;
    brow[nj] = wf + (int64_t)gcol * C + 4 * ks;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) acc[ti][nj] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // D channel groups (16 channels each) in flight per wavefront: the fragments come from L2 / HBM with ~1.5-2 us of latency under
  // load while a group's MFMAs take 0.2-1.5 us - one group ahead left the matrix pipe waiting (first version: 18.6 us for the 4x4
  // level's linear attention, 3 us of it MFMA).  The ring is indexed statically (the stage loop is unrolled); the group count
  // is a multiple of D (the caller picks D so), fetches past the end are clamped re-reads.
  float4 ra[D][NT], rb[D][6];
  const int last = c_end - 16;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const int c = c_begin + 16 * d < last ? c_begin + 16 * d : last;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) ra[d][ti] = *reinterpret_cast<const float4*>(arow[ti] + c);
#pragma unroll
    for (int nj = 0; nj < 6; ++nj) rb[d][nj] = *reinterpret_cast<const float4*>(brow[nj] + c);
  }
  for (int c0 = c_begin; c0 < c_end; c0 += 16 * D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      float4 a[NT], b[6];
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) a[ti] = ra[d][ti];
#pragma unroll
      for (int nj = 0; nj < 6; ++nj) b[nj] = rb[d][nj];
      const int cn = c0 + 16 * (d + D) < last ? c0 + 16 * (d + D) : last;
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) ra[d][ti] = *reinterpret_cast<const float4*>(arow[ti] + cn);
#pragma unroll
      for (int nj = 0; nj < 6; ++nj) rb[d][nj] = *reinterpret_cast<const float4*>(brow[nj] + cn);
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) {
        s1[ti] += (a[ti].x + a[ti].y) + (a[ti].z + a[ti].w);
        s2[ti] += (a[ti].x * a[ti].x + a[ti].y * a[ti].y) + (a[ti].z * a[ti].z + a[ti].w * a[ti].w);
      }
#pragma unroll
      for (int nj = 0; nj < 6; ++nj)
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
          acc[ti][nj] = mfma_16x16x4(a[ti].x, b[nj].x, acc[ti][nj]);
          acc[ti][nj] = mfma_16x16x4(a[ti].y, b[nj].y, acc[ti][nj]);
          acc[ti][nj] = mfma_16x16x4(a[ti].z, b[nj].z, acc[ti][nj]);
          acc[ti][nj] = mfma_16x16x4(a[ti].w, b[nj].w, acc[ti][nj]);
        }
    }
  }
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {                // the four k-slots of a row
    s1[ti] += __shfl_xor(s1[ti], 16); s2[ti] += __shfl_xor(s2[ti], 16);
    s1[ti] += __shfl_xor(s1[ti], 32); s2[ti] += __shfl_xor(s2[ti], 32);
  }
}

// The same share with a compile-time number G of channel groups, fully unrolled, D groups in flight: loads by inline asm, retired by
// counted s_waitcnt (lfdm_device.h).  Loads return in order, so when group g is consumed the groups issued after it - at most D-1 -
// may still be outstanding: vmcnt((NT+6) * min(D-1, G-1-g)).
template <int NT, int G, int D>
__device__ __forceinline__ void project_wave_asm(const float* __restrict__ x, int ldx, int64_t row0, int64_t rstride, int nrows,
                                                 int c_begin, const float* __restrict__ wf, int C, int head,
                                                 f32x4 (&acc)[NT][6], float (&s1)[NT], float (&s2)[NT]) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, ks = lane >> 4;
  const float* arow[NT];
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    int r = 16 * ti + l15;
    if (r >= nrows) r = nrows - 1;
    arow[ti] = x + (row0 + (int64_t)r * rstride) * ldx + c_begin + 4 * ks;
    s1[ti] = s2[ti] = 0.f;
  }
  const float* brow[6];
#pragma unroll
  for (int nj = 0; nj < 6; ++nj) {
    const int gcol = (nj >> 1) * OUT_LD + head * DH + (nj & 1) * 16 + l15;
    brow[nj] = wf + (int64_t)gcol * C + c_begin + 4 * ks;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) acc[ti][nj] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  f32x4 ra[D][NT], rb[D][6];
  auto issue = [&](auto gc) {
    constexpr int g = decltype(gc)::value, st = g % D;
    lfdm_static_for<0, NT>([&](auto ti) { lfdm_gload_f4<64 * g>(ra[st][decltype(ti)::value], arow[decltype(ti)::value]); });
    lfdm_static_for<0, 6>([&](auto nj) { lfdm_gload_f4<64 * g>(rb[st][decltype(nj)::value], brow[decltype(nj)::value]); });
  };
  lfdm_static_for<0, (D < G ? D : G)>(issue);
  lfdm_static_for<0, G>([&](auto gc) {
    constexpr int g = decltype(gc)::value, st = g % D;
    constexpr int younger = (G - 1 - g) < (D - 1) ? (G - 1 - g) : (D - 1);
    lfdm_vmwait<younger * (NT + 6)>();
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) lfdm_tie(ra[st][ti]);
#pragma unroll
    for (int nj = 0; nj < 6; ++nj) lfdm_tie(rb[st][nj]);
    f32x4 a[NT], b[6];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) a[ti] = ra[st][ti];
#pragma unroll
    for (int nj = 0; nj < 6; ++nj) b[nj] = rb[st][nj];
    // refill of this stage: before the group's MFMAs where the registers allow a (D+1)-th stage to be live, after them otherwise
    // (NT >= 3: 96 accumulator registers; the group's own MFMAs are then 1-1.5 us of cover for the next group)
    if constexpr (NT <= 2 && g + D < G) issue(std::integral_constant<int, g + D>{});
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
      s1[ti] += (a[ti][0] + a[ti][1]) + (a[ti][2] + a[ti][3]);
      s2[ti] += (a[ti][0] * a[ti][0] + a[ti][1] * a[ti][1]) + (a[ti][2] * a[ti][2] + a[ti][3] * a[ti][3]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int nj = 0; nj < 6; ++nj)
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) acc[ti][nj] = mfma_16x16x4(a[ti][e], b[nj][e], acc[ti][nj]);
    if constexpr (NT > 2 && g + D < G) issue(std::integral_constant<int, g + D>{});
  });
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    s1[ti] += __shfl_xor(s1[ti], 16); s2[ti] += __shfl_xor(s2[ti], 16);
    s1[ti] += __shfl_xor(s1[ti], 32); s2[ti] += __shfl_xor(s2[ti], 32);
  }
}

// 2 / 4 / 8 channel groups per wavefront (C = 128 / 256 / 512 split four ways, or C = 128 unsplit): the unrolled asm pipeline;
// any other count: the compiler-scheduled loop
template <int NT, bool ASM = true>
__device__ __forceinline__ void project_wave(const float* __restrict__ x, int ldx, int64_t row0, int64_t rstride, int nrows,
                                             int c_begin, int c_end, const float* __restrict__ wf, int C, int head,
                                             f32x4 (&acc)[NT][6], float (&s1)[NT], float (&s2)[NT]) {
  const int groups = ASM ? (c_end - c_begin) >> 4 : 0;
  // (NT + 6) float4 per stage in flight.  The depth is bounded by the register file, and that bound is HARD: with more live values
  // than registers hipcc spills the destination of an asm load that has not landed yet, or re-uses it (tools/check_asm_pipeline.py
  // finds both in the assembly; tests/test_host_and_abi.py runs it on every file that uses lfdm_gload_f4)
  constexpr int D = NT == 1 ? 4 : 2;
  if (groups == 8) project_wave_asm<NT, 8, D>(x, ldx, row0, rstride, nrows, c_begin, wf, C, head, acc, s1, s2);
  else if (groups == 4) project_wave_asm<NT, 4, D>(x, ldx, row0, rstride, nrows, c_begin, wf, C, head, acc, s1, s2);
  else if (groups == 2) project_wave_asm<NT, 2, 2>(x, ldx, row0, rstride, nrows, c_begin, wf, C, head, acc, s1, s2);
  else project_wave_d<NT, 1>(x, ldx, row0, rstride, nrows, c_begin, c_end, wf, C, head, acc, s1, s2);
}

// K-split projection of 16*NT rows by the whole workgroup (4 wavefronts): result (LayerNorm folded) in part[0..16*NT)[LDQ].
// part: 4 * 16*NT * LDQ floats; pstat: 2 * 4 * 16*NT floats; wsl: the head's 96 column sums of W'.
template <int NT, bool ASM = true>
__device__ __forceinline__ void project_ksplit(const float* __restrict__ x, int ldx, int64_t row0, int64_t rstride, int nrows,
                                               const float* __restrict__ wf, int C, int head, float eps, float* part,
                                               float* pstat, float* wsl, float wsum_val) {
  constexpr int R = 16 * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, ks = lane >> 4;
  f32x4 acc[NT][6];
  float s1[NT], s2[NT];
  const int cw = C / 4;
  project_wave<NT, ASM>(x, ldx, row0, rstride, nrows, wave * cw, (wave + 1) * cw, wf, C, head, acc, s1, s2);
  float* mine = part + wave * (R * LDQ);
  if (tid < NQ) wsl[tid] = wsum_val;
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    if (ks == 0) {
      pstat[(0 * 4 + wave) * R + 16 * ti + l15] = s1[ti];
      pstat[(1 * 4 + wave) * R + 16 * ti + l15] = s2[ti];
    }
#pragma unroll
    for (int nj = 0; nj < 6; ++nj)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(16 * ti + 4 * ks + r) * LDQ + 16 * nj + l15] = acc[ti][nj][r];
  }
  __syncthreads();
  // every thread finishes whole float4 items of the tile: sum of the four K slices, LayerNorm fold, written over slice 0
  const float inv_c = 1.0f / (float)C;
  for (int it = tid; it < R * (NQ / 4); it += 256) {
    const int row = it / (NQ / 4), q4 = it - row * (NQ / 4);
    float sv = 0.f, qv = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      sv += pstat[(0 * 4 + w) * R + row];
      qv += pstat[(1 * 4 + w) * R + row];
    }
    const float mean = sv * inv_c;
    float var = qv * inv_c - mean * mean;
    if (var < 0.f) var = 0.f;
    const float rstd = 1.0f / sqrtf(var + eps);
    float4 v = *reinterpret_cast<const float4*>(part + row * LDQ + 4 * q4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 u = *reinterpret_cast<const float4*>(part + w * (R * LDQ) + row * LDQ + 4 * q4);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const float4 ws = *reinterpret_cast<const float4*>(wsl + 4 * q4);
    v.x = rstd * (v.x - mean * ws.x); v.y = rstd * (v.y - mean * ws.y);
    v.z = rstd * (v.z - mean * ws.z); v.w = rstd * (v.w - mean * ws.w);
    *reinterpret_cast<float4*>(part + row * LDQ + 4 * q4) = v;
  }
  __syncthreads();
}

// the head's 96 column sums of W': requested at kernel entry, parked in a register while the projection runs, written to LDS in front of
// the projection's first barrier - a load -> LDS -> barrier sequence at the top put a global round trip in front of every
// workgroup's first fragment load
__device__ __forceinline__ float load_wsum(const float* __restrict__ wsum, int head) {
  const int tid = threadIdx.x;
  return tid < NQ ? wsum[(tid >> 5) * OUT_LD + head * DH + (tid & 31)] : 0.f;
}

// ------------------------------------------------------------------------------------------------------------------
// Linear attention: grid (n_frames * 8), 256 threads.  MSPLIT: hw = 256 (each wavefront projects its own 64 rows over all
// channels, no partial tiles); otherwise the rows are walked in groups of 16*NT with the channels split over the wavefronts.
template <int NT, bool MSPLIT>
__global__ __launch_bounds__(256) void linattn_lowres_kernel(const float* __restrict__ x, int ldx, int C,
                                                             const float* __restrict__ wf, const float* __restrict__ wsum,
                                                             float* __restrict__ out, int hw, float eps) {
  constexpr int R = 16 * NT;
  constexpr int ROWS = MSPLIT ? 4 * R : R;                       // rows held in LDS
  __shared__ __attribute__((aligned(16))) float qkv[(MSPLIT ? ROWS : 4 * R) * LDQ];   // K-split: the four partial tiles, result in tile 0
  __shared__ float pstat[2 * 4 * R];
  __shared__ __attribute__((aligned(16))) float wsl[NQ];
  __shared__ float red[8][DH], kmax[DH], ksum[DH];
  __shared__ __attribute__((aligned(16))) float ctx[DH][DH + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, ks = lane >> 4;
  const int f = blockIdx.x >> 3, head = blockIdx.x & 7;
  const int64_t row0 = (int64_t)f * hw;
  const float wsum_val = load_wsum(wsum, head);

  if (MSPLIT) {
    f32x4 acc[NT][6];
    float s1[NT], s2[NT];
    const int nrows_w = hw - wave * R;              // rows left for this wavefront (>= 1: host check hw > 3*R)
    project_wave<NT>(x, ldx, row0 + wave * R, 1, nrows_w < R ? nrows_w : R, 0, C, wf, C, head, acc, s1, s2);
    if (tid < NQ) wsl[tid] = wsum_val;
    if (ks == 0) {
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) {
        const float mean = s1[ti] / (float)C;
        float var = s2[ti] / (float)C - mean * mean;
        if (var < 0.f) var = 0.f;
        pstat[wave * R + 16 * ti + l15] = mean;
        pstat[4 * R + wave * R + 16 * ti + l15] = 1.0f / sqrtf(var + eps);
      }
    }
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = 16 * ti + 4 * ks + r;
        const float mean = pstat[wave * R + rl], rstd = pstat[4 * R + wave * R + rl];
#pragma unroll
        for (int nj = 0; nj < 6; ++nj)
          qkv[(wave * R + rl) * LDQ + 16 * nj + l15] = rstd * (acc[ti][nj][r] - mean * wsl[16 * nj + l15]);
      }
    __syncthreads();
  } else {
    project_ksplit<NT>(x, ldx, row0, 1, hw, wf, C, head, eps, qkv, pstat, wsl, wsum_val);
  }

  // ---- k softmax over the pixels (per feature d), in place: k <- exp(k - max_n k); ksum[d] ----
  {
    const int d = tid & 31, part = tid >> 5;          // 8 row parts x 32 features
    float m = -3.0e38f;
    for (int n = part; n < hw; n += 8) m = fmaxf(m, qkv[n * LDQ + DH + d]);
    red[part][d] = m;
    __syncthreads();
    if (tid < DH) {
      float mm = red[0][tid];
#pragma unroll
      for (int p = 1; p < 8; ++p) mm = fmaxf(mm, red[p][tid]);
      kmax[tid] = mm;
    }
    __syncthreads();
    const float mx = kmax[d];
    float s = 0.f;
    for (int n = part; n < hw; n += 8) {
      const float e = expf(qkv[n * LDQ + DH + d] - mx);
      qkv[n * LDQ + DH + d] = e;
      s += e;
    }
    red[part][d] = s;
    __syncthreads();
    if (tid < DH) {
      float ss = 0.f;
#pragma unroll
      for (int p = 0; p < 8; ++p) ss += red[p][tid];
      ksum[tid] = ss;
    }
    __syncthreads();
  }
  // ---- context[d][e] = sum_n k~[n][d] v[n][e] / ksum[d]: thread = (d, four e) ----
  {
    const int d = tid >> 3, e0 = (tid & 7) * 4;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int n = 0; n < hw; ++n) {
      const float e = qkv[n * LDQ + DH + d];
      const float4 v4 = *reinterpret_cast<const float4*>(qkv + n * LDQ + 2 * DH + e0);
      a0 = fmaf(e, v4.x, a0); a1 = fmaf(e, v4.y, a1); a2 = fmaf(e, v4.z, a2); a3 = fmaf(e, v4.w, a3);
    }
    const float ss = ksum[d];
    *reinterpret_cast<float4*>(&ctx[d][e0]) = make_float4(a0 / ss, a1 / ss, a2 / ss, a3 / ss);
  }
  // ---- q softmax over the 32 features of a pixel (8 lanes per pixel, four features each), in place, scaled ----
  for (int n0 = 0; n0 < hw; n0 += 32) {             // (uniform trip count: the shuffles below involve every lane of a wavefront)
    const bool live = n0 + (tid >> 3) < hw;
    const int n = live ? n0 + (tid >> 3) : hw - 1;
    float4 q4 = *reinterpret_cast<const float4*>(qkv + n * LDQ + 4 * (tid & 7));
    float m = fmaxf(fmaxf(q4.x, q4.y), fmaxf(q4.z, q4.w));
    m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
    q4.x = expf(q4.x - m); q4.y = expf(q4.y - m); q4.z = expf(q4.z - m); q4.w = expf(q4.w - m);
    float s = (q4.x + q4.y) + (q4.z + q4.w);
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    q4.x = q4.x / s * ATT_SCALE; q4.y = q4.y / s * ATT_SCALE; q4.z = q4.z / s * ATT_SCALE; q4.w = q4.w / s * ATT_SCALE;
    if (live) *reinterpret_cast<float4*>(qkv + n * LDQ + 4 * (tid & 7)) = q4;
  }
  __syncthreads();
  // ---- out[n][e] = sum_d q~[n][d] context[d][e]: item = (pixel, four e) ----
  for (int it = tid; it < hw * 8; it += 256) {
    const int n = it >> 3, e0 = (it & 7) * 4;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int d4 = 0; d4 < DH / 4; ++d4) {
      const float4 q4 = *reinterpret_cast<const float4*>(qkv + n * LDQ + 4 * d4);
      const float qd[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 c4 = *reinterpret_cast<const float4*>(&ctx[4 * d4 + i][e0]);
        o.x = fmaf(c4.x, qd[i], o.x); o.y = fmaf(c4.y, qd[i], o.y); o.z = fmaf(c4.z, qd[i], o.z); o.w = fmaf(c4.w, qd[i], o.w);
      }
    }
    *reinterpret_cast<float4*>(out + (row0 + n) * OUT_LD + head * DH + e0) = o;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Softmax attention: grid (sequences * 8), 256 threads.  mode 0 (temporal): sequence = the `frames` rows of one pixel (row stride
// hw); mode 1 (spatial, the mid block): sequence = the hw rows of one frame.  16*NT >= tokens.  After the K-split projection
// wavefront ti runs query tile ti of the softmax attention from LDS.
template <int NT>
__global__ __launch_bounds__(256) void attn_lowres_kernel(const float* __restrict__ x, int ldx, int C,
                                                          const float* __restrict__ wf, const float* __restrict__ wsum,
                                                          float* __restrict__ out, int frames, int hw, int mode,
                                                          const float* __restrict__ bias, const float* __restrict__ rot_cos,
                                                          const float* __restrict__ rot_sin, float eps) {
  constexpr int R = 16 * NT;
  __shared__ __attribute__((aligned(16))) float qkv[4 * R * LDQ];
  __shared__ float pstat[2 * 4 * R];
  __shared__ __attribute__((aligned(16))) float wsl[NQ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
  const int64_t unit = blockIdx.x;
  const int64_t seq = unit >> 3;
  const int head = (int)(unit & 7);
  int64_t row0, tstride;
  if (mode == 0) {
    const int64_t b = seq / hw, pix = seq - b * hw;
    row0 = b * frames * hw + pix;
    tstride = hw;
  } else {
    row0 = seq * hw;
    tstride = 1;
  }
  const int L = mode == 0 ? frames : hw;
  const float wsum_val = load_wsum(wsum, head);
  // (49 ... 64 tokens: with 96 accumulator registers, the ring and the attention's own fragments the asm pipeline does not fit the
  //  register file safely - tools/check_asm_pipeline.py - and no LFDM configuration has more than 40 frames: compiler-scheduled loop)
  project_ksplit<NT, (NT <= 3)>(x, ldx, row0, tstride, L, wf, C, head, eps, qkv, pstat, wsl, wsum_val);

  for (int ti = wave; ti < NT; ti += 4) {           // one query tile per wavefront (wave-uniform)
    // Q fragment of this tile / K fragments of every tile: features 8*lq .. 8*lq+7 of token 16*t + l15 (scale, rotary in registers)
    float qf[8], kf[NT][8];
    {
      const int t = ti * 16 + l15;
      const float* src = qkv + t * LDQ + 8 * lq;
      const float4 q0 = *reinterpret_cast<const float4*>(src), q1 = *reinterpret_cast<const float4*>(src + 4);
      qf[0] = q0.x * ATT_SCALE; qf[1] = q0.y * ATT_SCALE; qf[2] = q0.z * ATT_SCALE; qf[3] = q0.w * ATT_SCALE;
      qf[4] = q1.x * ATT_SCALE; qf[5] = q1.y * ATT_SCALE; qf[6] = q1.z * ATT_SCALE; qf[7] = q1.w * ATT_SCALE;
      if (rot_cos && t < L) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          const float c = rot_cos[t * 16 + 4 * lq + pr], sn = rot_sin[t * 16 + 4 * lq + pr];
          const float qx = qf[2 * pr], qy = qf[2 * pr + 1];
          qf[2 * pr] = qx * c - qy * sn;
          qf[2 * pr + 1] = qy * c + qx * sn;
        }
      }
    }
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
      const int t = tj * 16 + l15;
      const float* src = qkv + t * LDQ + DH + 8 * lq;
      const float4 k0 = *reinterpret_cast<const float4*>(src), k1 = *reinterpret_cast<const float4*>(src + 4);
      kf[tj][0] = k0.x; kf[tj][1] = k0.y; kf[tj][2] = k0.z; kf[tj][3] = k0.w;
      kf[tj][4] = k1.x; kf[tj][5] = k1.y; kf[tj][6] = k1.z; kf[tj][7] = k1.w;
      if (rot_cos && t < L) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          const float c = rot_cos[t * 16 + 4 * lq + pr], sn = rot_sin[t * 16 + 4 * lq + pr];
          const float kx = kf[tj][2 * pr], ky = kf[tj][2 * pr + 1];
          kf[tj][2 * pr] = kx * c - ky * sn;
          kf[tj][2 * pr + 1] = ky * c + kx * sn;
        }
      }
    }
    // S^T = K Q^T: lane = query token 16*ti + l15, registers = key tokens 16*tj + 4*lq + r (attention.hip)
    f32x4 st[NT];
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) a = mfma_16x16x4(kf[tj][s], qf[s], a);
      st[tj] = a;
    }
    const int qt = ti * 16 + l15;
    float m = -3.0e38f;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = tj * 16 + lq * 4 + r;
        float v = st[tj][r];
        if (key >= L) v = -3.0e38f;
        else if (bias && qt < L) v += bias[((int64_t)head * L + qt) * L + key];
        st[tj][r] = v;
        m = fmaxf(m, v);
      }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = (tj * 16 + lq * 4 + r) < L ? expf(st[tj][r] - m) : 0.f;
        st[tj][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    // O = P V: B operand = v[token 16*tj + 4*lq + r][16*half + l15]
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = st[tj][r] / sum;
        const float* vr = qkv + (16 * tj + 4 * lq + r) * LDQ + 2 * DH;
        o0 = mfma_16x16x4(p, vr[l15], o0);
        o1 = mfma_16x16x4(p, vr[16 + l15], o1);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = ti * 16 + lq * 4 + r;
      if (t < L) {
        float* dst = out + (row0 + (int64_t)t * tstride) * OUT_LD + head * DH;
        dst[l15] = o0[r];
        dst[16 + l15] = o1[r];
      }
    }
  }
}

bool lowres_args_ok(const float* x, int ldx, int channels, const float* wqkv, const float* wsum, const float* out) {
  return x && wqkv && wsum && out && channels >= 64 && channels % 64 == 0 && ldx >= channels && ldx % 4 == 0 &&
         ((((uintptr_t)x) | ((uintptr_t)wqkv) | ((uintptr_t)out)) & 15) == 0;
}

}  // namespace

extern "C" int lfdm_linear_attention_lowres_cl_f32(const float* x, int ldx, int channels, const float* wqkv, const float* wsum,
                                                   float* out, int n_frames, int hw, float ln_eps, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!lowres_args_ok(x, ldx, channels, wqkv, wsum, out) || n_frames <= 0 || hw <= 0 || hw > 256 || (int64_t)n_frames * HEADS > 0x7fffffff) {
    lfdm_set_error("linear_attention_lowres: needs C % 64 == 0, 1 <= hw <= 256 pixels per frame, 16-byte aligned rows");
    return LFDM_EINVAL;
  }
  const dim3 grid((unsigned)(n_frames * HEADS)), block(256);
  if (hw <= 16) LFDM_LAUNCH((linattn_lowres_kernel<1, false>), grid, block, 0, stream, x, ldx, channels, wqkv, wsum, out, hw, ln_eps);
  else if (hw <= 32) LFDM_LAUNCH((linattn_lowres_kernel<2, false>), grid, block, 0, stream, x, ldx, channels, wqkv, wsum, out, hw, ln_eps);
  else if (hw <= 64) LFDM_LAUNCH((linattn_lowres_kernel<4, false>), grid, block, 0, stream, x, ldx, channels, wqkv, wsum, out, hw, ln_eps);
  else if (hw > 192) LFDM_LAUNCH((linattn_lowres_kernel<4, true>), grid, block, 0, stream, x, ldx, channels, wqkv, wsum, out, hw, ln_eps);
  else {
    lfdm_set_error("linear_attention_lowres: 64 < hw <= 192 pixels per frame is not built (use the unfused entry points)");
    return LFDM_EINVAL;
  }
  return lfdm_check_launch("linear_attention_lowres");
}

extern "C" int lfdm_attention_lowres_cl_f32(const float* x, int ldx, int channels, const float* wqkv, const float* wsum,
                                            float* out, int batch, int frames, int hw, int mode, const float* bias,
                                            const float* rot_cos, const float* rot_sin, float ln_eps, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int L = mode == 0 ? frames : hw;
  const int64_t nseq = mode == 0 ? (int64_t)batch * hw : (int64_t)batch * frames;
  if (!lowres_args_ok(x, ldx, channels, wqkv, wsum, out) || batch <= 0 || frames <= 0 || hw <= 0 || (mode != 0 && mode != 1) || L > 64 ||
      ((rot_cos == nullptr) != (rot_sin == nullptr)) || nseq * HEADS > 0x7fffffff) {
    lfdm_set_error("attention_lowres: needs C % 64 == 0, <= 64 tokens per sequence, 16-byte aligned rows");
    return LFDM_EINVAL;
  }
  const dim3 grid((unsigned)(nseq * HEADS)), block(256);
  if (L <= 16) LFDM_LAUNCH((attn_lowres_kernel<1>), grid, block, 0, stream, x, ldx, channels, wqkv, wsum, out, frames, hw, mode, bias, rot_cos, rot_sin, ln_eps);
  else if (L <= 32) LFDM_LAUNCH((attn_lowres_kernel<2>), grid, block, 0, stream, x, ldx, channels, wqkv, wsum, out, frames, hw, mode, bias, rot_cos, rot_sin, ln_eps);
  else if (L <= 48) LFDM_LAUNCH((attn_lowres_kernel<3>), grid, block, 0, stream, x, ldx, channels, wqkv, wsum, out, frames, hw, mode, bias, rot_cos, rot_sin, ln_eps);
  else LFDM_LAUNCH((attn_lowres_kernel<4>), grid, block, 0, stream, x, ldx, channels, wqkv, wsum, out, frames, hw, mode, bias, rot_cos, rot_sin, ln_eps);
  return lfdm_check_launch("attention_lowres");
}
