// Exploratory micro-benchmark (report-only - nothing in the library uses it): the fp32 contraction
//   C[m][n] = sum_k A[m][k] * W[n][k]        (the 1x1-convolution / to_out form: both operands K-contiguous)
// once on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32, what every kernel of this repo uses) and once as a 3-term bf16 split
//   x = hi + mid + lo  (each a bf16, round-to-nearest-even of the running remainder),
//   a*b ~= hi*hi + (hi*mid + mid*hi) + (hi*lo + lo*hi + mid*mid)        (6 of the 9 products, fp32 accumulate)
// on v_mfma_f32_32x32x16_bf16 (16x the fp32 rate per instruction -> 16/6 = 2.67x fewer matrix-pipe cycles).
// SAME tiling for both (128x128 workgroup tile, 4 waves x (64x64), 32-deep K chunks through LDS, next chunk's global loads issued before
// the current chunk's MFMAs), so the ratio is about the instruction mix, not about tuning.  Prints time, fp32-equivalent TFLOP/s
// (2*M*N*K) and the error of both against an fp64 host reference on sampled rows.
// Fragment convention: A and B fragments take the SAME (lane >> 5, element) -> k mapping, so the result does not depend on which k
// the hardware assigns to which element slot; rows / columns = lane & 31 (cdna_hip_programming.md, fragment layout).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bf16x3_gemm.bin tools/ubench/bf16x3_gemm.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDF = BK + 4;      // fp32 LDS row stride (floats): 16-byte aligned rows, conflict-free 2-float reads
constexpr int LDH = BK + 8;      // bf16 LDS row stride (halfs): 80-byte rows -> the 16-byte fragment reads of 32 rows spread over all banks

__device__ __forceinline__ uint16_t bf16_rne(float x) {
  uint32_t u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

__device__ __forceinline__ void split3(float x, uint16_t& h, uint16_t& m, uint16_t& l) {
  h = bf16_rne(x);
  const float r1 = x - bf16_f32(h);           // exact
  m = bf16_rne(r1);
  const float r2 = r1 - bf16_f32(m);          // exact
  l = bf16_rne(r2);
}

// ---------------------------------------------------------------- fp32 MFMA
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 3))) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C,
                                                          int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) float sA[BM * LDF], sB[BN * LDF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  f32x16 acc[2][2];
  #pragma unroll
  for (int i = 0; i < 2; ++i)
    #pragma unroll
    for (int j = 0; j < 2; ++j)
      #pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // staging: thread -> 4 float4 of A and 4 of B per chunk: row = (tid >> 3) + 32 * it, float4 column tid & 7
  const int sr = tid >> 3, sc = (tid & 7) * 4;
  f32x4 ra[4], rb[4];
  const float* ap = A + (size_t)(m0 + sr) * K + sc;
  const float* wp = W + (size_t)(n0 + sr) * K + sc;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    ra[it] = *reinterpret_cast<const f32x4*>(ap + (size_t)32 * it * K);
    rb[it] = *reinterpret_cast<const f32x4*>(wp + (size_t)32 * it * K);
  }
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
    #pragma unroll
    for (int it = 0; it < 4; ++it) {
      *reinterpret_cast<f32x4*>(sA + (sr + 32 * it) * LDF + sc) = ra[it];
      *reinterpret_cast<f32x4*>(sB + (sr + 32 * it) * LDF + sc) = rb[it];
    }
    __syncthreads();
    const int kn = k0 + BK < K ? k0 + BK : k0;    // (the last chunk re-requests itself: branch-free)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      ra[it] = *reinterpret_cast<const f32x4*>(ap + (size_t)32 * it * K + kn);
      rb[it] = *reinterpret_cast<const f32x4*>(wp + (size_t)32 * it * K + kn);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[2], b[2];
      #pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = sA[(wm + 32 * i + (lane & 31)) * LDF + kk + (lane >> 5)];
      #pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = sB[(wn + 32 * j + (lane & 31)) * LDF + kk + (lane >> 5)];
      #pragma unroll
      for (int i = 0; i < 2; ++i)
        #pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  #pragma unroll
  for (int i = 0; i < 2; ++i)
    #pragma unroll
    for (int j = 0; j < 2; ++j)
      #pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        C[(size_t)row * N + n0 + wn + 32 * j + (lane & 31)] = acc[i][j][r];
      }
}

// ---------------------------------------------------------------- 3-term bf16 split on the bf16 MFMA
// TERMS = 6: all products of order <= 2 (the fp32-grade variant); TERMS = 3: hi*hi + hi*mid + mid*hi (~16-bit mantissa, for reference)
template <int TERMS>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C,
                                                             int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) uint16_t sA[3][BM * LDH], sB[3][BN * LDH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  f32x16 acc[2][2];
  #pragma unroll
  for (int i = 0; i < 2; ++i)
    #pragma unroll
    for (int j = 0; j < 2; ++j)
      #pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int sr = tid >> 3, sc = (tid & 7) * 4;
  float4 ra[4], rb[4];
  auto gload = [&](int k0) {
    #pragma unroll
    for (int it = 0; it < 4; ++it) {
      ra[it] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + sr + 32 * it) * K + k0 + sc);
      rb[it] = *reinterpret_cast<const float4*>(W + (size_t)(n0 + sr + 32 * it) * K + k0 + sc);
    }
  };
  auto stage = [&](uint16_t (*dst)[BM * LDH], const float4& v, int row) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    uint16_t h[4], m[4], l[4];
    #pragma unroll
    for (int e = 0; e < 4; ++e) split3(x[e], h[e], m[e], l[e]);
    const int o = row * LDH + sc;                         // 8-byte aligned (LDH * 2 = 80 bytes per row, sc * 2 = multiple of 8)
    *reinterpret_cast<uint2*>(&dst[0][o]) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
    *reinterpret_cast<uint2*>(&dst[1][o]) = make_uint2(m[0] | ((uint32_t)m[1] << 16), m[2] | ((uint32_t)m[3] << 16));
    *reinterpret_cast<uint2*>(&dst[2][o]) = make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
  };
  gload(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
    #pragma unroll
    for (int it = 0; it < 4; ++it) {
      stage(sA, ra[it], sr + 32 * it);
      stage(sB, rb[it], sr + 32 * it);
    }
    __syncthreads();
    gload(k0 + BK < K ? k0 + BK : k0);          // (the last chunk re-requests itself: branch-free, the registers stay registers)
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      bf16x8 a[3][2], b[3][2];
      #pragma unroll
      for (int t = 0; t < 3; ++t) {
        #pragma unroll
        for (int i = 0; i < 2; ++i)
          a[t][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(&sA[t][(wm + 32 * i + (lane & 31)) * LDH + kk + 8 * (lane >> 5)]));
        #pragma unroll
        for (int j = 0; j < 2; ++j)
          b[t][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(&sB[t][(wn + 32 * j + (lane & 31)) * LDH + kk + 8 * (lane >> 5)]));
      }
      #pragma unroll
      for (int i = 0; i < 2; ++i)
        #pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 c = acc[i][j];
          if (TERMS == 6) {                               // smallest terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);
          }
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
  }
  #pragma unroll
  for (int i = 0; i < 2; ++i)
    #pragma unroll
    for (int j = 0; j < 2; ++j)
      #pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        C[(size_t)row * N + n0 + wn + 32 * j + (lane & 31)] = acc[i][j][r];
      }
}

// ---------------------------------------------------------------- host
static float frand(uint64_t& s) {      // approximately N(0, 1): sum of 4 uniforms, rescaled
  float a = 0.f;
  for (int i = 0; i < 4; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    a += (float)((s >> 40) & 0xFFFFFF) / 16777216.0f;
  }
  return (a - 2.0f) * 1.7320508f;
}

template <typename F>
static float time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return ms / reps;
}

static void errors(const std::vector<float>& hA, const std::vector<float>& hW, const float* dC, int M, int N, int K, const char* tag) {
  // fp64 reference on 64 sampled rows (all columns); error relative to the largest |C| of the sample
  std::vector<float> row(N);
  double max_abs = 0.0, max_ref = 0.0, sum_sq = 0.0, sum_ref_sq = 0.0;
  for (int s = 0; s < 64; ++s) {
    const int m = (int)(((int64_t)s * 2654435761ll) % M);
    hipMemcpy(row.data(), dC + (size_t)m * N, N * sizeof(float), hipMemcpyDeviceToHost);
    for (int n = 0; n < N; ++n) {
      double ref = 0.0;
      for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k];
      const double d = fabs((double)row[n] - ref);
      max_abs = d > max_abs ? d : max_abs;
      max_ref = fabs(ref) > max_ref ? fabs(ref) : max_ref;
      sum_sq += d * d;
      sum_ref_sq += ref * ref;
    }
  }
  printf("    %-18s max|err| / max|C| = %.3e   rms err / rms C = %.3e\n", tag, max_abs / max_ref, sqrt(sum_sq / sum_ref_sq));
}

static void run(int M, int N, int K, const char* what) {
  std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
  uint64_t seed = 0x1234567ull + M + 31 * N + 977 * K;
  for (auto& v : hA) v = frand(seed);
  for (auto& v : hW) v = frand(seed) / sqrtf((float)K);
  float *dA, *dW, *dC;
  hipMalloc(&dA, hA.size() * 4);
  hipMalloc(&dW, hW.size() * 4);
  hipMalloc(&dC, (size_t)M * N * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
  const dim3 grid(M / BM, N / BN), block(256);
  const double flop = 2.0 * M * N * K;
  const int reps = flop > 1e11 ? 5 : 20;
  printf("C[%d x %d] = A[%d x %d] . W[%d x %d]^T  (%s; %.1f GFLOP, %d workgroups)\n", M, N, M, K, N, K, what, flop / 1e9, grid.x * grid.y);
  float ms = time_ms([&] { hipLaunchKernelGGL(gemm_f32_kernel, grid, block, 0, 0, dA, dW, dC, M, N, K); }, reps);
  printf("  fp32 MFMA 32x32x2      : %8.3f ms  %7.1f TFLOP/s\n", ms, flop / ms / 1e9);
  errors(hA, hW, dC, M, N, K, "fp32 MFMA");
  const float ms32 = ms;
  ms = time_ms([&] { hipLaunchKernelGGL(gemm_bf16x3_kernel<6>, grid, block, 0, 0, dA, dW, dC, M, N, K); }, reps);
  printf("  bf16 x3, 6 products    : %8.3f ms  %7.1f TFLOP/s (fp32-equivalent)   %.2fx the fp32 kernel\n", ms, flop / ms / 1e9, ms32 / ms);
  errors(hA, hW, dC, M, N, K, "bf16x3 / 6");
  ms = time_ms([&] { hipLaunchKernelGGL(gemm_bf16x3_kernel<3>, grid, block, 0, 0, dA, dW, dC, M, N, K); }, reps);
  printf("  bf16 x2, 3 products    : %8.3f ms  %7.1f TFLOP/s (fp32-equivalent)   %.2fx the fp32 kernel\n", ms, flop / ms / 1e9, ms32 / ms);
  errors(hA, hW, dC, M, N, K, "bf16x2 / 3");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("  HIP error: %s\n", hipGetErrorString(e));
  hipFree(dA);
  hipFree(dW);
  hipFree(dC);
}

int main() {
  run(40960, 256, 256, "to_out-like 1x1 at 32x32 x 40 frames");
  run(40960, 512, 512, "B = 1 level-0-sized rows, 512 channels");
  run(327680, 256, 2304, "3x3 256->256 of a B = 8 training step as an im2col GEMM (C4's LFAE bottleneck)");
  run(8192, 4096, 4096, "square-ish saturating GEMM");
  return 0;
}
