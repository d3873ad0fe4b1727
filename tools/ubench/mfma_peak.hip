// Micro-benchmark: peak issue rate of v_mfma_f32_32x32x2_f32 with NACC independent accumulators per wave,
// 1 or 2 waves per SIMD. Build twice (with / without -mllvm -amdgpu-mfma-vgpr-form).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int blocks, const char* tag) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.f, 2.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)blocks * 4 * iters * NACC * 2.0 * 32 * 32 * 2;
  printf("%s NACC=%d blocks=%d: %.3f ms  %.1f TFLOP/s\n", tag, NACC, blocks, ms, flop / ms / 1e9);
  hipFree(out);
}
int main(int argc, char** argv) {
  const char* tag = argc > 1 ? argv[1] : "";
  run<1>(256, tag); run<2>(256, tag); run<4>(256, tag); run<10>(256, tag);
  run<1>(512, tag); run<4>(512, tag); run<1>(1024, tag);
  return 0;
}
