// Box calibration (include/lfdm_hip.h: lfdm_calib_mfma_f32): a pure v_mfma_f32_32x32x2_f32 loop on every SIMD of the chip that also
// reads the shader-cycle counter and the constant 100 MHz real-time counter, so that bench.py can print - next to every timing - the
// fp32 matrix rate THIS box delivers right now and the clock it holds under that load.  Two boxes of the same pool differed by 13 % on
// the same commit in round 3 with nothing on the bench line to tell a slow box from a slow build; this is that missing number.
// Not on the product path (nothing in the samplers / training step calls it).
#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

#if defined(LFDM_EMU_BUILD)
static inline unsigned long long calib_cycles() { return 0ull; }
static inline unsigned long long calib_realtime() { return 0ull; }
static inline void calib_pin(float) {}
#else
__device__ __forceinline__ unsigned long long calib_cycles() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ unsigned long long calib_realtime() { return wall_clock64(); }      // s_memrealtime: 100 MHz
__device__ __forceinline__ void calib_pin(float v) { asm volatile("" ::"v"(v) : "memory"); }    // orders the second stamp behind the loop's results
#endif

// one wave per SIMD (256 threads per workgroup, one workgroup per CU), four independent accumulator chains: 99 % of the issue rate
// (MI355X_MICROARCH.md "Matrix cores").  out[2 * block] = shader cycles, out[2 * block + 1] = 100 MHz ticks of the timed loop of
// wave 0; out[2 * gridDim.x + thread] keeps the accumulators alive.
__global__ __launch_bounds__(256) void calib_mfma_kernel(float* __restrict__ out, int iters, float a0, float b0) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // per-lane pseudo-random operands in [-1, 1), a different pair per chain (constant or zero operands would let the chip clock higher
  // than any real kernel does: MI355X_MICROARCH.md "DVFS give-back")
  float a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned h = (threadIdx.x * 4u + (unsigned)i + blockIdx.x * 1024u) * 2654435761u;
    a[i] = a0 * ((float)((h >> 8) & 0xffffu) * (1.0f / 32768.0f) - 1.0f);
    b[i] = b0 * ((float)((h >> 12) & 0xffffu) * (1.0f / 32768.0f) - 1.0f);
  }
  const unsigned long long c0 = calib_cycles(), w0 = calib_realtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = mfma_32x32x2(a[i], b[i], acc[i]);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  calib_pin(s);
  const unsigned long long c1 = calib_cycles(), w1 = calib_realtime();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = (float)(c1 - c0);
    out[2 * blockIdx.x + 1] = (float)(w1 - w0);
  }
  if (s == 12345.678f) out[2 * gridDim.x + threadIdx.x] = s;      // never true for these operands; keeps the loop
}

}  // namespace

extern "C" int lfdm_calib_mfma_f32(float* out, int blocks, int iters, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!out || blocks <= 0 || iters <= 0) {
    lfdm_set_error("calib_mfma: out must hold 2 * blocks + 256 floats; blocks, iters > 0");
    return LFDM_EINVAL;
  }
  LFDM_LAUNCH(calib_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, out, iters, 1.0f, 0.5f);
  return lfdm_check_launch("calib_mfma");
}
