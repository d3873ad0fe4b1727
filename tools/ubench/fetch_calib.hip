// Calibration of rocprofv3's FETCH_SIZE on gfx950 per access width (MI355X_MICROARCH.md, HBM: "exactly 1/2 for 16 B/lane streaming reads,
// other widths uncalibrated"): streaming reads of a 2 GiB buffer (larger than L2 + Infinity Cache) with 4 / 8 / 16 bytes per lane and
// as the 8-byte-per-lane strided gather of the Winograd patch loads (8 adjacent lanes = one 64-byte segment, segments 1 KB apart).
// Run under `rocprofv3 --pmc FETCH_SIZE` (tools/prof_fetch_calib.sh): expected bytes per kernel = 2^31.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fetch_calib.bin tools/ubench/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__global__ __launch_bounds__(256) void stream_read(const T* __restrict__ p, size_t n, float* out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  float s = 0.f;
  for (; i < n; i += stride) {
    const T v = p[i];
    if constexpr (sizeof(T) == 4) s += v;
    else if constexpr (sizeof(T) == 8) s += v.x + v.y;
    else s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) out[0] = s;
}

// 8 lanes x 8 B = one 64-byte segment; a wave covers 8 segments that lie `seg_stride` bytes apart; every byte of the buffer is read once
__global__ __launch_bounds__(256) void gather64_read(const char* __restrict__ p, size_t bytes, size_t seg_stride, float* out) {
  const size_t nseg = bytes / 64, segs_per_row = seg_stride / 64;      // rows of seg_stride bytes
  size_t g = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 3;          // segment slot
  const int l8 = threadIdx.x & 7;
  const size_t stride = ((size_t)gridDim.x * 256) >> 3;
  float s = 0.f;
  for (; g < nseg; g += stride) {
    // slot g -> (row = g % rows, segment-in-row = g / rows): consecutive slots are seg_stride apart
    const size_t rows = nseg / segs_per_row;
    const size_t row = g % rows, sg = g / rows;
    const f32x2 v = *reinterpret_cast<const f32x2*>(p + row * seg_stride + sg * 64 + l8 * 8);
    s += v.x + v.y;
  }
  if (s == 123.456f) out[0] = s;
}

// the filter-fragment loads of conv_wino.hip / conv_wino4.hip: lane (column = l & 31, k-half = l >> 5) reads 16 bytes at column * 64 + k-half * 32,
// then the 16 bytes behind them (two instructions per 2 KB of a wave; every byte read once)
__global__ __launch_bounds__(256) void frag16_read(const char* __restrict__ p, size_t bytes, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  size_t blk = ((size_t)blockIdx.x * 4 + wave);                       // 2 KB block per wave and trip
  const size_t nblk = bytes / 2048, stride = (size_t)gridDim.x * 4;
  float s = 0.f;
  for (; blk < nblk; blk += stride) {
    const char* q = p + blk * 2048 + (lane & 31) * 64 + (lane >> 5) * 32;
    const f32x4 a = *reinterpret_cast<const f32x4*>(q), b = *reinterpret_cast<const f32x4*>(q + 16);
    s += a.x + a.w + b.y + b.z;
  }
  if (s == 123.456f) out[0] = s;
}

int main() {
  const size_t bytes = (size_t)1 << 31;
  char* buf;
  float* out;
  hipMalloc(&buf, bytes);
  hipMalloc(&out, 4);
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  const int blocks = 256 * 16;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(stream_read<float>, dim3(blocks), dim3(256), 0, 0, (const float*)buf, bytes / 4, out);
    hipLaunchKernelGGL(stream_read<f32x2>, dim3(blocks), dim3(256), 0, 0, (const f32x2*)buf, bytes / 8, out);
    hipLaunchKernelGGL(stream_read<f32x4>, dim3(blocks), dim3(256), 0, 0, (const f32x4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(gather64_read, dim3(blocks), dim3(256), 0, 0, (const char*)buf, bytes, (size_t)1024, out);
    hipLaunchKernelGGL(frag16_read, dim3(blocks), dim3(256), 0, 0, (const char*)buf, bytes, out);
  }
  hipDeviceSynchronize();
  printf("fetch_calib: 5 kernels x 2 reps, %zu bytes each\n", bytes);
  return 0;
}
