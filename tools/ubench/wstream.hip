// What does the FILTER STREAM of a low-resolution Winograd launch cost by itself?  (round 4: K-group workgroups changed the per-chunk time
// of conv_wino_kernel at 4x4 far less than a latency model predicts - every configuration ends near 20-25 GB/s per CU - so what bounds
// the stream: the strided 2 KB pieces of the [pos][chunk][cout][16] pack, the five tile-block siblings that re-read every slice, or
// HBM latency itself?)  The kernels below issue exactly the filter-fragment loads of conv_wino_kernel<.,1> for the 512 -> 512 @4x4
// launch of a B = 1 step (grid 5 x 16 x ksplit, 256 threads, lane (co = l31, kh) reads two 16-byte pieces per position and chunk)
// and nothing else - no patches, no LDS, no MFMA; results are xor-folded into one store per thread.
//   layout 0: the library's pack  U[16 pos][nch][coutp][16]   (2 KB pieces, 32 KB apart per chunk, 1 MB apart per position)
//   layout 1: per column tile     U[coutp/32][nch][16 pos][32][16]  (a workgroup's K slice is ONE contiguous range)
//   sib = 5: all five tile-block workgroups read the slice (the real launch); sib = 1: only tile block 0 does (no redundancy)
//   depth: chunks requested ahead (1 = like the kernel: next chunk only; 4 = four chunks in flight)
// A ring of 20 packs (336 MB > the 256 MB Infinity Cache) keeps the filters cold.  hipcc --offload-arch=gfx950 -O3 -o wstream.bin wstream.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int CIN = 512, COUT = 512, NCH = CIN / 16, POS = 16;
typedef float f4 __attribute__((ext_vector_type(4)));

template <int LAYOUT, int DEPTH>
__global__ __launch_bounds__(256) void wstream_kernel(const float* __restrict__ w, float* __restrict__ out, int ksplit, int sib, int remap) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (remap) {      // the library's XCD-aware order (conv_wino.hip): XCD k (= linear id % 8) owns the column tiles k, k + 8
    const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned j = L >> 3, ny8 = gridDim.y >> 3;
    by = (L & 7u) + 8u * (j % ny8);
    const unsigned rest = j / ny8;
    bx = rest % gridDim.x;
    bz = rest / gridDim.x;
  }
  if (bx >= sib) return;
  const int kc0 = NCH * bz / ksplit, kc1 = NCH * (bz + 1) / ksplit;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  auto addr = [&](int pos, int chunk) -> const f4* {
    const int64_t off = LAYOUT == 0 ? ((((int64_t)pos * NCH + chunk) * COUT + by * 32 + l31) * 16 + 8 * kh)
                                    : (((((int64_t)by * NCH + chunk) * POS + pos) * 32 + l31) * 16 + 8 * kh);
    return reinterpret_cast<const f4*>(w + off);
  };
  f4 buf[DEPTH][4][2];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) {
      const int c = kc0 + d < kc1 ? kc0 + d : kc1 - 1;
      buf[d][pi][0] = addr(4 * wave + pi, c)[0];
      buf[d][pi][1] = addr(4 * wave + pi, c)[1];
    }
  for (int kc = kc0; kc < kc1; kc += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int nx = kc + d + DEPTH < kc1 ? kc + d + DEPTH : kc1 - 1;
#pragma unroll
      for (int pi = 0; pi < 4; ++pi) {
        acc += buf[d][pi][0] + buf[d][pi][1];
        buf[d][pi][0] = addr(4 * wave + pi, nx)[0];
        buf[d][pi][1] = addr(4 * wave + pi, nx)[1];
      }
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) acc += buf[d][pi][0] + buf[d][pi][1];
  out[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 256 + tid] = acc.x + acc.y + acc.z + acc.w;
}

template <int LAYOUT, int DEPTH>
static void run(float** packs, int ring, float* out, int ksplit, int sib, const char* what, int remap = 1) {
  const dim3 grid(5, COUT / 32, ksplit), block(256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < ring; ++i) hipLaunchKernelGGL((wstream_kernel<LAYOUT, DEPTH>), grid, block, 0, 0, packs[i], out, ksplit, sib, remap);
  const int reps = 5;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r)
    for (int i = 0; i < ring; ++i) hipLaunchKernelGGL((wstream_kernel<LAYOUT, DEPTH>), grid, block, 0, 0, packs[i], out, ksplit, sib, remap);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / (reps * ring), mb = 16.0 * CIN * COUT * 4 / 1e6;
  printf("%-34s remap %d layout %d depth %d ksplit %d siblings %d: %7.2f us per launch  (%.1f MB unique -> %.2f TB/s unique, %.2f TB/s L1-side)\n", what, remap, LAYOUT, DEPTH,
         ksplit, sib, us, mb, mb / us, mb * sib / us);
}

int main() {
  const int ring = 20;
  const size_t n = (size_t)16 * CIN * COUT;
  float* packs[ring];
  float* h = (float*)malloc(n * 4);
  unsigned s = 777u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) * (1.0f / 8388608.0f) - 1.0f; }
  for (int i = 0; i < ring; ++i) { hipMalloc(&packs[i], n * 4); hipMemcpy(packs[i], h, n * 4, hipMemcpyHostToDevice); }
  float* out;
  hipMalloc(&out, (size_t)5 * 16 * 8 * 256 * 4);
  for (int ksplit : {6, 3, 8}) {
    run<0, 1>(packs, ring, out, ksplit, 5, "library pack, next chunk ahead");
    run<0, 1>(packs, ring, out, ksplit, 1, "library pack, one sibling");
    run<0, 4>(packs, ring, out, ksplit, 5, "library pack, 4 chunks ahead");
    run<1, 1>(packs, ring, out, ksplit, 5, "contiguous slices, next chunk ahead");
    run<1, 1>(packs, ring, out, ksplit, 1, "contiguous slices, one sibling");
    run<1, 4>(packs, ring, out, ksplit, 5, "contiguous slices, 4 chunks ahead");
    run<1, 4>(packs, ring, out, ksplit, 1, "contiguous, 4 ahead, one sibling");
    run<0, 1>(packs, ring, out, ksplit, 5, "library pack, NATURAL block order", 0);
  }
  return 0;
}
