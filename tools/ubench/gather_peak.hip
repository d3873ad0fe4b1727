// What can the LFAE warp's ACCESS PATTERN reach on this chip with no arithmetic at all?
// (`warp_cl_kernel` sits at 0.61 of the 8 TB/s HBM peak on algorithmic bytes for three rounds, its counters say "issue
//  stalled on the vector-memory queue", and nothing proved a ceiling.)  This program keeps the kernel's instruction mix per lane item -
//  TAPS x R 16-byte gathers from a 4 MB source map at the four bilinear taps of a pixel displaced by a smooth pseudo-random flow,
//  R 16-byte streaming loads of `prev`, R 16-byte stores - and drops everything else: no low-resolution map reads, no bilinear set-up,
//  no occlusion blend.  Shape = the dominant warp launch of a 40-frame decode: 40 frames x 128 x 128 pixels x 64 channels, source
//  (128 x 128 x 64) shared by all frames, 4 lanes per pixel (R = 4 float4 chunks per lane, chunk stride 4 lanes).
//    variants: TAPS = 4 (the kernel's mix), 2, 1, 0 (TAPS = 0: prev -> out copy = the streaming rate at this launch geometry)
//    bytes counted like bench.py's warp roofline: 12 B per output element with `prev` (4 gathered once + 4 prev + 4 written), 8 without.
//  hipcc --offload-arch=gfx950 -O3 -o gather_peak.bin gather_peak.hip && ./gather_peak.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int H = 128, W = 128, C = 64, FRAMES = 40, R = 4, G = C / 4 / R;   // G = 4 lanes per pixel

template <int TAPS, bool PREV>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ src, const float* __restrict__ prev, float* __restrict__ out,
                                                     const int* __restrict__ disp) {
  const int hw = H * W, per_frame = hw * G;
  for (int n = blockIdx.y; n < FRAMES; n += gridDim.y)
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < per_frame; idx += gridDim.x * 256) {
      const int pix = idx >> 2, c = (idx & 3) * 4;
      const int oy = pix >> 7, ox = pix & 127;
      // displacement of this pixel (one packed int per pixel and frame: dy in the high half, dx in the low half; |d| <= 6 like
      // identity + 0.1 randn on a 128-pixel map) - ONE 4-byte load instead of the kernel's 12 low-resolution map reads
      const int d = disp[n * hw + pix];
      int y0 = oy + (d >> 16), x0 = ox + (int)(short)(d & 0xffff);
      y0 = y0 < 0 ? 0 : (y0 > H - 2 ? H - 2 : y0);
      x0 = x0 < 0 ? 0 : (x0 > W - 2 ? W - 2 : x0);
      const int64_t gp = (int64_t)n * hw + pix;
      float4 pv[R], v[R][TAPS > 0 ? TAPS : 1];
      if (PREV) {
#pragma unroll
        for (int r = 0; r < R; ++r) pv[r] = *reinterpret_cast<const float4*>(prev + gp * C + c + r * 4 * G);
      }
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < TAPS; ++k)
          v[r][k] = *reinterpret_cast<const float4*>(src + (int64_t)((y0 + (k >> 1)) * W + x0 + (k & 1)) * C + c + r * 4 * G);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float4 a = PREV ? pv[r] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < TAPS; ++k) { a.x += v[r][k].x; a.y += v[r][k].y; a.z += v[r][k].z; a.w += v[r][k].w; }
        *reinterpret_cast<float4*>(out + gp * C + c + r * 4 * G) = a;
      }
    }
}

template <int TAPS, bool PREV>
static void run(const float* src, const float* prev, float* out, const int* disp, int gx, const char* what) {
  const dim3 grid(gx, FRAMES), block(256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gather_kernel<TAPS, PREV>), grid, block, 0, 0, src, prev, out, disp);
  const int reps = 20;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gather_kernel<TAPS, PREV>), grid, block, 0, 0, src, prev, out, disp);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, elems = (double)FRAMES * H * W * C;
  const double bytes = elems * (4.0 * (TAPS > 0) + 4.0 * PREV + 4.0);
  printf("%-58s grid_x %4d  %7.1f us  %6.2f TB/s algorithmic (%4.1f B/elem)  %5.3f of 8 TB/s\n", what, gx, us, bytes / us / 1e6,
         bytes / elems, bytes / us / 1e6 / 8.0);
}

int main() {
  const size_t n_out = (size_t)FRAMES * H * W * C, n_src = (size_t)H * W * C;
  float *src, *prev, *out;
  int* disp;
  hipMalloc(&src, n_src * 4);
  hipMalloc(&prev, n_out * 4);
  hipMalloc(&out, n_out * 4);
  hipMalloc(&disp, (size_t)FRAMES * H * W * 4);
  float* h = (float*)malloc(n_out * 4);
  unsigned s = 12345u;
  for (size_t i = 0; i < n_out; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) * (1.0f / 8388608.0f) - 1.0f; }
  hipMemcpy(prev, h, n_out * 4, hipMemcpyHostToDevice);
  hipMemcpy(src, h, n_src * 4, hipMemcpyHostToDevice);
  int* hd = (int*)malloc((size_t)FRAMES * H * W * 4);
  for (int i = 0; i < FRAMES * H * W; ++i) {       // sum of three uniforms in [-2, 2] each: a bell of width ~ +-6
    int dy = 0, dx = 0;
    for (int k = 0; k < 3; ++k) { s = s * 1664525u + 1013904223u; dy += (int)((s >> 16) % 5u) - 2; dx += (int)((s >> 24) % 5u) - 2; }
    hd[i] = (dy << 16) | (dx & 0xffff);
  }
  hipMemcpy(disp, hd, (size_t)FRAMES * H * W * 4, hipMemcpyHostToDevice);
  for (int gx : {64, 256}) {       // 64 x 40 = 2560 workgroups (the library's launch), 256 x 40 = one item per thread
    run<4, true>(src, prev, out, disp, gx, "4 taps x4 + prev + store (warp_cl_kernel's instruction mix)");
    run<2, true>(src, prev, out, disp, gx, "2 taps x4 + prev + store");
    run<1, true>(src, prev, out, disp, gx, "1 tap  x4 + prev + store");
    run<0, true>(src, prev, out, disp, gx, "prev -> out copy (no gather)");
    run<4, false>(src, prev, out, disp, gx, "4 taps x4 + store, no prev (pure warp)");
  }
  return 0;
}
