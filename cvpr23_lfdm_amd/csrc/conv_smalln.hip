// Convolution with at most FOUR output channels (include/lfdm_hip.h: lfdm_conv2d_smalln_cl_f32): the LFAE generator's
// final 7x7 RGB projection (LFAE/modules/generator.py:54,161-162: Conv2d(64 -> 3, 7x7) + sigmoid) over every decoded frame.
//
// On the 32-column MFMA tiles this layer wastes 7/8 of the matrix unit (3 real columns of 32): 8 ms per 320-frame
// decode, 10 % of a DM training step.  v_mfma_f32_4x4x1 runs sixteen independent 4x4 blocks per instruction at the same
// MAC rate, so with lane = output pixel (A operand), B = the four filters, every multiply is useful:
//   out[pixel 4b+i][co j] += x[pixel 4b+i + tap][c] * w[tap][c][j]          (block b = lane>>2)
// A workgroup owns a 16x16 pixel tile of one image; per 16-channel chunk the (16+k-1)^2 input patch and the chunk's
// weights are staged in LDS (patch rows padded to 20 floats: 16-byte fragment reads are conflict free), and each
// wave walks over all taps for its 4 pixel rows: one ds_read_b128 of the patch + one of the weights per 4 MFMAs.
#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

constexpr int TS = 16;              // output tile edge
constexpr int CK = 16;              // channels per chunk
constexpr int PLD = CK + 4;         // patch row stride (floats)
constexpr int MAXK = 7;

__global__ __launch_bounds__(256) void conv_smalln_kernel(const float* __restrict__ x, int ldx, int cin, int n_img, int h, int w,
                                                          const float* __restrict__ wgt,   // [k*k][cin][4]
                                                          const float* __restrict__ bias,  // [4]
                                                          float* __restrict__ out, int ldo, int cout, int k, int act) {
  constexpr int PS = TS + MAXK - 1;                                  // patch edge (max)
  __shared__ __attribute__((aligned(16))) float patch[PS * PS * PLD];
  __shared__ __attribute__((aligned(16))) float wl[MAXK * MAXK * 4 * CK];   // [tap][j][c]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (w + TS - 1) / TS, tiles_y = (h + TS - 1) / TS;
  const int img = blockIdx.x / (tiles_x * tiles_y);
  const int trem = blockIdx.x - img * tiles_x * tiles_y;
  const int ty0 = (trem / tiles_x) * TS, tx0 = (trem - (trem / tiles_x) * tiles_x) * TS;
  const int pad = k / 2, ps = TS + k - 1;
  // this lane's output pixel: wave -> 4 rows of the tile, lane -> (row, col)
  const int py = 4 * wave + (lane >> 4), px = lane & 15;
  const int j = lane & 3;                                            // B operand: filter index

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < cin; c0 += CK) {
    __syncthreads();
    // ---- stage the input patch (zeros outside the image) and the weight chunk ----
    // (round 6: batches of eight / four 16-byte loads per thread, unconditional - a pixel outside the image reads a clamped address and stores zeros.  One
    //  guarded load per trip of a strided loop made every trip a memory round trip of its own: eight in a row for the patch, twelve scalar ones for the
    //  weights, per 16-channel chunk - 1.9 ms for a launch that moves 1.3 GB.)
    {
      constexpr int SU = 8;
      const int total = ps * ps * (CK / 4);
      for (int f0 = tid; f0 < total; f0 += 256 * SU) {
        float4 v[SU];
        bool ok[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const int f = f0 + 256 * u, fc = f < total ? f : total - 1;
          const int pix = fc / (CK / 4), c4 = fc - pix * (CK / 4);
          const int yy = ty0 + pix / ps - pad, xx = tx0 + (pix - (pix / ps) * ps) - pad;
          ok[u] = yy >= 0 && yy < h && xx >= 0 && xx < w;
          const int yc = yy < 0 ? 0 : (yy >= h ? h - 1 : yy), xc = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
          v[u] = *reinterpret_cast<const float4*>(x + (((int64_t)img * h + yc) * w + xc) * ldx + c0 + 4 * c4);
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const int f = f0 + 256 * u;
          if (f < total) {
            const int pix = f / (CK / 4), c4 = f - pix * (CK / 4);
            *reinterpret_cast<float4*>(patch + pix * PLD + 4 * c4) = ok[u] ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
    }
    {
      // weights [tap][cin][4] -> wl[tap][j][c]: one 16-byte load per (tap, channel), its four filters scattered
      constexpr int WU = 4;
      const int total = k * k * CK;
      for (int f0 = tid; f0 < total; f0 += 256 * WU) {
        float4 v[WU];
#pragma unroll
        for (int u = 0; u < WU; ++u) {
          const int f = f0 + 256 * u, fc = f < total ? f : total - 1;
          const int tap = fc / CK, c = fc - tap * CK;
          v[u] = *reinterpret_cast<const float4*>(wgt + ((int64_t)tap * cin + c0 + c) * 4);
        }
#pragma unroll
        for (int u = 0; u < WU; ++u) {
          const int f = f0 + 256 * u;
          if (f < total) {
            const int tap = f / CK, c = f - tap * CK;
            float* dst = wl + tap * 4 * CK + c;
            dst[0] = v[u].x; dst[CK] = v[u].y; dst[2 * CK] = v[u].z; dst[3 * CK] = v[u].w;
          }
        }
      }
    }
    __syncthreads();
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) {
        const float* pa = patch + ((py + ky) * ps + px + kx) * PLD;
        const float* pb = wl + ((ky * k + kx) * 4 + j) * CK;
#pragma unroll
        for (int q = 0; q < CK / 4; ++q) {
          const float4 a = *reinterpret_cast<const float4*>(pa + 4 * q);
          const float4 b = *reinterpret_cast<const float4*>(pb + 4 * q);
          acc = mfma_4x4x1(a.x, b.x, acc);
          acc = mfma_4x4x1(a.y, b.y, acc);
          acc = mfma_4x4x1(a.z, b.z, acc);
          acc = mfma_4x4x1(a.w, b.w, acc);
        }
      }
  }
  // D: this lane holds filter j of the pixels 4*(lane>>2) + i of its wave
  if (j < cout) {
    const float bb = bias ? bias[j] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int l = 4 * (lane >> 2) + i;                              // the lane whose pixel this is
      const int oy = ty0 + 4 * wave + (l >> 4), ox = tx0 + (l & 15);
      if (oy < h && ox < w) out[(((int64_t)img * h + oy) * w + ox) * ldo + j] = apply_act(acc[i] + bb, act);
    }
  }
}

}  // namespace

extern "C" int lfdm_conv2d_smalln_cl_f32(const float* x, int ldx, int cin, int n_img, int h, int w, const float* wgt,
                                         const float* bias, float* out, int ldo, int cout, int k, int act,
                                         lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !wgt || !out || cin <= 0 || cin % 16 != 0 || ldx < cin || ldx % 4 != 0 || (((uintptr_t)x) & 15) || (((uintptr_t)wgt) & 15) || n_img <= 0 ||
      h <= 0 || w <= 0 || cout < 1 || cout > 4 || ldo < cout || k < 1 || k > MAXK || (k & 1) == 0) {
    lfdm_set_error("conv2d_smalln: needs cout <= 4, odd k <= 7, C_in % 16 == 0, 16-byte aligned rows and filters");
    return LFDM_EINVAL;
  }
  const int64_t tiles = (int64_t)n_img * ((h + TS - 1) / TS) * ((w + TS - 1) / TS);
  if (tiles >= (1ll << 31)) { lfdm_set_error("conv2d_smalln: too many tiles"); return LFDM_EINVAL; }
  LFDM_LAUNCH(conv_smalln_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, x, ldx, cin, n_img, h, w, wgt, bias, out, ldo, cout,
              k, act);
  return lfdm_check_launch("conv2d_smalln");
}
