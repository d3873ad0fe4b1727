// LFAE warp kernels (include/lfdm_hip.h: lfdm_warp_cl_f32 / lfdm_warp_planar_f32).
// One pass fuses what the reference does in 3-5 ATen calls per use
// (LFAE/modules/generator.py:59-88): bilinear up-sampling of the low-resolution sampling grid and
// occlusion map (never materialised at feature resolution), grid_sample (bilinear, zeros padding,
// align_corners=False) and the occlusion blend  out = warped*occ + prev*(1-occ).
// HBM-bound: per output element 4 B gathered + 4 B written (+4 B prev).
//  - channels-last variant: one thread per (pixel, 4 channels); the four taps are contiguous
//    C-vectors, so gathers are 16 B per lane and whole rows per wavefront.
//  - planar variant (reference layout; source image planes): a workgroup stages one source
//    channel plane (<= 128x128 fp32 = 64 KB) in LDS once and produces that channel for a block of
//    frames, so the random-access taps hit LDS and every source byte is read from HBM once per
//    frame block instead of once per tap.
#include <stdlib.h>

#include "lfdm_device.h"
#include "../../include/lfdm_hip.h"

namespace {

struct PixelTaps {
  int x0, y0;          // north-west tap (may be out of range)
  float wx0, wx1, wy0, wy1;
  float occ;
};

// ATen upsample_bilinear2d (align_corners=False) source index
__device__ __forceinline__ void resize_src(int dst, int n_in, int n_out, int& i0, int& i1, float& l1) {
  const float scale = (float)n_in / (float)n_out;
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

__device__ __forceinline__ float bilerp(const float* m, int fw, int y0, int y1, int x0, int x1,
                                        float ly, float lx) {
  const float v00 = m[y0 * fw + x0], v01 = m[y0 * fw + x1];
  const float v10 = m[y1 * fw + x0], v11 = m[y1 * fw + x1];
  return (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

__device__ __forceinline__ PixelTaps pixel_taps(const lfdm_warp_params& p, int b, int t, int oy, int ox) {
  const int64_t moff = (int64_t)b * p.fsb + (int64_t)t * p.fst;
  int y0, y1, x0, x1;
  float ly, lx;
  resize_src(oy, p.fh, p.h, y0, y1, ly);
  resize_src(ox, p.fw, p.w, x0, x1, lx);
  float gx, gy;
  if (p.fh == p.h && p.fw == p.w) {
    gx = p.flow_x[moff + oy * p.fw + ox];
    gy = p.flow_y[moff + oy * p.fw + ox];
  } else {
    gx = bilerp(p.flow_x + moff, p.fw, y0, y1, x0, x1, ly, lx);
    gy = bilerp(p.flow_y + moff, p.fw, y0, y1, x0, x1, ly, lx);
  }
  PixelTaps r;
  r.occ = 1.f;
  if (p.occ) {
    float o;
    if (p.fh == p.h && p.fw == p.w) o = p.occ[moff + oy * p.fw + ox] * p.occ_scale + p.occ_bias;
    else {
      // the reference resizes the [0,1] occlusion map; the affine map commutes with the
      // (convex) bilinear weights up to rounding, apply it on the taps to follow the reference
      const float* m = p.occ + moff;
      const float v00 = m[y0 * p.fw + x0] * p.occ_scale + p.occ_bias;
      const float v01 = m[y0 * p.fw + x1] * p.occ_scale + p.occ_bias;
      const float v10 = m[y1 * p.fw + x0] * p.occ_scale + p.occ_bias;
      const float v11 = m[y1 * p.fw + x1] * p.occ_scale + p.occ_bias;
      o = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
    r.occ = o;
  }
  // grid_sample unnormalise (align_corners=False)
  float ix = ((gx + 1.f) * (float)p.w - 1.f) * 0.5f;
  float iy = ((gy + 1.f) * (float)p.h - 1.f) * 0.5f;
  // far-out-of-range samples contribute zeros either way; keep the int conversion defined
  ix = fminf(fmaxf(ix, -1.0e6f), 1.0e6f);
  iy = fminf(fmaxf(iy, -1.0e6f), 1.0e6f);
  const float fx = floorf(ix), fy = floorf(iy);
  r.x0 = (int)fx;
  r.y0 = (int)fy;
  r.wx1 = ix - fx;
  r.wx0 = (fx + 1.f) - ix;
  r.wy1 = iy - fy;
  r.wy0 = (fy + 1.f) - iy;
  return r;
}

// ---------------- channels-last ----------------
// A pixel is served by G = C/(4*R) adjacent lanes; lane j owns the float4 chunks j, j+G, ... (R of
// them), so the taps / occlusion (12 low-res map reads + the bilinear set-up) are computed once per
// 4*R channels and every access of the G lanes is one contiguous G*16-byte segment.
// grid (blocks over one frame's hw * g lane items, frames): the frame index comes from blockIdx.y (scalar unit) and the
// per-lane decomposition is 32-bit with g = 1, 2, 4, ... a power of two in every LFAE shape (shift instead of divide) - the
// earlier flat 64-bit index cost three 64-bit divisions per lane item, ~40 % of the kernel's VALU time.
template <int R>
__global__ __launch_bounds__(256) void warp_cl_kernel(lfdm_warp_params p) {
  const int g = (p.c >> 2) / R;             // lanes per pixel
  const int hw = p.h * p.w;
  const int per_frame = hw * g;
  const int gshift = (g & (g - 1)) == 0 ? __builtin_ctz(g) : -1;
  for (int n = blockIdx.y; n < p.batch * p.frames; n += gridDim.y)
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < per_frame; idx += gridDim.x * 256) {
    const int pix = gshift >= 0 ? (idx >> gshift) : idx / g;
    const int c = (idx - pix * g) * 4;
    const int64_t gp = (int64_t)n * hw + pix;   // output pixel row
    const int b = n / p.frames, t = n - b * p.frames;
    const int oy = pix / p.w, ox = pix - oy * p.w;
    const PixelTaps tp = pixel_taps(p, b, t, oy, ox);
    const float* sb = p.src + (int64_t)b * hw * p.ld_src + c;
    // tap offsets/weights (weight 0 + clamped address for out-of-range taps: zeros padding)
    float wgt[4];
    int64_t off[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = tp.y0 + (k >> 1), xx = tp.x0 + (k & 1);
      const bool in = yy >= 0 && yy < p.h && xx >= 0 && xx < p.w;
      wgt[k] = in ? ((k & 1) ? tp.wx1 : tp.wx0) * ((k >> 1) ? tp.wy1 : tp.wy0) : 0.f;
      off[k] = in ? (int64_t)(yy * p.w + xx) * p.ld_src : 0;
    }
    float4 pv[R];
    if (p.prev) {
#pragma unroll
      for (int r = 0; r < R; ++r) pv[r] = *reinterpret_cast<const float4*>(p.prev + gp * p.ld_prev + c + r * 4 * g);
    }
    float4 v[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[r][k] = *reinterpret_cast<const float4*>(sb + off[k] + r * 4 * g);
    const float o = tp.occ, om = 1.f - o;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 4; ++k) {       // same tap order as before: (y0,x0) (y0,x1) (y1,x0) (y1,x1)
        acc.x = fmaf(v[r][k].x, wgt[k], acc.x);
        acc.y = fmaf(v[r][k].y, wgt[k], acc.y);
        acc.z = fmaf(v[r][k].z, wgt[k], acc.z);
        acc.w = fmaf(v[r][k].w, wgt[k], acc.w);
      }
      if (p.occ) {
        if (p.prev) {
          acc.x = acc.x * o + pv[r].x * om;
          acc.y = acc.y * o + pv[r].y * om;
          acc.z = acc.z * o + pv[r].z * om;
          acc.w = acc.w * o + pv[r].w * om;
        } else {
          acc.x *= o; acc.y *= o; acc.z *= o; acc.w *= o;
        }
      }
      *reinterpret_cast<float4*>(p.out + gp * p.ld_out + c + r * 4 * g) = acc;
    }
  }
}

// ---------------- planar ----------------
// grid (frame_blocks, C, B).  The source plane (b, c) is staged in LDS when it fits.
constexpr int PLANE_MAX = 128 * 128;

template <bool STAGED>
__global__ __launch_bounds__(256) void warp_planar_kernel(lfdm_warp_params p, int fpb) {
  __shared__ __attribute__((aligned(16))) float plane[STAGED ? PLANE_MAX : 4];
  const int b = blockIdx.z, c = blockIdx.y;
  const int hw = p.h * p.w;
  const float* sp = p.src + ((int64_t)b * p.c + c) * hw;
  if (STAGED) {
    for (int i = threadIdx.x * 4; i < hw; i += 256 * 4)
      *reinterpret_cast<float4*>(plane + i) = *reinterpret_cast<const float4*>(sp + i);
    __syncthreads();
  }
  const float* S = STAGED ? plane : sp;
  const int t0 = blockIdx.x * fpb;
  const int t1 = t0 + fpb < p.frames ? t0 + fpb : p.frames;
  for (int t = t0; t < t1; ++t) {
    const int64_t obase = (((int64_t)b * p.c + c) * p.frames + t) * hw;
    for (int pix = threadIdx.x; pix < hw; pix += 256) {
      const int oy = pix / p.w, ox = pix - oy * p.w;
      const PixelTaps tp = pixel_taps(p, b, t, oy, ox);
      float acc = 0.f;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const int yy = tp.y0 + dy;
        if (yy < 0 || yy >= p.h) continue;
        const float wy = dy ? tp.wy1 : tp.wy0;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int xx = tp.x0 + dx;
          if (xx < 0 || xx >= p.w) continue;
          acc = fmaf(S[yy * p.w + xx], (dx ? tp.wx1 : tp.wx0) * wy, acc);
        }
      }
      if (p.occ) {
        const float o = tp.occ;
        if (p.prev) {
          const float pv = p.prev_is_cl
                               ? p.prev[(((int64_t)b * p.frames + t) * hw + pix) * p.ld_prev + c]
                               : p.prev[obase + pix];
          acc = acc * o + pv * (1.f - o);
        } else {
          acc *= o;
        }
      }
      p.out[obase + pix] = acc;
    }
  }
}

// Few source planes (the RGB image, C = 3): one thread per output pixel computes the taps once and
// produces up to 4 channels; the 196 KB source stays in L1/L2, outputs are coalesced plane rows.
// grid (ceil(hw/256), ceil(C/4), B*T).
__global__ __launch_bounds__(256) void warp_planar_pixel_kernel(lfdm_warp_params p) {
  const int hw = p.h * p.w;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= hw) return;
  const int n = blockIdx.z;
  const int b = n / p.frames, t = n - b * p.frames;
  const int oy = pix / p.w, ox = pix - oy * p.w;
  const PixelTaps tp = pixel_taps(p, b, t, oy, ox);
  float wgt[4];
  int off[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int yy = tp.y0 + (k >> 1), xx = tp.x0 + (k & 1);
    const bool in = yy >= 0 && yy < p.h && xx >= 0 && xx < p.w;
    wgt[k] = in ? ((k & 1) ? tp.wx1 : tp.wx0) * ((k >> 1) ? tp.wy1 : tp.wy0) : 0.f;
    off[k] = in ? yy * p.w + xx : 0;
  }
  const int c0 = blockIdx.y * 4;
  const int c1 = c0 + 4 < p.c ? c0 + 4 : p.c;
  const float o = tp.occ;
  for (int c = c0; c < c1; ++c) {
    const float* S = p.src + ((int64_t)b * p.c + c) * hw;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc = fmaf(S[off[k]], wgt[k], acc);
    const int64_t obase = (((int64_t)b * p.c + c) * p.frames + t) * hw;
    if (p.occ) {
      if (p.prev) {
        const float pv = p.prev_is_cl ? p.prev[(((int64_t)b * p.frames + t) * hw + pix) * p.ld_prev + c]
                                      : p.prev[obase + pix];
        acc = acc * o + pv * (1.f - o);
      } else {
        acc *= o;
      }
    }
    p.out[obase + pix] = acc;
  }
}

}  // namespace

static int check_warp(const lfdm_warp_params* p, const char* who) {
  if (!p || !p->src || !p->out || !p->flow_x || !p->flow_y || p->batch <= 0 || p->frames <= 0 ||
      p->h <= 0 || p->w <= 0 || p->c <= 0 || p->fh <= 0 || p->fw <= 0) {
    lfdm_set_error(who);
    return LFDM_EINVAL;
  }
  return LFDM_OK;
}

extern "C" int lfdm_warp_cl_f32(const lfdm_warp_params* pp, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_warp(pp, "warp_cl: bad arguments");
  if (rc) return rc;
  lfdm_warp_params p = *pp;
  if (p.c % 4 != 0 || p.ld_src % 4 != 0 || p.ld_out % 4 != 0 || (p.prev && p.ld_prev % 4 != 0) ||
      p.ld_src < p.c || p.ld_out < p.c) {
    lfdm_set_error("warp_cl: channels and row strides must be multiples of 4");
    return LFDM_EINVAL;
  }
  int r = (p.c % 16 == 0) ? 4 : (p.c % 8 == 0 ? 2 : 1);
  if (const char* e = lfdm_knob("LFDM_WARP_R")) {      // experiment knob (bench.py warp object): lanes per pixel = C / (4 r)
    const int v = atoi(e);
    if ((v == 1 || v == 2 || v == 4) && p.c % (4 * v) == 0) r = v;
  }
  const int64_t per_frame = (int64_t)p.h * p.w * (p.c / (4 * r));
  if (per_frame >= (1ll << 31) - 256 * 65536) { lfdm_set_error("warp_cl: frame too large"); return LFDM_EINVAL; }
  int64_t nbx = (per_frame + 255) / 256;
  if (nbx > 65536) nbx = 65536;
  int64_t nby = (int64_t)p.batch * p.frames;
  if (nby > 65535) nby = 65535;
  const dim3 grid((unsigned)nbx, (unsigned)nby);
  if (r == 4) LFDM_LAUNCH((warp_cl_kernel<4>), grid, dim3(256), 0, stream, p);
  else if (r == 2) LFDM_LAUNCH((warp_cl_kernel<2>), grid, dim3(256), 0, stream, p);
  else LFDM_LAUNCH((warp_cl_kernel<1>), grid, dim3(256), 0, stream, p);
  return lfdm_check_launch("warp_cl");
}

extern "C" int lfdm_warp_planar_f32(const lfdm_warp_params* pp, lfdm_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_warp(pp, "warp_planar: bad arguments");
  if (rc) return rc;
  lfdm_warp_params p = *pp;
  const int hw = p.h * p.w;
  // frames per workgroup: amortise the plane staging, but keep >= ~512 workgroups in flight
  int fpb = (int)(((int64_t)p.frames * p.c * p.batch) / 512);
  if (fpb < 1) fpb = 1;
  if (fpb > 8) fpb = 8;
  if ((int64_t)p.c * p.batch * p.frames < 2048 && (int64_t)p.batch * p.frames <= 65535) {
    const dim3 pgrid((hw + 255) / 256, (p.c + 3) / 4, p.batch * p.frames);
    LFDM_LAUNCH(warp_planar_pixel_kernel, pgrid, dim3(256), 0, stream, p);
    return lfdm_check_launch("warp_planar");
  }
  const dim3 grid((p.frames + fpb - 1) / fpb, p.c, p.batch), block(256);
  const bool staged = hw <= PLANE_MAX && (hw % 4) == 0 && (((uintptr_t)p.src & 15) == 0);
  if (staged) LFDM_LAUNCH((warp_planar_kernel<true>), grid, block, 0, stream, p, fpb);
  else LFDM_LAUNCH((warp_planar_kernel<false>), grid, block, 0, stream, p, fpb);
  return lfdm_check_launch("warp_planar");
}
