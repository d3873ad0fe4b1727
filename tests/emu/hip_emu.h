// TEST INFRASTRUCTURE ONLY.
// A tiny fiber emulator of the HIP execution model (workgroups, 64-wide wavefronts, LDS,
// __syncthreads, cross-lane shuffles and the fp32 MFMA shapes the kernels use).  One fiber per
// HIP thread (a 7-instruction x86-64 stack switch, no syscall); the workgroups of a launch are
// dealt to a few OS threads (LFDM_EMU_THREADS, default: the cores, at most 8), each with its own
// fibers, scheduler state and LDS (`__shared__` = static thread_local).  It lets the kernel sources under cvpr23_lfdm_amd/csrc/ be compiled for x86 and their
// index arithmetic checked against the CPU oracle in the build container, which has no GPU.
// It is NOT a backend: the product library (liblfdm_hip.so) is built by hipcc for gfx950 only
// and never contains this file.  MFMA fragment layouts follow
// /opt/skills/guides/cdna_hip_programming.md section 3.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define LFDM_EMU 1

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__
#endif

namespace emu {

struct Fiber {
  void* sp;         // saved stack pointer while the fiber is not running
  bool done;
  dim3 tid;
  unsigned linear;  // linear thread id in the block
};

struct WaveState {
  alignas(16) unsigned char slot[64][64];  // 64 lanes x up to 64 bytes
  unsigned arrived = 0;
  unsigned gen = 0;
  unsigned size = 64;
};

struct State {
  dim3 bidx, bdim, gdim;
  Fiber* cur = nullptr;
  void* sched_sp = nullptr;
  unsigned nthreads = 0;
  unsigned bar_arrived = 0, bar_gen = 0;
  std::vector<WaveState> waves;
  unsigned char* dyn_smem = nullptr;
};
extern thread_local State g;

void yield();
void syncthreads();
void wave_sync();
void launch(dim3 grid, dim3 block, size_t dyn_smem_bytes, const std::function<void()>& body);

static inline unsigned lane_id() { return g.cur->linear & 63u; }
static inline WaveState& my_wave() { return g.waves[g.cur->linear >> 6]; }

template <class T>
static inline T wave_read_from(T v, unsigned src_lane) {
  static_assert(sizeof(T) <= 64, "slot too small");
  WaveState& w = my_wave();
  memcpy(w.slot[lane_id()], &v, sizeof(T));
  wave_sync();
  T r;
  memcpy(&r, w.slot[src_lane & 63u], sizeof(T));
  wave_sync();
  return r;
}

}  // namespace emu

#define threadIdx (emu::g.cur->tid)
#define blockIdx (emu::g.bidx)
#define blockDim (emu::g.bdim)
#define gridDim (emu::g.gdim)
static inline void __syncthreads() { emu::syncthreads(); }

template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; return emu::wave_read_from(v, emu::lane_id() ^ (unsigned)mask); }
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) { (void)width; unsigned s = emu::lane_id() + d; return emu::wave_read_from(v, s < 64 ? s : emu::lane_id()); }
template <class T> static inline T __shfl(T v, int src, int width = 64) { (void)width; return emu::wave_read_from(v, (unsigned)src); }

// real atomics: workgroups of one launch run on several OS threads
static inline float atomicAdd(float* p, float v) {
  unsigned* u = reinterpret_cast<unsigned*>(p);
  unsigned old = __atomic_load_n(u, __ATOMIC_RELAXED), want;
  float f;
  do {
    memcpy(&f, &old, 4);
    f += v;
    memcpy(&want, &f, 4);
  } while (!__atomic_compare_exchange_n(u, &old, want, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED));
  memcpy(&f, &old, 4);
  return f;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicMax(unsigned* p, unsigned v) {
  unsigned o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v > o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) {}
  return o;
}
static inline unsigned atomicMin(unsigned* p, unsigned v) {
  unsigned o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v < o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) {}
  return o;
}

static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline float fminf_(float a, float b) { return a < b ? a : b; }

typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: A[i][k] from lane i+32k, B[k][j] from lane j+32k;
// D: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); k-ordered fmaf chain.
static inline emu_f32x16 emu_mfma_32x32x2(float a, float b, emu_f32x16 c) {
  emu::WaveState& w = emu::my_wave();
  unsigned lane = emu::lane_id();
  float ab[2] = {a, b};
  memcpy(w.slot[lane], ab, 8);
  emu::wave_sync();
  unsigned col = lane & 31u;
  for (int r = 0; r < 16; ++r) {
    unsigned row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    float acc = c[r];
    for (int k = 0; k < 2; ++k) {
      float av, bv;
      memcpy(&av, w.slot[row + 32 * k], 4);
      memcpy(&bv, w.slot[col + 32 * k] + 4, 4);
      acc = fmaf(av, bv, acc);
    }
    c[r] = acc;
  }
  emu::wave_sync();
  return c;
}

// v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4 blocks, K = 1.  Block b = lane>>2: A_b[i] from lane 4b+i,
// B_b[j] from lane 4b+j; D_b[i][j]: register i of lane 4b+j.
static inline emu_f32x4 emu_mfma_4x4x1(float a, float b, emu_f32x4 c) {
  emu::WaveState& w = emu::my_wave();
  unsigned lane = emu::lane_id();
  float ab[2] = {a, b};
  memcpy(w.slot[lane], ab, 8);
  emu::wave_sync();
  const unsigned blk = lane >> 2;
  float bv;
  memcpy(&bv, w.slot[lane] + 4, 4);                 // B_b[j = lane&3] is this lane's own b
  for (int i = 0; i < 4; ++i) {
    float av;
    memcpy(&av, w.slot[4 * blk + i], 4);
    c[i] = fmaf(av, bv, c[i]);
  }
  emu::wave_sync();
  return c;
}

// v_mfma_f32_16x16x4_f32: A[i][k] from lane i+16k, B[k][j] from lane j+16k;
// D: col = lane&15, row = (lane>>4)*4 + reg.
static inline emu_f32x4 emu_mfma_16x16x4(float a, float b, emu_f32x4 c) {
  emu::WaveState& w = emu::my_wave();
  unsigned lane = emu::lane_id();
  float ab[2] = {a, b};
  memcpy(w.slot[lane], ab, 8);
  emu::wave_sync();
  unsigned col = lane & 15u;
  for (int r = 0; r < 4; ++r) {
    unsigned row = (lane >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      memcpy(&av, w.slot[row + 16 * k], 4);
      memcpy(&bv, w.slot[col + 16 * k] + 4, 4);
      acc = fmaf(av, bv, acc);
    }
    c[r] = acc;
  }
  emu::wave_sync();
  return c;
}
