// Device-side helpers shared by every kernel file: wavefront-64 shuffles, the two fp32 MFMA
// shapes of gfx950 (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32, exact f32 = fmaf chain,
// cdna_hip_programming.md section 3) and the launch macro.
//
// LFDM_EMU_BUILD is a TEST-ONLY switch: tests/emu/ compiles these same kernel sources for x86
// against a fiber emulator so that index arithmetic can be checked against the CPU oracle in a
// container without a GPU.  The product library is built by hipcc for gfx950 only.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <type_traits>

#if defined(LFDM_EMU_BUILD)
#include "hip_emu.h"
typedef emu_f32x16 f32x16;
typedef emu_f32x4 f32x4;
#define LFDM_LAUNCH(kern, grid, block, smem, stream, ...) \
  emu::launch(grid, block, smem, [=]() { kern(__VA_ARGS__); })
static inline f32x16 mfma_32x32x2(float a, float b, f32x16 c) { return emu_mfma_32x32x2(a, b, c); }
static inline f32x4 mfma_16x16x4(float a, float b, f32x4 c) { return emu_mfma_16x16x4(a, b, c); }
static inline f32x4 mfma_4x4x1(float a, float b, f32x4 c) { return emu_mfma_4x4x1(a, b, c); }
#else
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LFDM_LAUNCH(kern, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kern, grid, block, smem, stream, __VA_ARGS__)
// lane l: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]; D col = l&31, row = (r&3)+8*(r>>2)+4*(l>>5)
__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// lane l: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; D col = l&15, row = (l>>4)*4 + r
__device__ __forceinline__ f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// 16 independent 4x4 blocks (K = 1): block b = l>>2, A_b[i] from lane 4b+i, B_b[j] from lane 4b+j; D_b[i][j] = register i
// of lane 4b+j.  Full MFMA rate with only FOUR output columns per block: the shape for skinny-N contractions.
__device__ __forceinline__ f32x4 mfma_4x4x1(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}
#endif

#define LFDM_WAVE 64

// compile-time loop: f(std::integral_constant<int, I>) for I = BEGIN .. END-1 (indices usable as template / asm immediates)
template <int BEGIN, int END, class F>
__device__ __forceinline__ void lfdm_static_for(F&& f) {
  if constexpr (BEGIN < END) {
    f(std::integral_constant<int, BEGIN>{});
    lfdm_static_for<BEGIN + 1, END>(f);
  }
}

// Buffer-descriptor loads: an out-of-range byte offset returns zeros, which turns "is this filter tap
// inside the image" into one v_cndmask on the offset instead of a divergent branch around the load
// (branches inside the MFMA loop also made hipcc copy all accumulator registers every iteration).
#if defined(LFDM_EMU_BUILD)
struct lfdm_buf {
  const char* base;
  uint32_t bytes;
};
static inline lfdm_buf lfdm_make_buf(const void* p, uint32_t bytes) { return lfdm_buf{(const char*)p, bytes}; }
static inline float4 lfdm_buf_load_f4(lfdm_buf b, uint32_t off) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((uint64_t)off + 16u <= (uint64_t)b.bytes) memcpy(&v, b.base + off, 16);
  return v;
}
static inline float2 lfdm_buf_load_f2(lfdm_buf b, uint32_t off) {
  float2 v = make_float2(0.f, 0.f);
  if ((uint64_t)off + 8u <= (uint64_t)b.bytes) memcpy(&v, b.base + off, 8);
  return v;
}
#else
typedef __amdgpu_buffer_rsrc_t lfdm_buf;
typedef int lfdm_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ lfdm_buf lfdm_make_buf(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 lfdm_buf_load_f4(lfdm_buf b, uint32_t off) {
  const lfdm_i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b, (int)off, 0, 0);
  float4 f;
  f.x = __int_as_float(v.x);
  f.y = __int_as_float(v.y);
  f.z = __int_as_float(v.z);
  f.w = __int_as_float(v.w);
  return f;
}
typedef int lfdm_i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 lfdm_buf_load_f2(lfdm_buf b, uint32_t off) {
  const lfdm_i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(b, (int)off, 0, 0);
  return make_float2(__int_as_float(v.x), __int_as_float(v.y));
}
#endif
#define LFDM_BUF_OOB 0xFFFFFFF0u

// 16-byte WRITE-THROUGH store / L1-bypassing load through a buffer descriptor (sc1: cdna_hip_programming.md Guideline 16 form R1 - the payload of
// an in-launch hand-off between workgroups).  One 16-byte sc1 store is ONE fabric write; the same float4 as two 8-byte agent-scope atomic
// stores costs 2.7x the time per byte (MI355X_MICROARCH.md, "stores of each flavour") and an 8-byte sc1 load runs at 0.54-0.70x the
// 16-byte rate.  Ordering against the flag / ticket is the caller's: every storing wave drains (s_waitcnt vmcnt(0)) before the ticket.
#if defined(LFDM_EMU_BUILD)
static inline void lfdm_buf_store_f4_sc1(lfdm_buf b, uint32_t off, float4 v) {
  if ((uint64_t)off + 16u <= (uint64_t)b.bytes) memcpy(const_cast<char*>(b.base) + off, &v, 16);
}
static inline float4 lfdm_buf_load_f4_sc1(lfdm_buf b, uint32_t off) { return lfdm_buf_load_f4(b, off); }
#else
__device__ __forceinline__ void lfdm_buf_store_f4_sc1(lfdm_buf b, uint32_t off, float4 v) {
  lfdm_i32x4 d;
  d.x = __float_as_int(v.x); d.y = __float_as_int(v.y); d.z = __float_as_int(v.z); d.w = __float_as_int(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(d, b, (int)off, 0, 16);          // aux 16 = sc1 on gfx940+
}
__device__ __forceinline__ float4 lfdm_buf_load_f4_sc1(lfdm_buf b, uint32_t off) {
  const lfdm_i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b, (int)off, 0, 16);
  float4 f;
  f.x = __int_as_float(v.x); f.y = __int_as_float(v.y); f.z = __int_as_float(v.z); f.w = __int_as_float(v.w);
  return f;
}
#endif

// Register-operand load pipelines.  hipcc (ROCm 7.2) undoes a source-level "load group g+1, multiply group g" loop whenever the loaded
// values feed MFMAs straight from registers: the load is sunk across the back-edge to its use and every group pays the full memory
// latency (attn_lowres.hip, first version: 18.6 us for 3 us of MFMA).  The loads of such loops are therefore issued by inline asm
// (volatile: never moved or sunk) and retired by hand-counted s_waitcnt vmcnt(N) - loads return in order, N = the number of loads
// issued AFTER the ones being waited for; the waited registers are "+v" operands of the wait, so no consumer can be scheduled above
// it (cdna_hip_programming.md 5.4 rule 18, 5.7).  Only in fully unrolled straight-line code (no loop-carried asm outputs).
#if defined(LFDM_EMU_BUILD)
template <int OFF = 0> static inline void lfdm_gload_f4(f32x4& dst, const float* p) { memcpy(&dst, (const char*)p + OFF, 16); }
template <int N> static inline void lfdm_vmwait() {}
template <int N> static inline void lfdm_vmwait(f32x4&) {}
template <int N> static inline void lfdm_vmwait(f32x4&, f32x4&) {}
template <int N> static inline void lfdm_vmwait(f32x4&, f32x4&, f32x4&, f32x4&) {}
static inline void lfdm_tie(f32x4&) {}
static inline int lfdm_uniform(int v) { return v; }
#else
template <int OFF = 0>      // OFF: byte offset folded into the instruction (0 .. 4095)
__device__ __forceinline__ void lfdm_gload_f4(f32x4& dst, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void lfdm_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void lfdm_vmwait(f32x4& a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void lfdm_vmwait(f32x4& a, f32x4& b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void lfdm_vmwait(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
__device__ __forceinline__ void lfdm_tie(f32x4& a) { asm volatile("" : "+v"(a)::"memory"); }      // orders a consumer behind the preceding wait
// a value the compiler cannot prove wave-uniform (anything derived from threadIdx) made uniform for it: keeps buffer descriptors /
// scalar selects out of per-load "waterfall" loops (cdna_hip_programming.md T20)
__device__ __forceinline__ int lfdm_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif

// In-launch hand-off between workgroups (split-K: the last workgroup of a tile reduces the slabs).  Protocol of
// cdna_hip_programming.md section 6 Guideline 16 (counter form): every storing wave drains its stores, one lane issues an
// agent-scope release and takes a relaxed agent-scope ticket; the workgroup that draws the last ticket issues ONE
// agent-scope acquire before any of its waves reads the slabs.  Placement independent (8 XCDs, non-coherent L2s).
#if defined(LFDM_EMU_BUILD)
#define LFDM_DRAIN_STORES() ((void)0)
#define LFDM_FENCE_RELEASE_AGENT() ((void)0)
#define LFDM_FENCE_ACQUIRE_AGENT() ((void)0)
static inline unsigned lfdm_ticket_take(unsigned* c) { return __atomic_fetch_add(c, 1u, __ATOMIC_SEQ_CST); }   // workgroups run on several OS threads
static inline void lfdm_ticket_reset(unsigned* c) { __atomic_store_n(c, 0u, __ATOMIC_SEQ_CST); }
#else
#define LFDM_DRAIN_STORES() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define LFDM_FENCE_RELEASE_AGENT() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define LFDM_FENCE_ACQUIRE_AGENT() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
__device__ __forceinline__ unsigned lfdm_ticket_take(unsigned* c) {
  return __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lfdm_ticket_reset(unsigned* c) {
  __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

// Agent-scope words shared by the workgroups of ONE launch without a fence (cdna_hip_programming.md section 6 Guideline 16, the "8-byte agent
// atomics on both sides" form: write-through sc1 stores, L1-bypassing loads): payload granules and arrival counters of the cooperative
// split-K reduce + GroupNorm kernel (norm.hip).
// WHAT THIS RESTS ON.  Relaxed agent-scope stores -> `s_waitcnt vmcnt(0)` -> relaxed ticket RMW -> relaxed agent-scope loads carry no
// release / acquire, so the HIP / LLVM memory model promises no happens-before between payload and ticket.  The hand-off is correct on
// gfx950 because of how that target lowers and executes it (MI355X_MICROARCH.md, "inter-workgroup visibility": sc1 stores are written
// through and acknowledged at memory before vmcnt drops; sc1 loads and atomics are served past the per-XCD L2s; observed untorn for 8-byte
// granules) - hardware behaviour validated on gfx950 / ROCm 7.2, not an architectural guarantee.  Hence the compile-time guard below (the
// library is built for gfx950 only, _build.py), the 128-byte boundary the launcher rounds every slab base up to (two column tiles on different
// XCDs never share a cache line), and tests/test_ops_parity.py::test_wino_fused_reduce_stress (bit-equal to the separate reduce pass over
// many launches under load).  A ticket array belongs to ONE stream: launches that share it must be ordered.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(LFDM_EMU_BUILD)
#error "fence-free hand-off (lfdm_agent_store_u64 / lfdm_agent_load_u64 / lfdm_ticket_take) validated on gfx950 only: re-validate, or use LFDM_FENCE_RELEASE_AGENT / LFDM_FENCE_ACQUIRE_AGENT, before building for another target"
#endif
#if defined(LFDM_EMU_BUILD)
static inline void lfdm_agent_store_u64(unsigned long long* p, unsigned long long v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long lfdm_agent_load_u64(const unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
static inline unsigned lfdm_agent_load_u32(const unsigned* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
static inline void lfdm_sleep() {}
#else
__device__ __forceinline__ void lfdm_agent_store_u64(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long lfdm_agent_load_u64(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned lfdm_agent_load_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lfdm_sleep() { __builtin_amdgcn_s_sleep(2); }
#endif

// Ordering point for LDS traffic that stays INSIDE one wavefront (a wave-private LDS region written by some lanes and read by others): the
// LDS unit executes a wave's operations in order, so all that is needed is that the compiler keeps them in program order and that every
// lane has issued its writes - no workgroup barrier, the other waves of the workgroup are not involved.
#if defined(LFDM_EMU_BUILD)
static inline void lfdm_wave_lds_sync() { emu::wave_sync(); }
#else
__device__ __forceinline__ void lfdm_wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x / (1.0f + expf(-x)); }

// SiLU on the hardware exponential / reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each): for activations recomputed inside a consumer's
// load path (lfdm_conv_params.gn_in_*), where the IEEE division + expf expansion of siluf_ would cost more than the launch it saves
#if defined(LFDM_EMU_BUILD)
static inline float silu_fast_(float x) { return x / (1.0f + expf(-x)); }
#else
__device__ __forceinline__ float silu_fast_(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
#endif

// Tuning knobs.  Schedule constants that were found by sweeps (tile shapes, workgroup counts, thresholds) can be overridden from the
// environment ONLY in a library built with -DLFDM_TUNING_KNOBS (python -m cvpr23_lfdm_amd._build hip --knobs; tools/sweep_*.sh, tools/bench_*.py and
// the tests that force a tile shape ask the library whether it has them: lfdm_has_tuning_knobs).  The shipped build compiles every one
// of them to its default: no getenv on a launch path, no variant the parity suite does not run.
#if defined(LFDM_TUNING_KNOBS) || defined(LFDM_EMU_BUILD)
static inline const char* lfdm_knob(const char* name) { return getenv(name); }
#else
static inline const char* lfdm_knob(const char*) { return nullptr; }
#endif

// activation codes shared by the C ABI (include/lfdm_hip.h)
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return v > 0.f ? v : 0.f;
  if (act == 2) return sigmoidf_(v);
  if (act == 3) return siluf_(v);
  return v;
}

// host-side error plumbing (lfdm_capi.cpp)
extern "C" void lfdm_set_error(const char* msg);
int lfdm_check_launch(const char* what);
