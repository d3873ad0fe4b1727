// TEST INFRASTRUCTURE ONLY - see hip_emu.h.
#include "hip_emu.h"

#include <atomic>
#include <thread>

// lfdm_switch(&save_sp, to_sp): store the callee-saved registers on the current stack, publish the stack pointer, continue
// on the other stack (System V x86-64; glibc's swapcontext also saves the signal mask - a system call per switch, and the
// MFMA emulation switches twice per instruction and lane).
extern "C" void lfdm_switch(void** save_sp, void* to_sp);
asm(R"(
.text
.globl lfdm_switch
.type lfdm_switch,@function
lfdm_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size lfdm_switch,.-lfdm_switch
)");

namespace emu {

thread_local State g;

static const size_t kStackBytes = 256 * 1024;
static thread_local std::vector<Fiber> fibers;
static thread_local std::vector<char*> stacks;
static thread_local const std::function<void()>* cur_body = nullptr;
static thread_local unsigned live_threads = 0;

void yield() { lfdm_switch(&g.cur->sp, g.sched_sp); }

void syncthreads() {
  unsigned my_gen = g.bar_gen;
  g.bar_arrived++;
  if (g.bar_arrived >= live_threads) {
    g.bar_arrived = 0;
    g.bar_gen++;
    return;
  }
  while (g.bar_gen == my_gen) yield();
}

void wave_sync() {
  WaveState& w = my_wave();
  unsigned my_gen = w.gen;
  w.arrived++;
  if (w.arrived >= w.size) {
    w.arrived = 0;
    w.gen++;
    return;
  }
  while (w.gen == my_gen) yield();
}

static void fiber_entry() {
  (*cur_body)();
  Fiber* f = g.cur;
  f->done = true;
  // an exited thread no longer takes part in barriers / wave rendezvous
  live_threads--;
  if (live_threads > 0 && g.bar_arrived >= live_threads && g.bar_arrived > 0) {
    g.bar_arrived = 0;
    g.bar_gen++;
  }
  WaveState& w = g.waves[f->linear >> 6];
  w.size--;
  if (w.size > 0 && w.arrived >= w.size && w.arrived > 0) {
    w.arrived = 0;
    w.gen++;
  }
  lfdm_switch(&f->sp, g.sched_sp);
  abort();          // a finished fiber is never resumed
}

// one workgroup on the calling OS thread
static void run_block(dim3 grid, dim3 block, size_t dyn_smem_bytes, const std::function<void()>& body, unsigned bx, unsigned by,
                      unsigned bz) {
  const unsigned nt = block.x * block.y * block.z;
  if (fibers.size() < nt) fibers.resize(nt);
  while (stacks.size() < nt) stacks.push_back((char*)malloc(kStackBytes));
  std::vector<unsigned char> dyn(dyn_smem_bytes + 64);
  g.dyn_smem = dyn.data();
  g.gdim = grid;
  g.bdim = block;
  g.nthreads = nt;
  cur_body = &body;
  const unsigned nwaves = (nt + 63) / 64;
  g.bidx = dim3(bx, by, bz);
  g.bar_arrived = 0;
  g.bar_gen = 0;
  live_threads = nt;
  g.waves.assign(nwaves, WaveState());
  for (unsigned wv = 0; wv < nwaves; ++wv) {
    unsigned rem = nt - wv * 64;
    g.waves[wv].size = rem < 64 ? rem : 64;
  }
  for (unsigned t = 0; t < nt; ++t) {
    Fiber& f = fibers[t];
    f.done = false;
    f.linear = t;
    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    // initial frame: six callee-saved registers, then fiber_entry as the return address of lfdm_switch; after its `ret`
    // rsp = top - 8, i.e. 8 (mod 16) like after a call
    uintptr_t top = ((uintptr_t)stacks[t] + kStackBytes) & ~(uintptr_t)15;
    void** sp = (void**)(top - 16 - 48);
    for (int i = 0; i < 6; ++i) sp[i] = nullptr;
    sp[6] = (void*)fiber_entry;
    sp[7] = nullptr;
    f.sp = (void*)sp;
  }
  unsigned remaining = nt;
  unsigned long spins = 0;
  while (remaining > 0) {
    for (unsigned t = 0; t < nt; ++t) {
      Fiber& f = fibers[t];
      if (f.done) continue;
      g.cur = &f;
      lfdm_switch(&g.sched_sp, f.sp);
      if (f.done) remaining--;
    }
    if (++spins > 50000000ul) {
      fprintf(stderr, "emu: deadlock suspected in block (%u,%u,%u)\n", bx, by, bz);
      abort();
    }
  }
  g.dyn_smem = nullptr;
}

static unsigned worker_count() {
  static const unsigned n = [] {
    unsigned v = std::thread::hardware_concurrency();
    if (v == 0) v = 1;
    if (v > 8) v = 8;
    if (const char* e = getenv("LFDM_EMU_THREADS")) v = (unsigned)atoi(e);
    return v < 1 ? 1u : v;
  }();
  return n;
}

void launch(dim3 grid, dim3 block, size_t dyn_smem_bytes, const std::function<void()>& body) {
  const uint64_t nblocks = (uint64_t)grid.x * grid.y * grid.z;
  unsigned nw = worker_count();
  if (nblocks < nw) nw = (unsigned)nblocks;
  auto decode = [&](uint64_t i, unsigned& bx, unsigned& by, unsigned& bz) {
    bx = (unsigned)(i % grid.x);
    by = (unsigned)((i / grid.x) % grid.y);
    bz = (unsigned)(i / ((uint64_t)grid.x * grid.y));
  };
  if (nw <= 1) {
    for (uint64_t i = 0; i < nblocks; ++i) {
      unsigned bx, by, bz;
      decode(i, bx, by, bz);
      run_block(grid, block, dyn_smem_bytes, body, bx, by, bz);
    }
    return;
  }
  // workgroups dealt dynamically to the workers (the real machine promises no order either)
  std::atomic<uint64_t> next(0);
  auto work = [&]() {
    for (;;) {
      const uint64_t i = next.fetch_add(1);
      if (i >= nblocks) break;
      unsigned bx, by, bz;
      decode(i, bx, by, bz);
      run_block(grid, block, dyn_smem_bytes, body, bx, by, bz);
    }
  };
  std::vector<std::thread> pool;
  for (unsigned w = 1; w < nw; ++w) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
}

}  // namespace emu
