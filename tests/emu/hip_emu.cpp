// TEST INFRASTRUCTURE ONLY - see hip_emu.h.
#include "hip_emu.h"

namespace emu {

State g;

static const size_t kStackBytes = 256 * 1024;
static std::vector<Fiber> fibers;
static std::vector<char*> stacks;
static const std::function<void()>* cur_body = nullptr;
static unsigned live_threads = 0;

void yield() { swapcontext(&g.cur->ctx, &g.sched); }

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
  swapcontext(&f->ctx, &g.sched);
}

void launch(dim3 grid, dim3 block, size_t dyn_smem_bytes, const std::function<void()>& body) {
  unsigned nt = block.x * block.y * block.z;
  if (fibers.size() < nt) fibers.resize(nt);
  while (stacks.size() < nt) stacks.push_back((char*)malloc(kStackBytes));
  std::vector<unsigned char> dyn(dyn_smem_bytes + 64);
  g.dyn_smem = dyn.data();
  g.gdim = grid;
  g.bdim = block;
  g.nthreads = nt;
  cur_body = &body;
  unsigned nwaves = (nt + 63) / 64;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
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
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = stacks[t];
          f.ctx.uc_stack.ss_size = kStackBytes;
          f.ctx.uc_link = nullptr;
          makecontext(&f.ctx, (void (*)())fiber_entry, 0);
        }
        unsigned remaining = nt;
        unsigned long spins = 0;
        while (remaining > 0) {
          unsigned progressed = 0;
          for (unsigned t = 0; t < nt; ++t) {
            Fiber& f = fibers[t];
            if (f.done) continue;
            g.cur = &f;
            swapcontext(&g.sched, &f.ctx);
            if (f.done) {
              remaining--;
              progressed++;
            }
          }
          if (++spins > 50000000ul) {
            fprintf(stderr, "emu: deadlock suspected in block (%u,%u,%u)\n", bx, by, bz);
            abort();
          }
          (void)progressed;
        }
      }
  g.dyn_smem = nullptr;
}

}  // namespace emu
