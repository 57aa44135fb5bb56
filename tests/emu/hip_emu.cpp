// hip_emu.cpp — TEST INFRASTRUCTURE ONLY (see hip_emu.hpp).
#include "hip_emu.hpp"

#include <omp.h>

thread_local uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

namespace
{
constexpr size_t STACK_BYTES = 256 * 1024;

struct Fiber
{
  ucontext_t ctx;
  char *stack = nullptr;
  bool done = true;
};
struct Runner
{
  ucontext_t main;
  std::vector<Fiber> fibers;
  int current = -1;     // fiber index running, -1 = main context
  bool in_fiber = false;
  const std::function<void()> *body = nullptr;
  bool yielded = false;
};
thread_local Runner runner;

void trampoline()
{
  Runner &r = runner;
  (*r.body)();
  r.fibers[r.current].done = true;
  swapcontext(&r.fibers[r.current].ctx, &r.main);
}

inline void set_tid(unsigned t, const dim3 &b)
{
  threadIdx.x = t % b.x;
  threadIdx.y = (t / b.x) % b.y;
  threadIdx.z = t / (b.x * b.y);
}

void run_block(const dim3 &block, const std::function<void()> &body)
{
  Runner &r = runner;
  const unsigned n = block.x * block.y * block.z;
  if(r.fibers.size() < n)
    r.fibers.resize(n);
  r.body = &body;
  // thread 0 first, as a fiber: tells us whether the kernel synchronises
  auto start_fiber = [&](unsigned t) {
    Fiber &f = r.fibers[t];
    if(!f.stack)
      f.stack = (char *)std::malloc(STACK_BYTES);
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = STACK_BYTES;
    f.ctx.uc_link = nullptr;
    f.done = false;
    makecontext(&f.ctx, trampoline, 0);
  };
  auto resume = [&](unsigned t) {
    r.current = (int)t;
    r.in_fiber = true;
    set_tid(t, block);
    swapcontext(&r.main, &r.fibers[t].ctx);
    r.in_fiber = false;
    r.current = -1;
  };
  start_fiber(0);
  resume(0);
  if(r.fibers[0].done)
    {
      // no barrier reached by thread 0: run the rest as plain calls
      for(unsigned t = 1; t < n; ++t)
        {
          set_tid(t, block);
          body();
        }
      return;
    }
  for(unsigned t = 1; t < n; ++t)
    {
      start_fiber(t);
      resume(t);
    }
  for(;;)
    {
      bool any = false;
      for(unsigned t = 0; t < n; ++t)
        if(!r.fibers[t].done)
          {
            any = true;
            resume(t);
          }
      if(!any)
        break;
    }
}
} // namespace

void __syncthreads()
{
  Runner &r = runner;
  if(!r.in_fiber)
    {
      std::fprintf(stderr, "hip_emu: __syncthreads() reached by a thread after thread 0 exited "
                           "without synchronising (divergent barrier)\n");
      std::abort();
    }
  const int t = r.current;
  swapcontext(&r.fibers[t].ctx, &r.main);
}

void hip_emu_launch(dim3 grid, dim3 block, const std::function<void()> &body)
{
  const long nblocks = (long)grid.x * grid.y * grid.z;
  if(nblocks <= 0 || block.x * block.y * block.z == 0)
    return;
#pragma omp parallel for schedule(dynamic, 1) if(nblocks > 1)
  for(long b = 0; b < nblocks; ++b)
    {
      gridDim = grid;
      blockDim = block;
      blockIdx.x = (unsigned)(b % grid.x);
      blockIdx.y = (unsigned)((b / grid.x) % grid.y);
      blockIdx.z = (unsigned)(b / ((long)grid.x * grid.y));
      run_block(block, body);
    }
}
