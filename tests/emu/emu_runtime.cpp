// TEST INFRASTRUCTURE ONLY -- see hip/hip_runtime.h in this directory.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <sys/mman.h>
#include <ucontext.h>

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
thread_local EmuBlock *emu_block;
thread_local unsigned emu_phase;

// Every thread of the running workgroup is a fiber of the OS thread that launched the kernel. A fiber runs until it waits
// at a barrier; the wait hands the CPU to the next live fiber (round robin). Stacks are mapped once and reused.
namespace {
constexpr size_t STACK_BYTES = 256 << 10;
constexpr unsigned MAX_FIBERS = 1024;

struct Fiber {
  ucontext_t ctx;
  unsigned tid = 0, phase = 0;
  bool done = true;
  const EmuBarrier *waiting = nullptr;   // (diagnostics) what it is waiting at
  unsigned waiting_generation = 0;
};
struct Sched {
  Fiber fib[MAX_FIBERS];
  char *stacks = nullptr;
  ucontext_t main_ctx;
  unsigned n = 0, cur = 0, live = 0;
  unsigned long events = 0;            // arrivals, completed barriers, exits: a full round without any = deadlock
  const std::function<void()> *body = nullptr;
};
thread_local Sched *S;

Sched *sched() {
  if (!S) {
    S = new Sched();   // (kept for the life of the thread)
    S->stacks = (char *)mmap(nullptr, STACK_BYTES * MAX_FIBERS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (S->stacks == MAP_FAILED) { perror("emu: mmap of the fiber stacks"); abort(); }
  }
  return S;
}

void enter(unsigned i) {      // make fiber i the running one (registers of the GPU thread it stands for)
  Sched *s = S;
  s->cur = i;
  threadIdx = dim3(s->fib[i].tid);
  emu_phase = s->fib[i].phase;
}

// hand the CPU to the next live fiber; returns when this fiber is scheduled again
void yield_from(unsigned me) {
  Sched *s = S;
  s->fib[me].phase = emu_phase;
  unsigned nx = me;
  for (unsigned step = 0; step < s->n; step++) {
    nx = nx + 1 == s->n ? 0 : nx + 1;
    if (!s->fib[nx].done) break;
  }
  if (nx == me) return;       // alone
  enter(nx);
  swapcontext(&s->fib[me].ctx, &s->fib[nx].ctx);
  // (scheduled again: enter(me) was done by whoever switched to us)
}

void fiber_main() {
  Sched *s = S;
  const unsigned me = s->cur;
  (*s->body)();
  // this GPU thread has left the kernel: it no longer takes part in its barriers
  Fiber &f = s->fib[me];
  f.done = true;
  s->live--;
  s->events++;
  emu_block->bar.live--;
  emu_block->waves[f.tid / 64].bar.live--;
  // someone else goes on (a waiter whose barrier is now complete notices when it runs); the last one returns to the launcher
  for (unsigned step = 1; step <= s->n; step++) {
    const unsigned nx = (me + step) % s->n;
    if (!s->fib[nx].done) { enter(nx); setcontext(&s->fib[nx].ctx); }
  }
  setcontext(&s->main_ctx);
}
}  // namespace

void emu_barrier_wait(EmuBarrier *b) {
  Sched *s = S;
  const unsigned me = s->cur;
  const unsigned g = b->generation;
  b->arrived++;
  s->events++;
  for (;;) {
    if (b->generation != g) return;
    if (b->arrived >= b->live) { b->arrived = 0; b->generation++; s->events++; return; }
    s->fib[me].waiting = b; s->fib[me].waiting_generation = g;
    const unsigned long ev = s->events;
    yield_from(me);
    // every other live fiber has run since: if none of them arrived anywhere, completed a barrier or left, all are waiting
    if (s->events == ev && b->generation == g && b->arrived < b->live) {
      fprintf(stderr, "emu: deadlock -- thread %u of workgroup %u waits at a barrier (%u of %u arrived) that the other live threads never reach\n",
              s->fib[me].tid, blockIdx.x, b->arrived, b->live);
      abort();
    }
  }
}

void emu_launch(dim3 grid, dim3 block, const std::function<void()> &body) {
  const unsigned nthr = block.x, nw = (nthr + 63) / 64;
  if (nthr == 0 || nthr > MAX_FIBERS || nw > 16) { fprintf(stderr, "emu: workgroup of %u threads\n", nthr); abort(); }
  Sched *s = sched();
  EmuBlock eb;
  emu_block = &eb;
  blockDim = block; gridDim = grid;
  s->body = &body;
  s->n = nthr;
  for (unsigned b = 0; b < grid.x; b++) {
    blockIdx = dim3(b);
    eb.bar = EmuBarrier{nthr, 0, 0};
    for (unsigned w = 0; w < nw; w++) eb.waves[w].bar = EmuBarrier{std::min(64u, nthr - w * 64), 0, 0};
    for (unsigned i = 0; i < nthr; i++) {
      Fiber &f = s->fib[i];
      f.tid = i; f.phase = 0; f.done = false; f.waiting = nullptr;
      getcontext(&f.ctx);
      f.ctx.uc_stack.ss_sp = s->stacks + (size_t)i * STACK_BYTES;
      f.ctx.uc_stack.ss_size = STACK_BYTES;
      f.ctx.uc_link = nullptr;
      makecontext(&f.ctx, fiber_main, 0);
    }
    s->live = nthr;
    enter(0);
    swapcontext(&s->main_ctx, &s->fib[0].ctx);     // returns when the last fiber of the workgroup has finished
  }
  emu_block = nullptr;
}
