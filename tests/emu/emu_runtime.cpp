// TEST INFRASTRUCTURE ONLY -- see hip/hip_runtime.h in this directory.
#include <hip/hip_runtime.h>

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
thread_local EmuBlock *emu_block;
thread_local unsigned emu_phase;

// Workgroups run one after another on a pool of OS threads that persists across launches (creating and joining
// 64..1024 threads per workgroup used to dominate the CPU suite's run time).
#include <condition_variable>
#include <mutex>

namespace {
struct Pool {
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> workers;
  unsigned generation = 0, active = 0, remaining = 0;
  unsigned block = 0;
  dim3 grid, blockdim;
  EmuBlock *eb = nullptr;
  const std::function<void()> *body = nullptr;
  void worker(unsigned id) {
    unsigned seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(mu);
      cv_work.wait(lk, [&] { return generation != seen; });
      seen = generation;
      if (id >= active) continue;
      const std::function<void()> *fn = body;
      threadIdx = dim3(id); blockIdx = dim3(block); blockDim = blockdim; gridDim = grid;
      emu_block = eb;
      emu_phase = 0;
      lk.unlock();
      (*fn)();
      eb->bar->arrive_and_drop();
      lk.lock();
      if (--remaining == 0) cv_done.notify_one();
    }
  }
  void grow(unsigned n) {
    while (workers.size() < n) {
      const unsigned id = (unsigned)workers.size();
      workers.emplace_back([this, id] { worker(id); });
      workers.back().detach();
    }
  }
};
Pool *pool() { static Pool *p = new Pool(); return p; }   // leaked on purpose: its threads outlive static destruction
}  // namespace

void emu_launch(dim3 grid, dim3 block, const std::function<void()> &body) {
  const unsigned nthr = block.x, nw = (nthr + 63) / 64;
  EmuBlock *eb = new EmuBlock();
  for (unsigned w = 0; w < nw; w++) {
    unsigned cnt = std::min(64u, nthr - w * 64);
    pthread_barrier_init(&eb->waves[w].bar, nullptr, cnt);
  }
  Pool *P = pool();
  {
    std::lock_guard<std::mutex> lk(P->mu);
    P->grow(nthr);
  }
  for (unsigned b = 0; b < grid.x; b++) {
    std::barrier<> bar(nthr);
    eb->bar = &bar;
    std::unique_lock<std::mutex> lk(P->mu);
    P->active = nthr; P->remaining = nthr; P->block = b; P->grid = grid; P->blockdim = block; P->eb = eb; P->body = &body;
    P->generation++;
    P->cv_work.notify_all();
    P->cv_done.wait(lk, [&] { return P->remaining == 0; });
  }
  for (unsigned w = 0; w < nw; w++) pthread_barrier_destroy(&eb->waves[w].bar);
  delete eb;
}
