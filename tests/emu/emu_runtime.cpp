// TEST INFRASTRUCTURE ONLY -- see hip/hip_runtime.h in this directory.
#include <hip/hip_runtime.h>

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
thread_local EmuBlock *emu_block;
thread_local unsigned emu_phase;

void emu_launch(dim3 grid, dim3 block, const std::function<void()> &body) {
  const unsigned nthr = block.x, nw = (nthr + 63) / 64;
  EmuBlock *eb = new EmuBlock();
  for (unsigned w = 0; w < nw; w++) {
    unsigned cnt = std::min(64u, nthr - w * 64);
    pthread_barrier_init(&eb->waves[w].bar, nullptr, cnt);
  }
  for (unsigned b = 0; b < grid.x; b++) {
    std::barrier<> bar(nthr);
    eb->bar = &bar;
    std::vector<std::thread> ths;
    ths.reserve(nthr);
    for (unsigned t = 0; t < nthr; t++) {
      ths.emplace_back([=, &body]() {
        threadIdx = dim3(t); blockIdx = dim3(b); blockDim = block; gridDim = grid;
        emu_block = eb;
        emu_phase = 0;
        body();
        eb->bar->arrive_and_drop();
      });
    }
    for (auto &th : ths) th.join();
  }
  for (unsigned w = 0; w < nw; w++) pthread_barrier_destroy(&eb->waves[w].bar);
  delete eb;
}
