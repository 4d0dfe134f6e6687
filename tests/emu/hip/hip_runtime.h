// TEST INFRASTRUCTURE ONLY -- a tiny CPU stand-in for <hip/hip_runtime.h>.
//
// tests/emu/Makefile compiles the UNMODIFIED kernel sources of sh-assembly_amd/csrc with
// g++ against this header into tests/emu/libshk_emu.so, so that the kernels' logic
// (indexing, barriers, wave collectives, atomics) can be exercised against the oracle in
// the CPU test suite, under valgrind/ASan if wanted, before spending GPU time. It is never
// loaded by the product path: sh-assembly_amd/ only ever opens libshk.so built by hipcc.
//
// Model: workgroups run one after another; every thread of a workgroup is a FIBER (ucontext) of
// the calling OS thread, scheduled round robin; a fiber runs until it has to wait at a barrier
// (__syncthreads, or the per-wave barrier inside a wave collective) and then hands the CPU to
// the next one -- no OS threads, no futexes: the suite spends its time in the kernels' code.
// A wave is 64 consecutive threads that exchange values through a per-wave mailbox. Threads
// that have left the kernel no longer take part in barriers (waves may exit early on the GPU
// too); a collective that not all live lanes of a wave reach is a deadlock here as it would be
// undefined there: the scheduler aborts with a message when every live fiber is waiting.
// `__shared__` becomes `static` (one workgroup is alive at a time).
#pragma once
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 v = {x, y, z, w}; return v; }
struct uint2 { uint32_t x, y; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 v = {x, y}; return v; }

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
typedef int hipStream_t;
struct EmuEvent { double t; };
typedef EmuEvent *hipEvent_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
enum { hipHostMallocDefault = 0 };

// a barrier among fibers: the last live member to arrive opens the next generation; waiting = yielding
struct EmuBarrier { unsigned live, arrived, generation; };
void emu_barrier_wait(EmuBarrier *b);
// two mailboxes used alternately: one barrier per collective is enough (a lane can only overwrite a box two
// collectives later, i.e. after a barrier every reader of that box has already passed)
struct EmuWave { EmuBarrier bar; uint64_t box[2][64]; };
extern thread_local unsigned emu_phase;
struct EmuBlock { EmuBarrier bar; EmuWave waves[16]; };
extern thread_local EmuBlock *emu_block;

static inline void __syncthreads() { emu_barrier_wait(&emu_block->bar); }
static inline EmuWave *emu_wave() { return &emu_block->waves[threadIdx.x / 64]; }
static inline void emu_wave_barrier() { emu_barrier_wait(&emu_wave()->bar); }
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier()
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_s_memtime() 0ULL
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

template <typename T> static inline T emu_xchg(T v, int src_lane, bool valid) {
  static_assert(sizeof(T) <= 8, "shuffle of <= 8 bytes");
  EmuWave *w = emu_wave();
  uint64_t raw = 0;
  memcpy(&raw, &v, sizeof(T));
  const unsigned ph = emu_phase++ & 1;
  w->box[ph][threadIdx.x & 63] = raw;
  emu_barrier_wait(&w->bar);
  uint64_t got = valid ? w->box[ph][src_lane & 63] : raw;
  T out;
  memcpy(&out, &got, sizeof(T));
  return out;
}
template <typename T> static inline T __shfl(T v, int lane) { return emu_xchg(v, lane, true); }
template <typename T> static inline T __shfl_up(T v, unsigned d) {
  int l = (int)(threadIdx.x & 63) - (int)d;
  return emu_xchg(v, l, l >= 0);
}
template <typename T> static inline T __shfl_down(T v, unsigned d) {
  int l = (int)(threadIdx.x & 63) + (int)d;
  return emu_xchg(v, l, l < 64);
}
template <typename T> static inline T __shfl_xor(T v, int m) { return emu_xchg(v, (int)(threadIdx.x & 63) ^ m, true); }
// DPP move (the modes the kernels use: row_shr:n, row_shl:n, row_bcast15, row_bcast31)
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const int lane = threadIdx.x & 63, row = lane >> 4, within = lane & 15;
  int src_lane = -1;
  if (ctrl >= 0x111 && ctrl <= 0x11F) { int w = within - (ctrl - 0x110); if (w >= 0) src_lane = row * 16 + w; }
  else if (ctrl >= 0x101 && ctrl <= 0x10F) { int w = within + (ctrl - 0x100); if (w < 16) src_lane = row * 16 + w; }
  else if (ctrl == 0x142) { if (row >= 1) src_lane = row * 16 - 1; }
  else if (ctrl == 0x143) { if (row >= 2) src_lane = 31; }
  const int got = emu_xchg(src, src_lane < 0 ? 0 : src_lane, true);
  const bool enabled = ((row_mask >> row) & 1) && ((bank_mask >> (within >> 2)) & 1);
  if (!enabled) return old;
  if (src_lane < 0) return bound_ctrl ? 0 : old;
  return got;
}
static inline int __builtin_amdgcn_readlane(int v, int lane) { return emu_xchg(v, lane, true); }
static inline unsigned long long __ballot(int pred) {
  EmuWave *w = emu_wave();
  const unsigned ph = emu_phase++ & 1;
  w->box[ph][threadIdx.x & 63] = pred ? 1 : 0;
  emu_barrier_wait(&w->bar);
  unsigned long long m = 0;
  for (int i = 0; i < 64; i++) m |= (unsigned long long)(w->box[ph][i] & 1) << i;
  return m;
}
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

template <typename T, typename U> static inline T atomicAdd(T *p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_SEQ_CST); }
template <typename T, typename U> static inline T atomicSub(T *p, U v) { return __atomic_fetch_sub(p, (T)v, __ATOMIC_SEQ_CST); }
template <typename T, typename U> static inline T atomicOr(T *p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_SEQ_CST); }
template <typename T, typename U> static inline T atomicAnd(T *p, U v) { return __atomic_fetch_and(p, (T)v, __ATOMIC_SEQ_CST); }
template <typename T, typename U> static inline T atomicExch(T *p, U v) { return __atomic_exchange_n(p, (T)v, __ATOMIC_SEQ_CST); }
template <typename T, typename U> static inline T atomicMin(T *p, U v) {
  T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while ((T)v < old && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return old;
}
template <typename T, typename U> static inline T atomicMax(T *p, U v) {
  T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while ((T)v > old && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return old;
}
template <typename T, typename U, typename V> static inline T atomicCAS(T *p, U cmp, V val) {
  T expected = (T)cmp;
  __atomic_compare_exchange_n(p, &expected, (T)val, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return expected;
}

// ---------------------------------------------------------------- host API
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = 0; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); if (*p) memset(*p, 0xCD, n); return *p ? 0 : 2; }
static inline hipError_t hipFree(void *p) { free(p); return 0; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = calloc(n ? n : 1, 1); return *p ? 0 : 2; }
static inline hipError_t hipHostFree(void *p) { free(p); return 0; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return 0; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return 0; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline double emu_now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new EmuEvent(); return 0; }
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = 1; return 0; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new EmuEvent(); return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = emu_now(); return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return 0; }

void emu_launch(dim3 grid, dim3 block, const std::function<void()> &body);
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
  emu_launch((grid), (block), [=]() { kern(__VA_ARGS__); })
