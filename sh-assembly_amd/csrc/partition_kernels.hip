// Key-word partition by quotient region (MSD radix, one digit per level) and the
// device-wide exclusive scan the pipeline uses.
//
// The reference has no such step: it takes a region lock per k-mer and shifts slots in
// place (cqf/gqf.c:174-264, 1614-1915). Here every key of a batch is first routed to the
// 2048-quotient region that owns it, so that one workgroup can rebuild that region in LDS
// (cqf_kernels.hip). Order inside a region is irrelevant: the filter's bytes depend only
// on the key multiset.
#include "shk_device.h"

#define SHK_RP_TILE 4096     // keys per window
#define SHK_RP_MAXP 1024     // at most 10 digit bits per level
#define SHK_SCAN_TILE 2048   // elements per workgroup in the scan kernels

struct ShkRpLevel {
  uint32_t shift;     // digit = (region >> shift) & (P-1)
  uint32_t bits;      // P = 1 << bits
  uint32_t nbuckets;  // buckets entering this level (= product of earlier P's)
  uint32_t hb;
  uint64_t q_lo;      // first quotient this context owns (multi-GPU shards)
  uint64_t nslots;    // quotients this context owns
  uint32_t out32;     // 1 (last level): write 32-bit records (quotient in region << 8 | remainder) << SHK_CHUNK_BITS | chunk
  uint32_t ablate;    // diagnostics only (SHK_RP_ABLATE): 1 = no reservation atomics, all windows write the same few KB per digit (results invalid)
  uint32_t slot_cap;  // last level only, 0 = off: bucket (region) i owns the fixed slot [i * slot_cap, (i + 1) * slot_cap) of the
                      // output instead of an exact range from a histogram pass + scan (the keys are hash values: a
                      // region's share of a batch is its mean +- a few sigma); `cursor` starts at the slots' first
                      // positions and ends as the regions' END positions. A region that gets more raises SHK_E_SLOT_FULL
  uint32_t ng_log2;   // first level only: every digit's bucket is laid out as 2^ng_log2 sub-buckets, one per window group
                      // (window index mod 2^ng_log2), each with its own cursor, so that the windows of a batch do not all
                      // reserve from the same P addresses. Measured on 832 M keys: the scatter itself is unchanged (its
                      // first level stays at 4.7 ms against 3.2 for the second: not the atomics), the hash kernel's
                      // histogram flush gets 5 % off that kernel
};

__device__ __forceinline__ uint32_t shk_word_region(uint64_t w, uint32_t hb, uint64_t q_lo) {
  uint64_t key = hb >= 64 ? w : (w & ((1ULL << hb) - 1));
  return (uint32_t)(((key >> 8) - q_lo) >> SHK_REGION_LOG2);
}

// ---------------------------------------------------------------- exclusive scan (3 kernels)
template <typename T>
__global__ void k_scan_reduce(const T *in, uint64_t n_host, const uint64_t *n_dev, uint64_t *block_sums) {
  __shared__ uint64_t scratch[SHK_MAX_WAVES + 1];
  const uint64_t n = n_dev ? *n_dev : n_host;
  const uint64_t base = (uint64_t)blockIdx.x * SHK_SCAN_TILE;
  uint64_t s = 0;
  for (uint64_t i = base + threadIdx.x; i < base + SHK_SCAN_TILE && i < n; i += blockDim.x) s += in[i];
  uint64_t tot = shk_block_sum64(s, scratch);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}
// single workgroup: exclusive scan of the block sums in place; total to *total
__global__ void k_scan_top(uint64_t *block_sums, uint64_t nblocks, uint64_t *total) {
  __shared__ uint64_t scratch[SHK_MAX_WAVES + 1];
  __shared__ uint64_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint64_t b = 0; b < nblocks; b += blockDim.x) {
    uint64_t i = b + threadIdx.x;
    uint64_t v = i < nblocks ? block_sums[i] : 0, tot;
    uint64_t ex = shk_block_exscan64(v, &tot, scratch);
    uint64_t carry = carry_s;
    if (i < nblocks) block_sums[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry_s;
}
// out[i] = exclusive prefix; out[n] = total (written by the workgroup that owns index n)
template <typename T>
__global__ void k_scan_apply(const T *in, uint64_t n_host, const uint64_t *n_dev, const uint64_t *block_sums,
                             const uint64_t *total, uint64_t *out) {
  __shared__ uint64_t scratch[SHK_MAX_WAVES + 1];
  const uint64_t n = n_dev ? *n_dev : n_host;
  const uint64_t base = (uint64_t)blockIdx.x * SHK_SCAN_TILE;
  if (base > n) return;
  const unsigned per = SHK_SCAN_TILE / blockDim.x;  // blockDim.x divides the tile
  const uint64_t my = base + (uint64_t)threadIdx.x * per;
  uint64_t s = 0;
  for (unsigned j = 0; j < per; j++)
    if (my + j < n) s += in[my + j];
  uint64_t tot;
  uint64_t ex = shk_block_exscan64(s, &tot, scratch) + block_sums[blockIdx.x];
  for (unsigned j = 0; j < per; j++) {
    if (my + j < n) { out[my + j] = ex; ex += in[my + j]; }
  }
  if (threadIdx.x == 0 && n >= base && n < base + SHK_SCAN_TILE) out[n] = *total;
}

// ---------------------------------------------------------------- partition
// bucket_base for level 1 = {0, n}
__global__ void k_rp_base1(const uint64_t *n_p, uint64_t *bucket_base) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { bucket_base[0] = 0; bucket_base[1] = *n_p; }
}

// first bucket that has a key in window w (windows are SHK_RP_TILE keys of the input)
__global__ void k_rp_tile_first(const uint64_t *bucket_base, uint32_t nbuckets, const uint64_t *n_p, uint32_t *tfb) {
  const uint64_t n = *n_p;
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t pos = w * SHK_RP_TILE;
  if (pos >= n) return;
  uint32_t lo = 0, hi = nbuckets;  // last b with bucket_base[b] <= pos
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) / 2;
    if (bucket_base[mid] <= pos) lo = mid; else hi = mid;
  }
  tfb[w] = lo;
}

// Histogram of the next digit. One workgroup takes `wtiles` consecutive windows, so that the global
// counters of a (bucket, digit) see one atomic per workgroup and bucket, not one per window
// (at the first level every window would hit the same P addresses).
__global__ void k_rp_hist(const uint64_t *words, const uint64_t *n_p, const uint64_t *bucket_base,
                          const uint32_t *tfb, ShkRpLevel lv, uint64_t *hist, uint32_t wtiles) {
  __shared__ uint32_t lh[SHK_RP_MAXP];
  const uint64_t n = *n_p;
  const uint64_t wstart = (uint64_t)blockIdx.x * wtiles * SHK_RP_TILE;
  if (wstart >= n) return;
  const uint64_t wend = wstart + (uint64_t)wtiles * SHK_RP_TILE < n ? wstart + (uint64_t)wtiles * SHK_RP_TILE : n;
  const uint32_t P = 1u << lv.bits;
  for (uint32_t b = tfb[(uint64_t)blockIdx.x * wtiles]; b < lv.nbuckets && bucket_base[b] < wend; b++) {
    const uint64_t lo = bucket_base[b] > wstart ? bucket_base[b] : wstart;
    const uint64_t hi = bucket_base[b + 1] < wend ? bucket_base[b + 1] : wend;
    if (hi <= lo) continue;
    for (uint32_t d = threadIdx.x; d < (P << lv.ng_log2); d += blockDim.x) lh[d] = 0;
    __syncthreads();
    // four loads in flight per thread
    for (uint64_t i0 = lo; i0 < hi; i0 += 4ull * blockDim.x) {
      uint64_t w[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint64_t i = i0 + (uint64_t)u * blockDim.x + threadIdx.x;
        w[u] = i < hi ? words[i] : 0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint64_t i = i0 + (uint64_t)u * blockDim.x + threadIdx.x;
        if (i < hi) {
          const uint32_t dg = (shk_word_region(w[u], lv.hb, lv.q_lo) >> lv.shift) & (P - 1);
          const uint32_t grp = (uint32_t)(i >> SHK_RP_TILE0_LOG2) & ((1u << lv.ng_log2) - 1);     // the (first-level) scatter window this key lies in
          atomicAdd(&lh[(dg << lv.ng_log2) | grp], 1u);
        }
      }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < (P << lv.ng_log2); d += blockDim.x)
      if (lh[d]) atomicAdd((unsigned long long *)&hist[(((uint64_t)b * P) << lv.ng_log2) + d], (unsigned long long)lh[d]);
    __syncthreads();
  }
}

// bucket bases of the next level from the scanned sub-bucket bases of a grouped first level: base[d] = sub[d << ng_log2]
__global__ void k_rp_group_bases(const uint64_t *sub, uint32_t nd, uint32_t ng_log2, uint64_t *base) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d <= nd) base[d] = sub[(uint64_t)d << ng_log2];
}

// cursor[b*P+d] starts at the scanned base of (b,d) and is advanced by reservations.
// One window of SHK_RP_TILE keys per workgroup of SHK_RP_THREADS threads: the keys stay in
// registers; ONE LDS atomic per key yields both the digit count and the key's rank inside
// its digit (the order of keys inside a bucket is free: the rebuild folds them into a hash);
// after the scan the keys are staged digit by digit in LDS and leave as contiguous runs.
#define SHK_RP_THREADS 512
#define SHK_RP_KPT (SHK_RP_TILE / SHK_RP_THREADS)
// (TILE_LOG2, THREADS) = (12, 512) for all levels but the first of a context's own partition: (SHK_RP_TILE0_LOG2, 1024)
// Level 0 of a partition that starts from unsorted words (the sharded flow's received words): the first level's digit
// counts per window group as k_rp_hist gives them and, from the same pass over the keys, the SECOND level's counts
// (hist1[d0 * P1 + d1], the layout k_rp_hist would fill for level 1 with its buckets = the first level's digits) -- the
// second level then needs no counting pass of its own (1.7-2.0 ms per 832 M keys). For bits0 + bits1 <= LB2.
template <int LB2>
__global__ void k_rp_hist2(const uint64_t *words, const uint64_t *n_p, ShkRpLevel lv0, ShkRpLevel lv1, uint64_t *hist0, uint64_t *hist1,
                           uint32_t wtiles) {
  __shared__ uint32_t lh0[SHK_RP_MAXP];
  __shared__ uint32_t lh1[1u << LB2];
  const uint64_t n = *n_p;
  const uint64_t wstart = (uint64_t)blockIdx.x * wtiles * SHK_RP_TILE;
  if (wstart >= n) return;
  const uint64_t wend = wstart + (uint64_t)wtiles * SHK_RP_TILE < n ? wstart + (uint64_t)wtiles * SHK_RP_TILE : n;
  const uint32_t P0 = 1u << lv0.bits, C = 1u << (lv0.bits + lv1.bits);
  for (uint32_t d = threadIdx.x; d < (P0 << lv0.ng_log2); d += blockDim.x) lh0[d] = 0;
  for (uint32_t d = threadIdx.x; d < C; d += blockDim.x) lh1[d] = 0;
  __syncthreads();
  for (uint64_t i0 = wstart; i0 < wend; i0 += 4ull * blockDim.x) {
    uint64_t w[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint64_t i = i0 + (uint64_t)u * blockDim.x + threadIdx.x;
      w[u] = i < wend ? words[i] : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint64_t i = i0 + (uint64_t)u * blockDim.x + threadIdx.x;
      if (i < wend) {
        const uint32_t region = shk_word_region(w[u], lv0.hb, lv0.q_lo);
        const uint32_t d0 = (region >> lv0.shift) & (P0 - 1);
        const uint32_t grp = (uint32_t)(i >> SHK_RP_TILE0_LOG2) & ((1u << lv0.ng_log2) - 1);
        atomicAdd(&lh0[(d0 << lv0.ng_log2) | grp], 1u);
        atomicAdd(&lh1[(region >> lv1.shift) & (C - 1)], 1u);
      }
    }
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < (P0 << lv0.ng_log2); d += blockDim.x)
    if (lh0[d]) atomicAdd((unsigned long long *)&hist0[d], (unsigned long long)lh0[d]);
  for (uint32_t d = threadIdx.x; d < C; d += blockDim.x)
    if (lh1[d]) atomicAdd((unsigned long long *)&hist1[d], (unsigned long long)lh1[d]);
}

// fixed-capacity region slots: cursor[i] = i * cap
__global__ void k_rp_slot_cursors(uint64_t *cursor, uint64_t n, uint32_t cap) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) cursor[i] = i * cap;
}

template <int TILE_LOG2, int THREADS, int PMAX = SHK_RP_MAXP>
__global__ void __launch_bounds__(THREADS) k_rp_scatter(const uint64_t *in, uint64_t *out, const uint64_t *n_p,
                                                        const uint64_t *bucket_base, const uint32_t *tfb, ShkRpLevel lv,
                                                        uint64_t *cursor, uint32_t *err) {
  constexpr uint32_t SHK_RP_TILE_ = 1u << TILE_LOG2;
  constexpr int KPT_ = (int)(SHK_RP_TILE_ / THREADS);   // keys per thread (registers)
  static_assert(SHK_RP_TILE_ <= 65536, "a rank inside a digit takes 16 bits");
  __shared__ uint32_t lh[PMAX];      // digit counts (= next rank while counting); PMAX >= 2^lv.bits (the host's choice)
  __shared__ uint32_t lbase[PMAX];   // local exclusive base of each digit
  __shared__ uint64_t gbase[PMAX];   // reserved global base of each digit
  __shared__ uint64_t stage[SHK_RP_TILE_];
  __shared__ uint32_t scratch[SHK_MAX_WAVES + 1];
  const uint64_t n = *n_p;
  const uint64_t wstart = (uint64_t)blockIdx.x * SHK_RP_TILE_;
  if (wstart >= n) return;
  const uint64_t wend = wstart + SHK_RP_TILE_ < n ? wstart + SHK_RP_TILE_ : n;
  const uint32_t P = 1u << lv.bits;
  // (tfb is indexed by 4096-key windows)
  for (uint32_t b = (lv.nbuckets > 1 ? tfb[(uint64_t)blockIdx.x << (TILE_LOG2 - 12)] : 0u); b < lv.nbuckets && bucket_base[b] < wend; b++) {
    const uint64_t lo = bucket_base[b] > wstart ? bucket_base[b] : wstart;
    const uint64_t hi = bucket_base[b + 1] < wend ? bucket_base[b + 1] : wend;
    if (hi <= lo) continue;
    const uint32_t cnt = (uint32_t)(hi - lo);
    for (uint32_t d = threadIdx.x; d < P; d += THREADS) lh[d] = 0;
    uint64_t w[KPT_];
#pragma unroll
    for (int u = 0; u < KPT_; u++) {
      const uint32_t i = threadIdx.x + (uint32_t)u * THREADS;
      w[u] = i < cnt ? in[lo + i] : 0;
    }
    __syncthreads();
    uint32_t dr[KPT_];               // digit << 16 | rank inside the digit (rank < SHK_RP_TILE)
#pragma unroll
    for (int u = 0; u < KPT_; u++) {
      const uint32_t i = threadIdx.x + (uint32_t)u * THREADS;
      dr[u] = 0;
      if (i < cnt) {
        const uint32_t d = (shk_word_region(w[u], lv.hb, lv.q_lo) >> lv.shift) & (P - 1);
        dr[u] = (d << 16) | atomicAdd(&lh[d], 1u);
      }
    }
    __syncthreads();
    // exclusive scan of the digit counts (P <= 1024)
    uint32_t carry = 0;
    for (uint32_t d0 = 0; d0 < P; d0 += THREADS) {
      uint32_t d = d0 + threadIdx.x;
      uint32_t v = d < P ? lh[d] : 0, tot;
      uint32_t ex = shk_block_exscan(v, &tot, scratch);
      if (d < P) {
        lbase[d] = carry + ex;
        if (lv.ablate & 1) gbase[d] = cursor[((uint64_t)b * P + d) << lv.ng_log2] + (uint64_t)(blockIdx.x % 1024) * 24;
        else gbase[d] = v ? atomicAdd((unsigned long long *)&cursor[((((uint64_t)b * P) + d) << lv.ng_log2) | (blockIdx.x & ((1u << lv.ng_log2) - 1))], (unsigned long long)v) : 0;
        if (lv.slot_cap && v && gbase[d] + v > ((uint64_t)b * P + d + 1) * lv.slot_cap) {
          atomicOr(err, SHK_E_SLOT_FULL);
          gbase[d] = ~0ULL;              // (nothing of this run is written)
        }
      }
      carry += tot;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < KPT_; u++) {
      const uint32_t i = threadIdx.x + (uint32_t)u * THREADS;
      if (i < cnt) stage[lbase[dr[u] >> 16] + (dr[u] & 0xFFFFu)] = w[u];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cnt; i += THREADS) {
      uint64_t x = stage[i];
      uint32_t d = (shk_word_region(x, lv.hb, lv.q_lo) >> lv.shift) & (P - 1);
      if (lv.out32) {
        // behind the last level a word's region is its bucket: the rebuild kernel only needs the
        // quotient inside the region, the remainder and the chunk -- half the bytes, no 64-bit arithmetic
        const uint64_t key = lv.hb >= 64 ? x : (x & ((1ULL << lv.hb) - 1));
        const uint64_t q = (key >> 8) - lv.q_lo;
        if (q >= lv.nslots) atomicOr(err, SHK_E_CORRUPT);
        const uint32_t rec = ((((uint32_t)q & (SHK_REGION - 1)) << 8 | (uint32_t)(key & 0xff)) << SHK_CHUNK_BITS) |
                             ((uint32_t)(x >> lv.hb) & (SHK_MAX_CHUNKS - 1));
        if (gbase[d] != ~0ULL) reinterpret_cast<uint32_t *>(out)[gbase[d] + (i - lbase[d])] = rec;
      } else {
        out[gbase[d] + (i - lbase[d])] = x;
      }
    }
    shk_lds_barrier();             // (LDS only: the runs just written drain to HBM behind it)
  }
}
