// Counting-quotient-filter rebuild kernels: the GPU form of qf_insert_advance
// (cqf/gqf.c:2432-2440 -> insert1_advance :1614-1915, insert_advance :2024-2136) and of
// the deNoise sweep (qf_clean_singleton :2792-2876, driven by CQF_mt.h:884-901,999-1039).
//
// The table in HBM is the reference's own byte layout (89-byte packed qfblocks,
// gqf.c:63-86), so export is a plain copy. One WAVE owns one region of 256 quotients
// (4 blocks; the kernels are launched with 64-thread workgroups, so every barrier is a
// wave-local no-op). It stages the region's old bytes in LDS, folds the batch's keys
// for that region into an LDS hash, merges old runs and new keys per quotient, and
// re-encodes the runs at their canonical positions (run of q starts at max(q, end of
// previous run + 1); remainders ascending; counters per encode_counter :1225-1255).
// Because those positions chain across regions, the work is split in three launches:
//   k_region_merge<false>  per region: (T = slots its runs need, c = where they end if
//                          nothing spills in), plus the new-key statistics
//   k_region_scan          free pointer at every region start: f' = max(f + T, c)
//   k_region_merge<true>   per region: lay the runs out from its free pointer, write table B
// Tables A (read) and B (written) ping-pong; B is zeroed before the write pass.
#include "shk_device.h"

#define SHK_EMPTY 0xFFFFFFFFu
#define SHK_SUM_STRIDE 8
#define SHK_SPILL_LENS SHK_REGION                                   // one length byte per quotient
#define SHK_SPILL_STRIDE 768                                         // + the lanes' staged run bytes, packed: a filled region needs
                                                                     // about 250 of the 512; one that needs more than the record holds is rebuilt from the list
#define SHK_NC_CAP 128                                               // entries of a region's first-chunk record; a region with more new keys
                                                                     // in one pass adds the rest to the chunk histogram directly
#define SHK_SPILL_PACK_MAX (SHK_SPILL_STRIDE - SHK_SPILL_LENS)
#define SHK_RSCAN_TILE 4096

struct ShkMergeArgs {
  const uint8_t *tabA;
  uint8_t *tabB;
  const uint64_t *finA;         // [nregions+1] free pointer at each region start of A
  uint64_t *finB;               // same for B (write pass input; written by the single-launch rebuild)
  const uint32_t *words;        // 32-bit records sorted by region (written by the last partition level); null when there are none
  const uint64_t *region_base;  // [nregions+1] offsets into words; with region_cap: [nregions] END offsets, region r starts at r * region_cap
  uint32_t region_cap;          // 0, or the fixed capacity of a region's slot in `words` (ShkRpLevel::slot_cap)
  uint64_t nslots, xnslots, nblocks;
  uint64_t q_lo;
  uint32_t hb;
  uint32_t chunk_lo, chunk_hi;    // only words whose chunk index lies in [lo, hi] take part
  uint32_t hist_base, hist_shift; // coarse histogram of the first chunk of every NEW key
  int want_hist;                  // 0: totals only (the common case: no deNoise point inside the batch)
  int denoise;                    // 1: drop entries whose count is exactly 1 (no new keys)
  uint32_t *summary;              // [SHK_SUM_STRIDE*nregions]: T, c (relative to the region start; 0 = empty),
                                  // new distinct, occurrences added, removed, new before hist_base
  unsigned long long *counters;   // 0 new distinct, 1 occurrences added, 2 removed, 3 new before hist_base
  unsigned long long *hist;       // [SHK_HIST_BINS]
  uint32_t *err;
  unsigned long long *lb_agg;     // [nregions] look-back records of the single-launch rebuild (zeroed per launch)
  unsigned long long *lb_incl;    // [nregions]
  unsigned long long *dbg;        // diagnostics only: per-phase cycle sums of sampled regions (null = off)
  uint32_t ablate;                // diagnostics only (SHK_ABLATE): skip phases to time them; results invalid
  // spill scheme (MODE 3 -> k_region_scan_* -> k_region_place): the summary launch keeps every region's
  // run lengths and encoded bytes so that the second launch only places them
  uint8_t *spill;                 // [nregions * SHK_SPILL_STRIDE]
  uint32_t *over_list;            // regions whose runs did not fit the spill record (rebuilt by MODE 1 from this list)
  unsigned long long *n_over;
  const uint32_t *list;           // regions to rebuild (null = all: region = blockIdx.x + r0)
  uint16_t *newchunks;            // [nregions * SHK_NC_CAP] first chunk of every NEW key of the region (null = off; needs want_hist)
  unsigned long long *chist;      // [SHK_MAX_CHUNKS] the chunk histogram k_chunk_hist sums the records into (zeroed BEFORE the pass: overflow adds here)
  // one-pass deNoise point (FUSED instantiation): words of chunks <= split are inserted BEFORE the round, the others after
  uint32_t split;
  uint32_t *isum;                 // [2 * nregions] (T, c) of the INTERMEDIATE table (after the chunks <= split, before the round)
  uint8_t *ilens;                 // [256 * nregions] its run length per quotient
  const uint64_t *prot_list;      // sorted quotients whose singleton the round's range walk protects (null on the first go)
  uint32_t nprot;
  uint32_t r0;                    // first region of this launch (a pass over more than 2^24 regions takes several launches:
                                  // HIP limits a grid to fewer than 2^32 threads)
  uint32_t rstride;               // region = (blockIdx.x + r0) * rstride: > 1 for the statistics pass over a sample of the regions
  int counted;                    // 1: the records' chunk field holds (multiplicity - 1) of a counted insert (insert_advance with
                                  // count > 1, gqf.c:2024-2136); every record takes part, no chunk statistics
};

__device__ __forceinline__ unsigned shk_img_slot_off(unsigned p) {
  return (p >> 6) * SHK_BLOCK_BYTES + SHK_OFF_SLOTS + (p & 63);
}

// decode_counter (gqf.c:1259-1299) on a block image; `pos` = first slot of the entry,
// `run_end` = last slot of its run. Returns the number of slots.
__device__ __forceinline__ unsigned shk_img_dec(const uint8_t *img, unsigned pos, unsigned run_end, unsigned *rem_out,
                                                uint64_t *count) {
  unsigned rem = img[shk_img_slot_off(pos)];
  *rem_out = rem;
  if (pos == run_end) { *count = 1; return 1; }
  unsigned digit = img[shk_img_slot_off(pos + 1)];
  if (digit > rem) { *count = 1; return 1; }
  unsigned n = 1;
  uint64_t cnt = 0;
  if (digit == 0) { n++; digit = img[shk_img_slot_off(pos + n)]; }
  while ((digit & 0x80) && pos + n < run_end) { cnt = cnt * 128 + (digit & 0x7f); n++; digit = img[shk_img_slot_off(pos + n)]; }
  cnt = cnt * 128 + (digit & 0x7f);
  *count = cnt + 1;
  return n + 1;
}

// workgroup exclusive scan of free-pointer functions (thread order = quotient order)
__device__ __forceinline__ ShkMP shk_block_exscan_mp(ShkMP v, ShkMP *total, long long *sa, long long *sb) {
  const unsigned lane = shk_lane(), wave = shk_wave(), nw = blockDim.x / SHK_WAVE;
  ShkMP incl = v;
  for (int d = 1; d < SHK_WAVE; d <<= 1) {
    ShkMP y;
    y.a = __shfl_up(incl.a, d);
    y.b = __shfl_up(incl.b, d);
    if (lane >= (unsigned)d) incl = shk_mp_compose(y, incl);
  }
  if (lane == SHK_WAVE - 1) { sa[wave] = incl.a; sb[wave] = incl.b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ShkMP run; run.a = 0; run.b = SHK_NEG_INF;
    for (unsigned w = 0; w < nw; w++) {
      ShkMP cur; cur.a = sa[w]; cur.b = sb[w];
      sa[w] = run.a; sb[w] = run.b;
      run = shk_mp_compose(run, cur);
    }
    sa[SHK_MAX_WAVES] = run.a; sb[SHK_MAX_WAVES] = run.b;
  }
  __syncthreads();
  ShkMP pre; pre.a = sa[wave]; pre.b = sb[wave];
  // exclusive within the wave: inclusive of the previous lane
  ShkMP prev;
  prev.a = __shfl_up(incl.a, 1);
  prev.b = __shfl_up(incl.b, 1);
  if (lane == 0) { prev.a = 0; prev.b = SHK_NEG_INF; }
  ShkMP res = shk_mp_compose(pre, prev);
  total->a = sa[SHK_MAX_WAVES];
  total->b = sb[SHK_MAX_WAVES];
  __syncthreads();
  return res;
}

// ---- one merged entry of a run: fast paths for the common small counts
__device__ __forceinline__ unsigned shk_enc_len_fast(unsigned rem, uint64_t total) {
  if (total <= 1) return (unsigned)total;
  if (total <= 128) return 2u + ((unsigned)(total - 1) > rem ? 1u : 0u);
  return shk_enc_len(rem, total);
}
// decode_counter (gqf.c:1259-1299) on the block image with a fast path for counts <= 128
__device__ __forceinline__ unsigned shk_img_dec_fast(const uint8_t *img, unsigned pos, unsigned run_end, unsigned *rem_out,
                                                     uint64_t *count) {
  const unsigned rem = img[shk_img_slot_off(pos)];
  *rem_out = rem;
  if (pos == run_end) { *count = 1; return 1; }
  const unsigned d = img[shk_img_slot_off(pos + 1)];
  if (d > rem) { *count = 1; return 1; }
  if (d != 0 && d < 0x80) { *count = (uint64_t)d + 1; return 2; }
  return shk_img_dec(img, pos, run_end, rem_out, count);
}

// ---- image -> table B (shared by the rebuild kernel's write modes and k_region_place)
template <int IMGB>
__device__ __forceinline__ void shk_store_image(const ShkMergeArgs &A, uint32_t r, uint32_t nregions, const uint8_t *nimg,
                                                unsigned tid, uint32_t nown, uint64_t b0, uint64_t q0, uint32_t out_lo,
                                                uint32_t out_hi, bool new_any, long long fout_rel) {
  // blocks past the last quotient hold only spilled runs: their offset bytes come from the
  // final free pointer and are written by the last region's wave
  if (r == nregions - 1) {
    const long long fend = fout_rel + (long long)q0;
    for (uint64_t b = A.nslots / 64 + tid; b < A.nblocks; b += SHK_WAVE) {
      long long o = fend - (long long)(64 * b);
      A.tabB[b * SHK_BLOCK_BYTES] = (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o));
    }
  }

  // ---- image -> table B. Own blocks: offset byte + occupieds. Slots and runends bytes of
  // [out_lo, out_hi) only; the first and last runends byte may be shared with the
  // neighbouring regions' runs, so they are OR-ed in atomically (B was zeroed).
  uint8_t *tb = A.tabB + b0 * SHK_BLOCK_BYTES;
  // When no earlier region spills into this one, the own blocks are written whole with
  // dword stores (a full region's own blocks are 356 contiguous, 4-byte aligned bytes).
  uint32_t vblk = nown;
  if (new_any && out_lo == 0 && (nown * SHK_BLOCK_BYTES) % 4 == 0) {
    vblk = 0;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(nimg);
    uint32_t *dst = reinterpret_cast<uint32_t *>(tb);
    for (uint32_t i = tid; i < nown * SHK_BLOCK_BYTES / 4; i += SHK_WAVE) dst[i] = src[i];
  }
  // the blocks before vblk: offset byte + occupieds always, slots/runends only where owned
  for (uint32_t i = tid; i < vblk * 9; i += SHK_WAVE) {
    const uint32_t blk = i / 9, byte = i % 9;
    tb[blk * SHK_BLOCK_BYTES + byte] = nimg[blk * SHK_BLOCK_BYTES + byte];
  }
  if (out_hi > out_lo) {
    // slots [out_lo, out_hi) minus what the dword stores covered ([vblk*64, nown*64))
    const uint32_t va = vblk * 64, vb = nown * 64;
    for (uint32_t p = out_lo + tid; p < out_hi; p += SHK_WAVE)
      if (p < va || p >= vb || vblk == nown) tb[shk_img_slot_off(p)] = nimg[shk_img_slot_off(p)];
    const uint32_t m0 = out_lo >> 3, m1 = (out_hi - 1) >> 3;
    for (uint32_t m = m0 + tid; m <= m1; m += SHK_WAVE) {
      const uint32_t bo = (m >> 3) * SHK_BLOCK_BYTES + SHK_OFF_RUN + (m & 7);
      const uint8_t v = nimg[bo];
      if (vblk < nown && (m >> 3) >= vblk && (m >> 3) < nown) continue;  // written by the dword stores
      if (m == m0 || m == m1) {
        if (v) {
          uint8_t *addr = tb + bo;
          uintptr_t ai = reinterpret_cast<uintptr_t>(addr);
          uint32_t *w = reinterpret_cast<uint32_t *>(ai & ~(uintptr_t)3);
          atomicOr(w, (uint32_t)v << ((ai & 3) << 3));
        }
      } else {
        tb[bo] = v;
      }
    }
  }
}

// MODE 0: summary (lengths + statistics). MODE 1: write pass of the two-launch scheme
// (free pointers come from k_region_scan_*). MODE 3: summary that also spills the run lengths
// and encodings for k_region_place. MODE 2: single launch -- the wave obtains its
// free pointer by looking back at the regions before it (see k_region_merge docs below).
#define SHK_STAGE_PER_LANE 32   // bytes of encoded run kept per lane between the length pass and placement
#define SHK_STAGE_STRIDE 36     // lanes 9 dwords apart: byte i of every lane's area falls into a different LDS bank
#define SHK_LB_VALID 0x80000000u
#define SHK_LB_INCL (1ULL << 63)

#define SHK_STAMP(i) do { if (A.dbg && (blockIdx.x & 63) == 0 && threadIdx.x == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&A.dbg[i], t_ - t_prev); t_prev = t_; } } while (0)

template <int MODE, int IMGB, bool FUSED = false>
__global__ void __launch_bounds__(SHK_MERGE_GROUP) k_region_merge(ShkMergeArgs A) {
  static_assert(!FUSED || MODE == 3, "the one-pass deNoise point is a spill-mode pass");
  constexpr bool WRITE = MODE == 1 || MODE == 2;   // builds the image and stores table B
  constexpr bool STAGE = WRITE || MODE == 3;       // keeps the runs' encodings per lane
  // LDS image of IMGB blocks: the region's own blocks + the blocks its runs may spill into.
  // IMGB = IMG_BLOCKS normally; the host retries a pass with IMG_BLOCKS_BIG when a cluster is longer.
  constexpr unsigned IMG_BLOCKS = IMGB, IMG_SLOTS = IMGB * 64, IMG_BYTES = IMGB * SHK_BLOCK_BYTES;
  static_assert(IMGB <= SHK_WAVE, "one lane per image block in the rank/select step");
  __shared__ __attribute__((aligned(16))) uint32_t hkey[SHK_HCAP];   // tag << 12 | first chunk ; tag = local quotient << 8 | remainder
  // occurrences in this batch; FUSED: two 16-bit counters per key, low = in the chunks <= split, high = behind it (one LDS
  // atomic; a region with 2^15 words or more in one batch leaves the one-pass point to the general path: SHK_E_FUSED).
  // Two 32-bit counters per key cost the one-pass kernel 2 KB of LDS and with them one workgroup per CU
  __shared__ __attribute__((aligned(16))) uint32_t hcnt[SHK_HCAP];
  __shared__ uint32_t s_added_b;
  // new entries per quotient, later the new run length: 16 bits each (at most SHK_HCAP keys; a run has at most 256 entries
  // of at most 11 slots), two to a word for the counting sort's atomics. LDS is what limits this kernel's workgroups per
  // CU: the allocation granule is 1280 bytes, and at nine granules (11520 bytes) 14 of them fit instead of 12
  __shared__ __attribute__((aligned(16))) uint16_t qcnt[SHK_REGION];
  uint32_t *qcntw = reinterpret_cast<uint32_t *>(qcnt);
  __shared__ uint16_t qoff[SHK_REGION + 2];
  __shared__ uint16_t nidx[SHK_HCAP];   // hash slots grouped by quotient, sorted by remainder
  __shared__ __attribute__((aligned(8))) uint16_t orend[SHK_REGION];// slot (image relative) of the j-th old runend of the region
  __shared__ uint16_t rstart[SHK_REGION];
  __shared__ __attribute__((aligned(16))) uint8_t oimg[IMG_BYTES + 16];
  __shared__ __attribute__((aligned(16))) uint8_t nimg[IMG_BYTES + 16];
  __shared__ __attribute__((aligned(16))) uint8_t stage[SHK_MERGE_THREADS * SHK_STAGE_STRIDE];
  __shared__ uint64_t oocc[SHK_REGION_BLOCKS];
  __shared__ uint64_t orunw[IMG_BLOCKS];
  __shared__ uint32_t oorank[SHK_REGION_BLOCKS + 1];
  __shared__ uint32_t orrank[IMG_BLOCKS + 1];
  __shared__ uint32_t s_fail, s_added, s_nlist;
  // one private word (pair) per lane: where the branch-free probe loop sends the lanes that have nothing to do, so that
  // their no-op atomics do not land on random banks next to the real ones (LDS bank conflicts were 18 % of this kernel's
  // cycles). The words live in `orend`, which is idle until the fold is over -- 512 bytes more would push the plain
  // kernel's LDS over an allocation step and cost it a workgroup per CU (measured: 12.6 -> 13.7 ms).
  uint32_t *hidle = reinterpret_cast<uint32_t *>(orend);
  // the coarse histogram of the merge pass lives in `orunw`, which only the rank/select step in front of the join uses
  static_assert(sizeof(uint64_t) * IMG_BLOCKS >= sizeof(uint32_t) * SHK_HIST_BINS, "lhist fits orunw");
  uint32_t *lhist = reinterpret_cast<uint32_t *>(orunw);

  unsigned long long t_prev = A.dbg ? __builtin_amdgcn_s_memtime() : 0;
  const unsigned tid = threadIdx.x;
  const unsigned ngrp = blockDim.x;              // all waves of the group: staging, init, key folding
  constexpr unsigned nthr = SHK_MERGE_THREADS;   // one wave does the rest
  const uint32_t r = A.list ? A.list[blockIdx.x] : (blockIdx.x + A.r0) * A.rstride;
  const uint32_t nregions = (uint32_t)((A.nslots + SHK_REGION - 1) / SHK_REGION);
  const uint64_t q0 = (uint64_t)r * SHK_REGION;
  const uint32_t nq = (uint32_t)((A.nslots - q0) < SHK_REGION ? (A.nslots - q0) : SHK_REGION);
  const uint32_t nown = (nq + 63) / 64;
  const uint64_t b0 = q0 / 64;

  // ---- old extent of this region in table A
  const uint64_t fa0 = A.finA[r], fa1 = A.finA[r + 1];
  const uint64_t olo_abs = fa0 > q0 ? fa0 : q0;
  const bool old_any = fa1 > olo_abs;
  const uint32_t olo = (uint32_t)(olo_abs - q0);
  const uint32_t ohi = old_any ? (uint32_t)((fa1 - q0) > 0xFFFFFFF ? 0xFFFFFFF : (fa1 - q0)) : olo;
  uint32_t nblk_old = old_any ? (ohi + 63) / 64 : 0;
  if (nblk_old < nown) nblk_old = nown;
  if (tid == 0) { s_fail = 0; s_added = 0; s_nlist = 0; s_added_b = 0; }
  // hash slots in use, in order of first insertion (lives in `stage`, which is idle until the merge pass)
  uint16_t *slist = reinterpret_cast<uint16_t *>(stage);
  bool fatal = false;
  if (nblk_old > IMG_BLOCKS || ohi > IMG_SLOTS) {
    if (tid == 0) atomicOr(A.err, SHK_E_OLD_EXTENT);
    fatal = true;
    nblk_old = nown;
  }
  if (b0 + nblk_old > A.nblocks) nblk_old = (uint32_t)(A.nblocks - b0);

  // A region without old runs and without words has nothing to say: T = 0, no statistics, all lengths 0
  // (sparse tables -- the first batches of a build, big filters -- are mostly such regions)
  if ((MODE == 0 || MODE == 3) && !old_any && !fatal && (!A.words || (A.region_cap ? A.region_base[r] == (uint64_t)r * A.region_cap : A.region_base[r] == A.region_base[r + 1]))) {
    if (tid < SHK_SUM_STRIDE) A.summary[(size_t)SHK_SUM_STRIDE * r + tid] = 0;
    if (MODE == 3 && tid < SHK_WAVE) reinterpret_cast<uint32_t *>(A.spill + (size_t)r * SHK_SPILL_STRIDE)[tid] = 0;
    if (FUSED) {
      if (tid < 2) A.isum[2 * (size_t)r + tid] = 0;
      if (tid < SHK_WAVE) reinterpret_cast<uint32_t *>(A.ilens + (size_t)r * SHK_REGION)[tid] = 0;
    }
    return;
  }

  // ---- stage the old bytes (dword copies; the region's first byte is 4-byte aligned)
  {
    const uint32_t nbytes = nblk_old * SHK_BLOCK_BYTES;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(A.tabA + b0 * SHK_BLOCK_BYTES);
    uint32_t *dst = reinterpret_cast<uint32_t *>(oimg);
    if (!(A.ablate & 64)) for (uint32_t i = tid; i < (nbytes + 3) / 4; i += ngrp) dst[i] = src[i];
    // (16-byte LDS stores: the LDS pipeline, shared by all waves of the CU, is what this kernel keeps busiest)
    {
      const uint4 e4 = make_uint4(SHK_EMPTY, SHK_EMPTY, SHK_EMPTY, SHK_EMPTY), z4 = make_uint4(0, 0, 0, 0);
      for (uint32_t i = tid; i < SHK_HCAP / 4; i += ngrp) reinterpret_cast<uint4 *>(hkey)[i] = e4;
      for (uint32_t i = tid; i < SHK_HCAP / 4; i += ngrp) reinterpret_cast<uint4 *>(hcnt)[i] = z4;
      for (uint32_t i = tid; i < SHK_REGION / 8; i += ngrp) reinterpret_cast<uint4 *>(qcnt)[i] = z4;
    }
    if (tid < 2 * SHK_WAVE) hidle[tid] = 0;
    if (WRITE) {
      uint32_t *z = reinterpret_cast<uint32_t *>(nimg);
      for (uint32_t i = tid; i < (IMG_BYTES + 16) / 4; i += ngrp) z[i] = 0;
    }
  }
  __syncthreads();

  SHK_STAMP(0);   // staging + init
  // ---- fold this region's new keys into the LDS hash
  // The kernel is bound by scalar-ALU issue (divergent branches cost s_*exec instructions for the whole
  // wave), so the probe loop is written branch-free: every lane executes every LDS operation, with
  // operands that make it a no-op for lanes that have nothing (more) to insert -- a compare value no
  // slot can hold, min with ~0, add 0. Four words per lane probe together; the loop condition is wave-uniform.
  uint32_t my_added = 0, my_added_b = 0;
  if (A.words && !fatal && !(A.ablate & 1)) {
    const uint64_t kb = A.region_cap ? (uint64_t)r * A.region_cap : A.region_base[r], ke = A.region_cap ? A.region_base[r] : A.region_base[r + 1];
    const unsigned lane = shk_lane();
    const bool wh = A.want_hist != 0;
    bool corrupt = false, hfull = false;
    const uint32_t nw = (uint32_t)(ke - kb);   // a region's share of one batch is far below 2^32 words
    if (FUSED && ((ke - kb) >= 32768 || A.counted) && tid == 0) atomicOr(A.err, SHK_E_FUSED);   // (16-bit counters, bit 31 of a word = the "new key" flag)
    const uint32_t *wp = A.words + kb;
    for (uint32_t i0 = 0; i0 < nw; i0 += 4 * ngrp) {
      uint32_t wv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t i = i0 + (uint32_t)u * ngrp + tid;
        wv[u] = i < nw ? wp[i] : ~0u;
      }
      uint32_t h[4], want[4], chk[4], wgt[4];
      bool pend[4], bef[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        // record = (quotient in region << 8 | remainder) << SHK_CHUNK_BITS | chunk (k_rp_scatter, last level)
        const uint32_t w = wv[u];
        const bool in = i0 + (uint32_t)u * ngrp + tid < nw;
        const uint32_t chunk = w & (SHK_MAX_CHUNKS - 1);
        const uint32_t tag = (w >> SHK_CHUNK_BITS) & 0xFFFFu;
        const bool inr = in && (A.counted || (chunk >= A.chunk_lo && chunk <= A.chunk_hi));
        corrupt |= inr && (tag >> 8) >= nq;
        pend[u] = inr && (tag >> 8) < nq;
        want[u] = tag << SHK_CHUNK_BITS;
        chk[u] = chunk;
        wgt[u] = A.counted ? chunk + 1u : 1u;
        bef[u] = FUSED && chunk <= A.split;
        h[u] = (__umul24(tag, 40503u) & 0xFFFFu) >> (16 - SHK_HCAP_LOG2);   // 16-bit multiplicative hash, full-rate multiply
        my_added += (pend[u] && !bef[u]) ? wgt[u] : 0u;
        if (FUSED) my_added_b += (pend[u] && bef[u]) ? wgt[u] : 0u;
      }
      if (A.ablate & 256) continue;   // diagnostics: loads + setup only
      uint32_t guard = 0;
      while (__ballot(pend[0] || pend[1] || pend[2] || pend[3])) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (!__ballot(pend[u])) continue;   // wave-uniform
          // the CAS both claims an empty slot and tells what the slot holds otherwise (no separate read);
          // 0xFFFFFFFE is never stored (tags use 28 bits, SHK_EMPTY is ~0): that compare cannot succeed
          const uint32_t prev = atomicCAS(pend[u] ? &hkey[h[u]] : &hidle[lane], pend[u] ? SHK_EMPTY : 0xFFFFFFFEu, want[u] | (SHK_MAX_CHUNKS - 1));
          const bool ins = pend[u] && prev == SHK_EMPTY;
          const uint32_t now = ins ? want[u] : prev;
          const bool match = pend[u] && (now >> SHK_CHUNK_BITS) == (want[u] >> SHK_CHUNK_BITS);
          const unsigned long long mi = __ballot(ins);
          if (mi) {   // wave-uniform: first occurrences append their slot to the list
            const unsigned first = (unsigned)(__ffsll((long long)mi) - 1);
            uint32_t base = 0;
            if (lane == first) base = atomicAdd(&s_nlist, (uint32_t)__popcll(mi));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)first);   // (first is wave-uniform: no LDS crossbar)
            if (ins) slist[base + (uint32_t)__popcll(mi & ((1ULL << lane) - 1))] = (uint16_t)h[u];
          }
          if (wh) atomicMin(match ? &hkey[h[u]] : &hidle[lane], match ? (want[u] | chk[u]) : 0xFFFFFFFFu);   // first chunk of the key
          atomicAdd(match ? &hcnt[h[u]] : &hidle[lane], match ? ((FUSED && !bef[u]) ? wgt[u] << 16 : wgt[u]) : 0u);
          pend[u] = pend[u] && !match;
          h[u] = pend[u] ? ((h[u] + 1) & (SHK_HCAP - 1)) : h[u];
        }
        if (A.ablate & 512) break;      // diagnostics: one probe step only
        if (++guard > SHK_HCAP) { hfull = pend[0] || pend[1] || pend[2] || pend[3]; break; }
      }
    }
    if (__ballot(corrupt) && corrupt) atomicOr(A.err, SHK_E_CORRUPT);
    if (__ballot(hfull) && hfull) atomicOr(&s_fail, SHK_E_HASH_FULL);
  }
  if (my_added) atomicAdd(&s_added, my_added);
  if (FUSED && my_added_b) atomicAdd(&s_added_b, my_added_b);
  __syncthreads();
  SHK_STAMP(1);   // key folding
  // Two things are left to prepare before the merge pass, and they do not depend on each other: the structure of the
  // OLD runs (rank/select over the staged blocks) and the grouping of the NEW keys by quotient. The second wave of the
  // group takes the first while the first wave does the second (a group of one wave does both); barriers inside the
  // two parts are wave-local. Then the helper leaves and one wave carries on.
  if (tid >= 2 * nthr) return;
  const bool helper = ngrp > nthr && tid >= nthr;
  if (helper || ngrp == nthr) {
    // ---- old structure: occupieds of the own blocks, runends inside [olo, ohi)
    const unsigned ln = tid & (SHK_WAVE - 1);
    {
      uint64_t ow = 0;
      if (ln < nown) ow = shk_ld64(oimg + ln * SHK_BLOCK_BYTES + SHK_OFF_OCC);
      if (ln < SHK_REGION_BLOCKS) oocc[ln] = ow;
      const uint32_t pc = (uint32_t)__popcll(ow);
      const uint32_t inc = shk_wave_incl_add(pc);
      if (ln < SHK_REGION_BLOCKS) oorank[ln] = inc - pc;
      if (ln == SHK_REGION_BLOCKS - 1) oorank[SHK_REGION_BLOCKS] = inc;
      uint64_t rw = 0;
      if (old_any && !fatal && ln < nblk_old) {
        rw = shk_ld64(oimg + ln * SHK_BLOCK_BYTES + SHK_OFF_RUN);
        const uint32_t s0 = ln * 64;
        if (s0 + 64 <= olo || s0 >= ohi) rw = 0;
        else {
          if (olo > s0) rw &= ~((1ULL << (olo - s0)) - 1);
          if (ohi < s0 + 64) rw &= ((1ULL << (ohi - s0)) - 1);
        }
      }
      const uint32_t rc = (uint32_t)__popcll(rw);
      const uint32_t rinc = shk_wave_incl_add(rc);
      if (ln < IMG_BLOCKS) { orrank[ln] = rinc - rc; orunw[ln] = rw; }
      if (ln == SHK_WAVE - 1) orrank[IMG_BLOCKS] = rinc;
    }
    shk_wave_sync();
    const uint32_t nruns = orrank[IMG_BLOCKS];
    if (nruns != oorank[SHK_REGION_BLOCKS] || nruns > SHK_REGION) {
      if (ln == 0) { atomicOr(A.err, SHK_E_CORRUPT); atomicOr(&s_fail, SHK_E_CORRUPT); }
    } else if (!fatal) {
      // position of the j-th runend: one lane per run (select over the masked runends words)
      for (uint32_t j = ln; j < nruns; j += SHK_WAVE) {
        uint32_t lo = 0, hi = IMG_BLOCKS;   // last word w with orrank[w] <= j
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (orrank[mid] <= j) lo = mid; else hi = mid;
        }
        orend[j] = (uint16_t)(lo * 64 + shk_select64(orunw[lo], j - orrank[lo]));
      }
    }
    if (helper) { __syncthreads(); return; }   // (the join below)
    shk_wave_sync();
  }
  my_added = tid == 0 ? s_added : 0;
  my_added_b = (FUSED && tid == 0) ? s_added_b : 0;
  SHK_STAMP(2);   // old structure (rank/select) when this wave did it
  // ---- group the new entries by quotient (counting sort of hash slots), sort by remainder
  const uint32_t nlist = (A.ablate & 16) ? 0 : s_nlist;
  for (uint32_t i = tid; i < nlist; i += nthr) { const uint32_t q = hkey[slist[i]] >> (SHK_CHUNK_BITS + 8); atomicAdd(&qcntw[q >> 1], 1u << (16 * (q & 1))); }
  shk_wave_sync();
  constexpr uint32_t per = SHK_REGION / nthr;  // consecutive quotients per lane
  const uint32_t qa = tid * per;
  {
    uint32_t sacc = 0;
#pragma unroll
    for (uint32_t j = 0; j < per; j++) sacc += qcnt[qa + j];
    uint32_t ex = shk_wave_incl_add(sacc) - sacc;
#pragma unroll
    for (uint32_t j = 0; j < per; j++) { qoff[qa + j] = (uint16_t)ex; ex += qcnt[qa + j]; }
    if (tid == nthr - 1) qoff[SHK_REGION] = (uint16_t)ex;
  }
  shk_wave_sync();
  for (uint32_t i = tid; i < nlist; i += nthr) {
    const uint32_t h = slist[i];
    const uint32_t q = hkey[h] >> (SHK_CHUNK_BITS + 8);
    const uint32_t pos = qoff[q] + (((atomicSub(&qcntw[q >> 1], 1u << (16 * (q & 1))) >> (16 * (q & 1))) & 0xFFFFu) - 1);
    nidx[pos] = (uint16_t)h;
  }
  shk_wave_sync();
  for (uint32_t j = 0; j < per; j++) {
    const uint32_t q = qa + j;
    const uint32_t a = qoff[q], b = qoff[q + 1];
    for (uint32_t i = a + 1; i < b; i++) {  // insertion sort: runs are short
      const uint16_t x = nidx[i];
      const uint32_t rx = (hkey[x] >> SHK_CHUNK_BITS) & 0xff;
      uint32_t k2 = i;
      while (k2 > a && ((hkey[nidx[k2 - 1]] >> SHK_CHUNK_BITS) & 0xff) > rx) { nidx[k2] = nidx[k2 - 1]; k2--; }
      nidx[k2] = x;
    }
  }
  // (each lane sorted only its own quotients' segments)
  __syncthreads();                       // join: the old structure is in place (the helper wave leaves here)
  if (s_fail & SHK_E_CORRUPT) fatal = true;
  if (tid < SHK_HIST_BINS) lhist[tid] = 0;   // (orunw is free from here on)
  shk_wave_sync();

  SHK_STAMP(3);   // counting sort + per-quotient sort
  // ---- one pass over the quotients: merge old run and new keys -> run length, statistics,
  // and (write modes) the run's encoding staged per lane
  uint32_t my_new = 0, my_removed = 0, my_before = 0;
  // (region-relative free-pointer functions: 32-bit arithmetic inside the kernel)
  ShkMPw mine; mine.a = 0; mine.b = SHK_NEG_INF_W;
  ShkMPw mine_i; mine_i.a = 0; mine_i.b = SHK_NEG_INF_W;   // FUSED: the same for the intermediate table
  uint32_t l4i = 0;                // FUSED: my four run lengths in the intermediate table, one byte each
  bool big_i = false;
  uint32_t st_used = 0;            // staged bytes of this lane
  bool st_over = false;            // a run did not fit: this lane re-merges at placement time
  uint8_t *mystage = stage + tid * SHK_STAGE_STRIDE;
  // The lane's four quotients are merged in ONE loop over (quotient, remainder) order: the old entries
  // of its runs on one side, its new keys (nidx is sorted the same way) on the other. A wave runs the
  // loop as often as its busiest lane has entries -- about half of what four per-quotient loops cost.
  if (qa < nq && !fatal && !(A.ablate & 2)) {
    const uint64_t ow = oocc[qa >> 6];
    uint32_t occ4 = (uint32_t)(ow >> (qa & 63)) & ((1u << per) - 1);          // which of my quotients had a run
    uint32_t jr = oorank[qa >> 6] + (uint32_t)__popcll(ow & ((1ULL << (qa & 63)) - 1));  // index of my first old run
    uint32_t oprev = jr ? (uint32_t)orend[jr - 1] + 1 : olo;                  // first slot behind the previous old run
    uint32_t oq = 0, opos = 0, oend = 0, orem = 0, on = 0;
    uint64_t ocnt = 0;
    bool ohas = false;
    if (occ4) {
      oq = qa + (uint32_t)__ffs((int)occ4) - 1; occ4 &= occ4 - 1;
      oend = orend[jr];
      opos = oprev > oq ? oprev : oq;
      on = shk_img_dec_fast(oimg, opos, oend, &orem, &ocnt);
      ohas = true;
    }
    uint32_t ni = qoff[qa];
    const uint32_t ne = qoff[qa + per];
    uint32_t nh = 0, nkey = 0;
    constexpr uint32_t NONE = 0xFFFFFFu;
    uint32_t ncomp = NONE;                                                   // quotient << 8 | remainder of the next new key
    if (ni < ne) { nh = nidx[ni]; nkey = hkey[nh]; ncomp = nkey >> SHK_CHUNK_BITS; }
    uint32_t curq = 0xFFFFFFFFu, len = 0, ilen = 0;
    while (ohas || ni < ne) {
      const uint32_t ocomp = ohas ? ((oq << 8) | orem) : NONE;
      const uint32_t comp = ocomp < ncomp ? ocomp : ncomp;
      const bool take_old = ocomp == comp, take_new = ncomp == comp;
      const uint32_t eq = comp >> 8, rem = comp & 0xff;
      if (eq != curq) {                     // the previous quotient's run is complete
        if (len) {
          qcnt[curq] = (uint16_t)len;
          ShkMPw m; m.a = (int)len; m.b = (int)(curq + len);
          mine = shk_mpw_compose(mine, m);
        }
        if (FUSED && ilen) {
          l4i |= (ilen > 255u ? 255u : ilen) << (8 * (curq - qa)); big_i |= ilen > 255u;
          ShkMPw m; m.a = (int)ilen; m.b = (int)(curq + ilen);
          mine_i = shk_mpw_compose(mine_i, m);
        }
        curq = eq; len = 0; ilen = 0;
      }
      uint64_t total = 0;
      bool is_new = false;
      uint32_t mc = 0, nhx = 0;
      if (FUSED) {
        // one-pass deNoise point: cb = the key's count when the round runs (old + chunks <= split), ca = what arrives behind it
        uint64_t cb = take_old ? ocnt : 0, ca = 0;
        if (take_old) {
          opos += on;
          if (opos <= oend) on = shk_img_dec_fast(oimg, opos, oend, &orem, &ocnt);
          else if (occ4) {
            oq = qa + (uint32_t)__ffs((int)occ4) - 1; occ4 &= occ4 - 1;
            jr++;
            oprev = oend + 1;
            oend = orend[jr];
            opos = oprev > oq ? oprev : oq;
            on = shk_img_dec_fast(oimg, opos, oend, &orem, &ocnt);
          } else ohas = false;
        }
        if (take_new) {
          { const uint32_t c2 = hcnt[nh]; cb += c2 & 0xFFFFu; ca = c2 >> 16; }
          if (A.newchunks && !take_old) hcnt[nh] = 0x80000000u;   // a key the table has not seen: its first chunk is collected below (the counts are consumed)
          ni++;
          ncomp = NONE;
          if (ni < ne) { nh = nidx[ni]; nkey = hkey[nh]; ncomp = nkey >> SHK_CHUNK_BITS; }
        }
        if (cb) ilen += shk_enc_len_fast(rem, cb);            // its slots in the intermediate table
        bool prot = false;
        if (cb == 1) {
          if (A.prot_list) {                                   // a singleton the range walk leaves alone? (sorted list, few entries)
            const uint64_t gq = q0 + eq;
            uint32_t lo2 = 0, hi2 = A.nprot;
            while (lo2 < hi2) { const uint32_t mid = (lo2 + hi2) >> 1; if (A.prot_list[mid] < gq) lo2 = mid + 1; else hi2 = mid; }
            prot = lo2 < A.nprot && A.prot_list[lo2] == gq;
          }
          if (!prot) my_removed++;
        }
        const uint64_t kept = cb >= 2 ? cb : (prot ? 1 : 0);
        total = kept + ca;
        if (total == 0) continue;
        if (kept == 0) my_new++;                               // (a dropped key that comes back counts as new, as after the reference's sweep)
      } else {
      if (take_old) {
        total = ocnt;
        if (A.denoise) {
          // a deNoise round drops the OLD singleton unless k_denoise_marks protected it (traveled bit in table A);
          // words of this pass (chunks behind the deNoise point) then count as if the key had never been seen
          const uint32_t tb = (opos >> 6) * SHK_BLOCK_BYTES + SHK_OFF_TRAV + ((opos & 63) >> 3);
          const bool prot = (oimg[tb] >> (opos & 7)) & 1;
          if (ocnt < 2 && !prot) { my_removed++; total = 0; is_new = true; }
        }
        opos += on;
        if (opos <= oend) on = shk_img_dec_fast(oimg, opos, oend, &orem, &ocnt);
        else if (occ4) {                    // my next old run
          oq = qa + (uint32_t)__ffs((int)occ4) - 1; occ4 &= occ4 - 1;
          jr++;
          oprev = oend + 1;
          oend = orend[jr];
          opos = oprev > oq ? oprev : oq;
          on = shk_img_dec_fast(oimg, opos, oend, &orem, &ocnt);
        } else ohas = false;
      } else is_new = true;
      if (take_new) {
        total += hcnt[nh];
        mc = nkey & (SHK_MAX_CHUNKS - 1); nhx = nh;
        ni++;
        ncomp = NONE;
        if (ni < ne) { nh = nidx[ni]; nkey = hkey[nh]; ncomp = nkey >> SHK_CHUNK_BITS; }
      } else is_new = false;
      if (total == 0) continue;
      }
      if (!FUSED && is_new) {
        my_new++;
        if (MODE != 1 && A.want_hist) {
          // exact mode: the key is only flagged here and its first chunk collected below (the count
          // was consumed above); the coarse histogram would cost one global atomic per bin and region
          if (A.newchunks) hcnt[nhx] |= 0x80000000u;
          else if (mc < A.hist_base) my_before++;
          else {
            uint32_t bin = (mc - A.hist_base) >> A.hist_shift;
            atomicAdd(&lhist[bin < SHK_HIST_BINS ? bin : SHK_HIST_BINS - 1], 1u);
          }
        }
      }
      const unsigned el = shk_enc_len_fast(rem, total);
      if (STAGE) {
        if (!st_over && st_used + el <= SHK_STAGE_PER_LANE) {
          if (total <= 128) {           // fast encode
            mystage[st_used] = (uint8_t)rem;
            if (total > 1) {
              const unsigned cdig = (unsigned)(total - 1);
              unsigned o = 1;
              if (cdig > rem) mystage[st_used + o++] = 0;
              mystage[st_used + o] = (uint8_t)cdig;
            }
          } else {
            uint8_t enc[12];
            const unsigned n = shk_enc_write(enc, rem, total);
            for (unsigned i = 0; i < n; i++) mystage[st_used + i] = enc[i];
          }
          st_used += el;
        } else st_over = true;
      }
      len += el;
    }
    if (len) {
      qcnt[curq] = (uint16_t)len;
      ShkMPw m; m.a = (int)len; m.b = (int)(curq + len);
      mine = shk_mpw_compose(mine, m);
    }
    if (FUSED && ilen) {
      l4i |= (ilen > 255u ? 255u : ilen) << (8 * (curq - qa)); big_i |= ilen > 255u;
      ShkMPw m; m.a = (int)ilen; m.b = (int)(curq + ilen);
      mine_i = shk_mpw_compose(mine_i, m);
    }
  }
  SHK_STAMP(4);   // merge pass
  if (FUSED) {
    // the intermediate table's (T, c) and run lengths: what the range walk of the round needs to know about this region
    const ShkMPw inc_i = shk_mpw_wave_scan(mine_i);
    reinterpret_cast<uint32_t *>(A.ilens + (size_t)r * SHK_REGION)[tid] = l4i;
    if (__ballot(big_i) && tid == 0) atomicOr(A.err, SHK_E_FUSED);
    if (tid == SHK_WAVE - 1) {
      A.isum[2 * (size_t)r] = fatal ? 0 : (uint32_t)inc_i.a;
      A.isum[2 * (size_t)r + 1] = (!fatal && inc_i.b > 0) ? (uint32_t)inc_i.b : 0;
      if (inc_i.a > 0xFFFF) atomicOr(A.err, SHK_E_FUSED);
    }
  }
  // wave scan of the free-pointer functions (lane order = quotient order)
  const ShkMPw incl = shk_mpw_wave_scan(mine);
  ShkMP tot, pre;          // (64-bit from here on: the placement works with absolute slots)
  tot.a = __builtin_amdgcn_readlane(incl.a, SHK_WAVE - 1);
  { const int tb = __builtin_amdgcn_readlane(incl.b, SHK_WAVE - 1); tot.b = tb > 0 ? tb : SHK_NEG_INF; }
  pre.a = 0; pre.b = SHK_NEG_INF;
  if (MODE == 1 || MODE == 2) {   // (only the placement needs the function in front of every lane)
    pre.a = __shfl_up(incl.a, 1);
    { const int pb = __shfl_up(incl.b, 1); pre.b = pb > 0 ? pb : SHK_NEG_INF; }
    if (tid == 0) { pre.a = 0; pre.b = SHK_NEG_INF; }
  }

  // per-region statistics (summed later by k_region_scan_c / k_stats_reduce: no same-address atomics)
  if (MODE != 1) {
    const uint32_t t_added = shk_wave_incl_add(my_added), t_new = shk_wave_incl_add(my_new),
                   t_removed = shk_wave_incl_add(my_removed), t_before = shk_wave_incl_add(FUSED ? my_added_b : my_before);
    if (tid == SHK_WAVE - 1) {
      uint32_t *sm = A.summary + (size_t)SHK_SUM_STRIDE * r;
      sm[0] = fatal ? 0 : (uint32_t)tot.a;
      sm[1] = (!fatal && tot.b > 0) ? (uint32_t)tot.b : 0;
      sm[2] = t_new; sm[3] = t_added; sm[4] = t_removed; sm[5] = t_before;
      if (!(A.want_hist && A.newchunks && !fatal)) sm[7] = 0;      // no first-chunk record of this region
      if (tot.a > 0xFFFF) atomicOr(A.err, SHK_E_RUN_TOO_LONG);
      if (s_fail) atomicOr(A.err, s_fail);
    }
    if (A.want_hist && !A.newchunks && tid < SHK_HIST_BINS && lhist[tid]) atomicAdd(&A.hist[tid], (unsigned long long)lhist[tid]);
  }
  SHK_STAMP(5);   // scan + statistics
  if (MODE != 1 && A.want_hist && A.newchunks && !fatal) {
    // first chunks of the region's new keys, compacted: k_chunk_hist turns them into the exact
    // per-chunk histogram from which the host reads the chunk of a deNoise point
    uint16_t *nc = A.newchunks + (size_t)r * SHK_NC_CAP;
    uint32_t base = 0;
    for (uint32_t i0 = 0; i0 < nlist; i0 += SHK_WAVE) {     // nidx[0, nlist) = every hash slot in use
      const uint32_t h = i0 + tid < nlist ? nidx[i0 + tid] : 0;
      const bool f = i0 + tid < nlist && (hcnt[h] >> 31);
      const unsigned long long m = __ballot(f);
      if (f) {
        const uint32_t at = base + (uint32_t)__popcll(m & ((1ULL << tid) - 1));
        const uint32_t ch = hkey[h] & (SHK_MAX_CHUNKS - 1);
        if (at < SHK_NC_CAP) nc[at] = (uint16_t)ch;
        else atomicAdd(&A.chist[ch], 1ULL);                 // (a region with more than SHK_NC_CAP new keys in one pass: rare)
      }
      base += (uint32_t)__popcll(m);
    }
    if (tid == 0) A.summary[(size_t)SHK_SUM_STRIDE * r + 7] = base < SHK_NC_CAP ? base : SHK_NC_CAP;   // entries of this region's record (k_chunk_hist)
  }
  if (MODE == 3) {
    // spill: 4 length bytes per lane, then the lanes' staged bytes back to back
    uint8_t *sp = A.spill + (size_t)r * SHK_SPILL_STRIDE;
    uint32_t l4 = 0;
    bool big = st_over;
#pragma unroll
    for (uint32_t j = 0; j < per; j++) {
      const uint32_t len = qcnt[qa + j];
      if (len > 255) big = true;
      l4 |= (len & 255u) << (8 * j);
    }
    reinterpret_cast<uint32_t *>(sp)[tid] = l4;
    const uint32_t sincl = shk_wave_incl_add(st_used);
    const uint32_t stot = (uint32_t)__builtin_amdgcn_readlane((int)sincl, SHK_WAVE - 1);
    const bool over = !fatal && (__ballot(big) != 0 || stot > SHK_SPILL_PACK_MAX);
    if (tid == 0) {
      A.summary[(size_t)SHK_SUM_STRIDE * r + 6] = over ? 1 : 0;
      if (over) A.over_list[atomicAdd(A.n_over, 1ULL)] = r;
      if (over && FUSED) atomicOr(A.err, SHK_E_FUSED);        // (the write pass for such regions does not know the split)
    }
    if (!over) {
      const uint32_t ex = sincl - st_used;
      for (uint32_t i = 0; i < st_used; i++) sp[SHK_SPILL_LENS + ex + i] = mystage[i];
    }
    return;
  }
  if (MODE == 0 || (A.ablate & 128)) return;

  // ================= placement =================
  long long fin_rel, fout_rel;
  if (MODE == 1) {
    fin_rel = (long long)A.finB[r] - (long long)q0;
    fout_rel = (long long)A.finB[r + 1] - (long long)q0;
  } else {
    // ---- single launch: look back over the regions before this one.
    // Every region publishes (T, c) as soon as it knows them (lb_agg) and its outgoing free
    // pointer once it knows its own incoming one (lb_incl). A window r'..r-1 composes to
    // f -> max(f + a, b); since no region's runs may end more than IMG_SLOTS behind its
    // start, f_in(r') <= start(r') + IMG_SLOTS - SHK_REGION, so the window already decides
    // f_in(r) = b as soon as start(r') + IMG_SLOTS - SHK_REGION + a <= b.
    if (tid == 0) {
      const uint32_t c_rel = (!fatal && tot.b > 0) ? (uint32_t)tot.b : 0;
      __hip_atomic_store(&A.lb_agg[r], (unsigned long long)(SHK_LB_VALID | ((uint32_t)tot.a << 12)) << 32 | c_rel,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    long long f_in = 0;
    bool done = (r == 0) || (A.ablate & 32);
    ShkMP win; win.a = 0; win.b = SHK_NEG_INF;       // composition of the regions already walked (nearest first)
    uint32_t back = 0;                                // regions walked so far
    uint32_t spins = 0;
    while (!done) {
      const long long rr = (long long)r - 1 - back - tid;   // this lane's predecessor
      unsigned long long incl_w = 0, agg_w = 0;
      if (rr >= 0) {
        incl_w = __hip_atomic_load(&A.lb_incl[rr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(incl_w & SHK_LB_INCL)) agg_w = __hip_atomic_load(&A.lb_agg[rr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const bool has_incl = rr >= 0 && (incl_w & SHK_LB_INCL);
      const bool has_agg = rr >= 0 && ((agg_w >> 32) & SHK_LB_VALID);
      const bool ready = rr < 0 || has_incl || has_agg;
      // lanes are ordered nearest-first; use the prefix of lanes that are ready
      const unsigned long long ready_m = __ballot(ready);
      const unsigned nready = ready_m == ~0ULL ? 64u : (unsigned)(__ffsll((long long)~ready_m) - 1);
      const unsigned long long stop_m = __ballot(rr < 0 || has_incl);   // lanes that end the walk
      // compose lane by lane (wave-uniform loop over the ready prefix)
      unsigned used = 0;
      for (; used < nready; used++) {
        const long long rr_u = (long long)r - 1 - back - used;
        if ((stop_m >> used) & 1) {
          const unsigned long long iw = __shfl(incl_w, used);
          const long long fo = rr_u < 0 ? 0 : (long long)(iw & ~SHK_LB_INCL);
          f_in = shk_mp_apply(win, fo);
          done = true;
          break;
        }
        const unsigned long long aw = __shfl(agg_w, used);
        ShkMP m;
        m.a = (long long)(((aw >> 32) & 0x7FFFFFFFu) >> 12);
        const uint32_t c_rel = (uint32_t)aw;
        m.b = c_rel ? rr_u * SHK_REGION + c_rel : SHK_NEG_INF;
        win = shk_mp_compose(m, win);      // farther region first, then what we had
        const long long bound = rr_u * SHK_REGION + (IMG_SLOTS - SHK_REGION);
        if (bound + win.a <= win.b) { f_in = win.b; done = true; break; }
      }
      if (!done) {
        back += used;
        if (used == 0) {
          if (++spins > (1u << 18)) { if (tid == 0) atomicOr(A.err, SHK_E_LOOKBACK); fatal = true; break; }
          __builtin_amdgcn_s_sleep(8);
        }
      }
    }
    // tot is in region-relative slots; f_in is absolute
    const long long f_out = fatal ? 0 : shk_mp_apply(tot, f_in - (long long)q0) + (long long)q0;
    if (tid == 0 && !(fatal && spins > (1u << 18)))
      __hip_atomic_store(&A.lb_incl[r], SHK_LB_INCL | (unsigned long long)f_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
      A.finB[r + 1] = (uint64_t)f_out;
      if (r == 0) A.finB[0] = 0;
      if (tot.a > 0 && f_out - (long long)q0 > IMG_SLOTS) atomicOr(A.err, SHK_E_NEW_EXTENT);
      if ((uint64_t)f_out > A.xnslots) atomicOr(A.err, SHK_E_TABLE_FULL);
    }
    fin_rel = f_in - (long long)q0;
    fout_rel = f_out - (long long)q0;
  }
  SHK_STAMP(6);   // look-back
  if (fatal) return;
  const uint32_t out_lo = fin_rel > 0 ? (uint32_t)fin_rel : 0;
  const bool new_any = tot.a > 0;
  const uint32_t out_hi = new_any ? (uint32_t)fout_rel : out_lo;
  if (new_any && (fout_rel > IMG_SLOTS || fout_rel < 0)) {
    if (tid == 0) atomicOr(A.err, SHK_E_NEW_EXTENT);
    return;
  }
  // run starts and block offsets: walk my quotients with the running free pointer; place the staged bytes
  uint32_t *nimg32 = reinterpret_cast<uint32_t *>(nimg);
  {
    long long f = shk_mp_apply(pre, fin_rel);
    uint32_t so = 0;
    for (uint32_t j = 0; j < per; j++) {
      const uint32_t q = qa + j;
      if (q < nq && (q & 63) == 0) {
        // offset = slots at the block start still owned by earlier runs (block_offset_strict,
        // gqf.c:599-601), saturating at 255 like the reference's 8-bit field
        long long o = f - (long long)q;
        nimg[(q >> 6) * SHK_BLOCK_BYTES] = (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o));
      }
      const uint32_t len = qcnt[q];
      if (len) {
        const long long stt = f > (long long)q ? f : (long long)q;
        rstart[q] = (uint16_t)stt;
        f = stt + len;
        if (!st_over) {
          const uint32_t wp = (uint32_t)stt;
          for (uint32_t i = 0; i < len; i++) nimg[shk_img_slot_off(wp + i)] = mystage[so + i];
          so += len;
          const uint32_t last = wp + len - 1;  // runend bit on the run's last slot; occupied bit on q
          const uint32_t bo = (last >> 6) * SHK_BLOCK_BYTES + SHK_OFF_RUN + ((last & 63) >> 3);
          atomicOr(&nimg32[bo >> 2], 1u << (((bo & 3) << 3) + (last & 7)));
          const uint32_t oo = (q >> 6) * SHK_BLOCK_BYTES + SHK_OFF_OCC + ((q & 63) >> 3);
          atomicOr(&nimg32[oo >> 2], 1u << (((oo & 3) << 3) + (q & 7)));
        }
      }
    }
  }
  // lanes whose runs did not fit the staging area merge once more, writing straight into the image
  if (st_over)
  for (uint32_t j = 0; j < per; j++) {
    const uint32_t q = qa + j;
    if (q >= nq || qcnt[q] == 0 || (A.ablate & 4)) continue;
    const bool occ = (oocc[q >> 6] >> (q & 63)) & 1;
    uint32_t opos = 0, oend = 0;
    bool ohas = false;
    if (occ) {
      const uint32_t jr = oorank[q >> 6] + (uint32_t)__popcll(oocc[q >> 6] & ((1ULL << (q & 63)) - 1));
      oend = orend[jr];
      opos = jr ? (uint32_t)orend[jr - 1] + 1 : olo;
      if (opos < q) opos = q;
      ohas = true;
    }
    uint32_t ni = qoff[q];
    const uint32_t ne = qoff[q + 1];
    uint32_t orem = 0, on = 0; uint64_t ocnt = 0;
    if (ohas) on = shk_img_dec(oimg, opos, oend, &orem, &ocnt);
    uint32_t wp = rstart[q];
    uint8_t enc[12];
    uint32_t nrem = 256, nh = 0;
    if (ni < ne) { nh = nidx[ni]; nrem = (hkey[nh] >> SHK_CHUNK_BITS) & 0xff; }
    while (ohas || ni < ne) {
      uint32_t rem; uint64_t total; bool prot = false; bool adv = false;
      if (ohas && orem <= nrem) {
        rem = orem; total = ocnt;
        if (A.denoise) {
          const uint32_t tb = (opos >> 6) * SHK_BLOCK_BYTES + SHK_OFF_TRAV + ((opos & 63) >> 3);
          prot = (oimg[tb] >> (opos & 7)) & 1;
        }
        if (A.denoise && ocnt < 2 && !prot) total = 0;
        if (orem == nrem) { total += hcnt[nh]; adv = true; }
        opos += on;
        if (opos <= oend) on = shk_img_dec(oimg, opos, oend, &orem, &ocnt); else ohas = false;
      } else {
        rem = nrem; total = hcnt[nh]; adv = true;
      }
      if (adv) {
        ni++;
        nrem = 256;
        if (ni < ne) { nh = nidx[ni]; nrem = (hkey[nh] >> SHK_CHUNK_BITS) & 0xff; }
      }
      if (total == 0) continue;
      const unsigned n = shk_enc_write(enc, rem, total);
      for (unsigned i = 0; i < n; i++) nimg[shk_img_slot_off(wp + i)] = enc[i];
      wp += n;
    }
    const uint32_t last = wp - 1;
    {
      const uint32_t bo = (last >> 6) * SHK_BLOCK_BYTES + SHK_OFF_RUN + ((last & 63) >> 3);
      atomicOr(&nimg32[bo >> 2], 1u << (((bo & 3) << 3) + (last & 7)));
      const uint32_t oo = (q >> 6) * SHK_BLOCK_BYTES + SHK_OFF_OCC + ((q & 63) >> 3);
      atomicOr(&nimg32[oo >> 2], 1u << (((oo & 3) << 3) + (q & 7)));
    }
  }
  __syncthreads();

  SHK_STAMP(7);   // placement into the image
  if (A.ablate & 8) return;
  shk_store_image<IMGB>(A, r, nregions, nimg, tid, nown, b0, q0, out_lo, out_hi, new_any, fout_rel);
  SHK_STAMP(8);   // stores to table B
}

// Second launch of the spill scheme: one wave per region places the spilled run bytes from the
// region's free pointer (k_region_scan_*) into an LDS image of its blocks and stores table B.
// Regions flagged in the summary (runs too long for the spill record) are left to MODE 1.
template <int IMGB>
__global__ void __launch_bounds__(SHK_WAVE) k_region_place(ShkMergeArgs A) {
  constexpr unsigned IMG_SLOTS = IMGB * 64, IMG_BYTES = IMGB * SHK_BLOCK_BYTES;
  __shared__ __attribute__((aligned(16))) uint8_t nimg[IMG_BYTES + 16];
  __shared__ __attribute__((aligned(16))) uint8_t pack[SHK_SPILL_PACK_MAX];
  const unsigned tid = threadIdx.x;
  const uint32_t r = blockIdx.x + A.r0;
  const uint32_t nregions = (uint32_t)((A.nslots + SHK_REGION - 1) / SHK_REGION);
  const uint32_t *sm = A.summary + (size_t)SHK_SUM_STRIDE * r;
  if (sm[6]) return;
  const uint64_t q0 = (uint64_t)r * SHK_REGION;
  const uint32_t nq = (uint32_t)((A.nslots - q0) < SHK_REGION ? (A.nslots - q0) : SHK_REGION);
  const uint32_t nown = (nq + 63) / 64;
  const uint64_t b0 = q0 / 64;
  const bool new_any = sm[0] > 0;
  const long long fin_rel = (long long)A.finB[r] - (long long)q0;
  const long long fout_rel = (long long)A.finB[r + 1] - (long long)q0;
  if (new_any && (fout_rel > IMG_SLOTS || fout_rel < 0)) {
    if (tid == 0) atomicOr(A.err, SHK_E_NEW_EXTENT);
    return;
  }
  const uint8_t *sp = A.spill + (size_t)r * SHK_SPILL_STRIDE;
  const uint32_t l4 = reinterpret_cast<const uint32_t *>(sp)[tid];
  {
    uint4 *z = reinterpret_cast<uint4 *>(nimg);          // (16-byte stores; nimg is 16-byte aligned and padded)
    for (uint32_t i = tid; i < (IMG_BYTES + 16) / 16; i += SHK_WAVE) z[i] = make_uint4(0, 0, 0, 0);
    if (tid == 0) for (uint32_t i = ((IMG_BYTES + 16) / 16) * 16; i < IMG_BYTES + 16; i++) nimg[i] = 0;
  }
  constexpr uint32_t per = SHK_REGION / SHK_WAVE;
  const uint32_t qa = tid * per;
  ShkMPw mine; mine.a = 0; mine.b = SHK_NEG_INF_W;       // (region-relative: 32 bits, scanned with DPP moves)
  uint32_t st_used = 0;
#pragma unroll
  for (uint32_t j = 0; j < per; j++) {
    const uint32_t len = (l4 >> (8 * j)) & 255u;
    if (len) {
      ShkMPw m; m.a = (int)len; m.b = (int)(qa + j + len);
      mine = shk_mpw_compose(mine, m);
      st_used += len;
    }
  }
  const ShkMPw incl = shk_mpw_wave_scan(mine);
  ShkMP pre;
  pre.a = __shfl_up(incl.a, 1);
  { const int pb = __shfl_up(incl.b, 1); pre.b = pb > 0 ? pb : SHK_NEG_INF; }
  if (tid == 0) { pre.a = 0; pre.b = SHK_NEG_INF; }
  const uint32_t sinc = shk_wave_incl_add(st_used);
  const uint32_t ex = sinc - st_used;
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)sinc, SHK_WAVE - 1);
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(sp + SHK_SPILL_LENS);
    uint32_t *dst = reinterpret_cast<uint32_t *>(pack);
    for (uint32_t i = tid; i < (total + 3) / 4; i += SHK_WAVE) dst[i] = src[i];
  }
  __syncthreads();
  const uint32_t out_lo = fin_rel > 0 ? (uint32_t)fin_rel : 0;
  const uint32_t out_hi = new_any ? (uint32_t)fout_rel : out_lo;
  uint32_t *nimg32 = reinterpret_cast<uint32_t *>(nimg);
  {
    long long f = shk_mp_apply(pre, fin_rel);
    uint32_t so = ex;
#pragma unroll
    for (uint32_t j = 0; j < per; j++) {
      const uint32_t q = qa + j;
      if (q < nq && (q & 63) == 0) {
        long long o = f - (long long)q;   // block_offset_strict, gqf.c:599-601, saturating like the 8-bit field
        nimg[(q >> 6) * SHK_BLOCK_BYTES] = (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o));
      }
      const uint32_t len = (l4 >> (8 * j)) & 255u;
      if (len) {
        const long long stt = f > (long long)q ? f : (long long)q;
        f = stt + len;
        const uint32_t wp = (uint32_t)stt;
        for (uint32_t i = 0; i < len; i++) nimg[shk_img_slot_off(wp + i)] = pack[so + i];
        so += len;
        const uint32_t last = wp + len - 1;
        const uint32_t bo = (last >> 6) * SHK_BLOCK_BYTES + SHK_OFF_RUN + ((last & 63) >> 3);
        atomicOr(&nimg32[bo >> 2], 1u << (((bo & 3) << 3) + (last & 7)));
        const uint32_t oo = (q >> 6) * SHK_BLOCK_BYTES + SHK_OFF_OCC + ((q & 63) >> 3);
        atomicOr(&nimg32[oo >> 2], 1u << (((oo & 3) << 3) + (q & 7)));
      }
    }
  }
  __syncthreads();
  shk_store_image<IMGB>(A, r, nregions, nimg, tid, nown, b0, q0, out_lo, out_hi, new_any, fout_rel);
}

// exact histogram of first chunks over all regions (input: newchunks + the regions' new-key counts)
#define SHK_CHIST_REGIONS 2048   // regions per workgroup
__global__ void k_chunk_hist(const uint16_t *newchunks, const uint32_t *summary, uint32_t nregions, unsigned long long *chist,
                             uint32_t rstride = 1) {
  __shared__ uint32_t lh[SHK_MAX_CHUNKS];
  for (uint32_t i = threadIdx.x; i < SHK_MAX_CHUNKS; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  const uint32_t lane = shk_lane(), wave = shk_wave(), nw = blockDim.x / SHK_WAVE;
  const uint32_t r0 = blockIdx.x * SHK_CHIST_REGIONS;
  for (uint32_t i = wave; i < SHK_CHIST_REGIONS; i += nw) {
    if ((uint64_t)(r0 + i) * rstride >= nregions) break;
    const uint32_t r = (r0 + i) * rstride;                    // (rstride > 1: the sampled statistics pass)
    const uint32_t n = summary[(size_t)SHK_SUM_STRIDE * r + 7];
    const uint16_t *nc = newchunks + (size_t)r * SHK_NC_CAP;
    for (uint32_t j = lane; j < n && j < SHK_NC_CAP; j += SHK_WAVE) atomicAdd(&lh[nc[j]], 1u);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < SHK_MAX_CHUNKS; i += blockDim.x)
    if (lh[i]) atomicAdd(&chist[i], (unsigned long long)lh[i]);
}

// statistics of a single-launch rebuild (MODE 2): sum the per-region records
__global__ void k_stats_reduce(const uint32_t *summary, uint32_t nregions, unsigned long long *counters) {
  __shared__ uint64_t scratch64[SHK_MAX_WAVES + 1];
  const uint32_t per = SHK_RSCAN_TILE / blockDim.x;
  const uint32_t r0 = blockIdx.x * SHK_RSCAN_TILE + threadIdx.x * per;
  for (int z = 0; z < 4; z++) {
    uint64_t v = 0;
    for (uint32_t j = 0; j < per; j++)
      if (r0 + j < nregions) v += summary[(size_t)SHK_SUM_STRIDE * (r0 + j) + 2 + z];
    const uint64_t t = shk_block_sum64(v, scratch64);
    if (threadIdx.x == 0 && t) atomicAdd(&counters[z], (unsigned long long)t);
  }
}

// ---------------------------------------------------------------- free pointers
// fin[r+1] = max(fin[r] + T_r, region start + c_r), as a 3-launch scan over tiles of regions.
__device__ __forceinline__ ShkMP shk_region_mp(const uint32_t *summary, uint32_t r, uint32_t nregions, uint32_t stride = SHK_SUM_STRIDE) {
  ShkMP m; m.a = 0; m.b = SHK_NEG_INF;
  if (r < nregions) {
    m.a = summary[(size_t)stride * r];
    const uint32_t c = summary[(size_t)stride * r + 1];
    if (c) m.b = (long long)r * SHK_REGION + c;
  }
  return m;
}
// A tile's (T, c) pairs, read ONCE and coalesced into LDS (lane i takes record i: a thread that walked its own `per`
// consecutive 32-byte records touched 64 lines per load, and the records came up from memory a dozen times: 0.78 GB per
// launch for 67 MB of summaries, profiles/r03_e_pmc_traffic.txt); stats[0..3] += the four statistics words of the records
// this thread loaded (stride >= 6 only)
__device__ __forceinline__ void shk_scan_tile_load(const uint32_t *summary, uint32_t base, uint32_t nregions, uint32_t stride, uint2 *ac,
                                                   uint64_t *stats) {
  for (uint32_t i = threadIdx.x; i < SHK_RSCAN_TILE; i += blockDim.x) {
    const uint32_t r = base + i;
    uint2 v = make_uint2(0u, 0u);
    if (r < nregions) {
      if (stride == SHK_SUM_STRIDE) {
        const uint4 lo = *reinterpret_cast<const uint4 *>(summary + (size_t)SHK_SUM_STRIDE * r);
        v = make_uint2(lo.x, lo.y);
        if (stats) {
          const uint2 hi = *reinterpret_cast<const uint2 *>(summary + (size_t)SHK_SUM_STRIDE * r + 4);
          stats[0] += lo.z; stats[1] += lo.w; stats[2] += hi.x; stats[3] += hi.y;
        }
      } else {
        v = make_uint2(summary[(size_t)stride * r], summary[(size_t)stride * r + 1]);
        if (stats && stride >= 6)
          for (int z = 0; z < 4; z++) stats[z] += summary[(size_t)stride * r + 2 + z];
      }
    }
    ac[i] = v;
  }
  __syncthreads();
}
__device__ __forceinline__ ShkMP shk_tile_mp(const uint2 *ac, uint32_t i, uint32_t r) {
  ShkMP m; m.a = ac[i].x; m.b = SHK_NEG_INF;
  if (ac[i].y) m.b = (long long)r * SHK_REGION + ac[i].y;
  return m;
}
// a: one workgroup per tile -> the tile's composed function
// (stride = words per region record: SHK_SUM_STRIDE for the summaries, 2 for the intermediate table of a one-pass deNoise point)
__global__ void k_region_scan_a(const uint32_t *summary, uint32_t nregions, long long *tile_a, long long *tile_b, uint32_t stride = SHK_SUM_STRIDE) {
  __shared__ long long mpa[SHK_MAX_WAVES + 1], mpb[SHK_MAX_WAVES + 1];
  __shared__ uint2 ac[SHK_RSCAN_TILE];
  const uint32_t base = blockIdx.x * SHK_RSCAN_TILE;
  const uint32_t per = SHK_RSCAN_TILE / blockDim.x;
  shk_scan_tile_load(summary, base, nregions, stride, ac, nullptr);
  ShkMP mine; mine.a = 0; mine.b = SHK_NEG_INF;
  for (uint32_t j = 0; j < per; j++) mine = shk_mp_compose(mine, shk_tile_mp(ac, threadIdx.x * per + j, base + threadIdx.x * per + j));
  ShkMP tot;
  shk_block_exscan_mp(mine, &tot, mpa, mpb);
  if (threadIdx.x == 0) { tile_a[blockIdx.x] = tot.a; tile_b[blockIdx.x] = tot.b; }
}
// b: one workgroup: free pointer at every tile start
// (f0: free pointer in front of the first region -- 0, or what the shards before this one spill over the border when the
// layout of the whole filter is wanted)
__global__ void k_region_scan_b(const long long *tile_a, const long long *tile_b, uint32_t ntiles, long long *tile_f, long long f0 = 0) {
  __shared__ long long mpa[SHK_MAX_WAVES + 1], mpb[SHK_MAX_WAVES + 1];
  __shared__ long long carry_s;
  if (threadIdx.x == 0) carry_s = f0;
  __syncthreads();
  for (uint32_t b0 = 0; b0 < ntiles; b0 += blockDim.x) {
    const uint32_t t = b0 + threadIdx.x;
    ShkMP m; m.a = 0; m.b = SHK_NEG_INF;
    if (t < ntiles) { m.a = tile_a[t]; m.b = tile_b[t]; }
    ShkMP tot;
    ShkMP pre = shk_block_exscan_mp(m, &tot, mpa, mpb);
    const long long carry = carry_s;
    if (t < ntiles) tile_f[t] = shk_mp_apply(pre, carry);
    __syncthreads();
    if (threadIdx.x == 0) carry_s = shk_mp_apply(tot, carry);
    __syncthreads();
  }
}
// c: one workgroup per tile: fin[] for its regions + the capacity checks
__global__ void k_region_scan_c(const uint32_t *summary, uint32_t nregions, const long long *tile_f, uint64_t xnslots,
                                uint32_t img_slots, uint64_t *fin, unsigned long long *counters, uint32_t *err,
                                uint32_t stride = SHK_SUM_STRIDE) {
  __shared__ long long mpa[SHK_MAX_WAVES + 1], mpb[SHK_MAX_WAVES + 1];
  __shared__ uint64_t scratch64[SHK_MAX_WAVES + 1];
  __shared__ uint2 ac[SHK_RSCAN_TILE];
  const uint32_t base = blockIdx.x * SHK_RSCAN_TILE;
  const uint32_t per = SHK_RSCAN_TILE / blockDim.x;
  const uint32_t r0 = base + threadIdx.x * per;
  uint64_t stats[4] = {0, 0, 0, 0};
  shk_scan_tile_load(summary, base, nregions, stride, ac, stats);
  ShkMP mine; mine.a = 0; mine.b = SHK_NEG_INF;
  for (uint32_t j = 0; j < per; j++) mine = shk_mp_compose(mine, shk_tile_mp(ac, threadIdx.x * per + j, r0 + j));
  ShkMP tot;
  ShkMP pre = shk_block_exscan_mp(mine, &tot, mpa, mpb);
  long long f = shk_mp_apply(pre, tile_f[blockIdx.x]);
  if (blockIdx.x == 0 && threadIdx.x == 0) fin[0] = (uint64_t)tile_f[0];
  uint64_t slots = 0;
  for (uint32_t j = 0; j < per; j++) {
    const uint32_t r = r0 + j;
    if (r >= nregions) break;
    const ShkMP m = shk_tile_mp(ac, threadIdx.x * per + j, r);
    f = shk_mp_apply(m, f);
    fin[r + 1] = (uint64_t)f;
    slots += (uint64_t)m.a;
    if (m.a > 0 && f - (long long)r * SHK_REGION > (long long)img_slots) atomicOr(err, SHK_E_NEW_EXTENT);
    if ((uint64_t)f > xnslots) atomicOr(err, SHK_E_TABLE_FULL);
  }
  if (stride == 2) {        // intermediate table of a one-pass deNoise point: slots in use (its shard's share of the free-pointer function)
    const uint64_t t = shk_block_sum64(slots, scratch64);
    if (threadIdx.x == 0 && t) atomicAdd(&counters[4 + SHK_HIST_BINS + 1], (unsigned long long)t);
  }
  // statistics of this tile's regions
  if (stride >= 6)
  for (int z = 0; z < 4; z++) {
    const uint64_t t = shk_block_sum64(stats[z], scratch64);
    if (threadIdx.x == 0 && t) atomicAdd(&counters[z], (unsigned long long)t);
  }
}

// ---------------------------------------------------------------- reference-style probes
// The same block arithmetic as the reference (run_end gqf.c:655-704, offset_lower_bound
// :706-718, find_first_empty_slot :738-748, find_first_nonempty_slot :751-774), reading the
// packed table in HBM. Used by the lookup kernel and by the deNoise range walk.
__device__ __forceinline__ uint64_t shk_g_occ(const uint8_t *t, uint64_t b) { return shk_ld64(t + b * SHK_BLOCK_BYTES + SHK_OFF_OCC); }
__device__ __forceinline__ uint64_t shk_g_run(const uint8_t *t, uint64_t b) { return shk_ld64(t + b * SHK_BLOCK_BYTES + SHK_OFF_RUN); }
__device__ __forceinline__ unsigned shk_g_off(const uint8_t *t, uint64_t b) { return t[b * SHK_BLOCK_BYTES]; }
__device__ __forceinline__ unsigned shk_g_slot(const uint8_t *t, uint64_t p) {
  return t[(p >> 6) * SHK_BLOCK_BYTES + SHK_OFF_SLOTS + (p & 63)];
}
__device__ __forceinline__ uint64_t shk_mask_lt(unsigned n) { return n >= 64 ? ~0ULL : ((1ULL << n) - 1); }

// end slot of the last run with quotient <= q (>= q). Saturated offsets (255) are resolved
// by walking left to a block whose offset is exact, instead of the reference's recursion.
__device__ uint64_t shk_g_run_end(const uint8_t *t, uint64_t q) {
  const uint64_t bi = q >> 6;
  const unsigned off = q & 63;
  // exact offset of block bi
  uint64_t bb = bi;
  while (bb > 0 && shk_g_off(t, bb) == 255) bb--;
  // free pointer (relative count form): walk forward from bb to bi keeping the exact offset
  uint64_t boff = shk_g_off(t, bb);
  while (bb < bi) {
    // offset of block bb+1 = end of last run with quotient < 64(bb+1), minus 64(bb+1), plus 1
    const uint64_t occ = shk_g_occ(t, bb);
    unsigned nruns = (unsigned)__popcll(occ);
    uint64_t endpos;
    if (nruns == 0) {
      endpos = boff ? 64 * bb + boff - 1 : 0;
      uint64_t nb = 64 * (bb + 1);
      boff = (boff && endpos + 1 > nb) ? endpos + 1 - nb : 0;
    } else {
      // find the nruns-th runend at or after slot 64*bb + boff
      uint64_t rb = bb + boff / 64;
      unsigned ignore = boff % 64;
      unsigned rank = nruns - 1;
      for (;;) {
        uint64_t w = shk_g_run(t, rb) & ~shk_mask_lt(ignore);
        unsigned c = (unsigned)__popcll(w);
        if (rank < c) { endpos = 64 * rb + shk_select64(w, rank); break; }
        rank -= c; rb++; ignore = 0;
      }
      uint64_t nb = 64 * (bb + 1);
      boff = endpos + 1 > nb ? endpos + 1 - nb : 0;
    }
    bb++;
  }
  const unsigned rank = (unsigned)__popcll(shk_g_occ(t, bi) & shk_mask_lt(off + 1));
  if (rank == 0) {
    if (boff <= off) return q;
    return 64 * bi + boff - 1;
  }
  uint64_t rb = bi + boff / 64;
  unsigned ignore = boff % 64;
  unsigned rr = rank - 1;
  uint64_t endpos;
  for (;;) {
    uint64_t w = shk_g_run(t, rb) & ~shk_mask_lt(ignore);
    unsigned c = (unsigned)__popcll(w);
    if (rr < c) { endpos = 64 * rb + shk_select64(w, rr); break; }
    rr -= c; rb++; ignore = 0;
  }
  return endpos < q ? q : endpos;
}

__device__ __forceinline__ int shk_g_offset_lower_bound(const uint8_t *t, uint64_t slot) {
  const uint64_t b = slot >> 6;
  const unsigned so = slot & 63;
  const unsigned boff = shk_g_off(t, b);
  const uint64_t occ = shk_g_occ(t, b) & shk_mask_lt(so + 1);
  if (boff <= so) {
    const uint64_t re = (shk_g_run(t, b) & shk_mask_lt(so)) >> boff;
    return __popcll(occ) - __popcll(re);
  }
  return (int)(boff - so) + __popcll(occ);
}
__device__ uint64_t shk_g_first_empty(const uint8_t *t, uint64_t from, uint64_t nblocks) {
  for (;;) {
    if ((from >> 6) >= nblocks) return from;  // memory past the table reads as zero
    int lb = shk_g_offset_lower_bound(t, from);
    if (lb == 0) return from;
    from += lb;
  }
}
__device__ uint64_t shk_g_first_nonempty(const uint8_t *t, uint64_t from, uint64_t nblocks, uint64_t xnslots) {
  uint64_t b = from >> 6;
  if (b >= nblocks) return xnslots;
  uint64_t w = shk_g_occ(t, b) & ~shk_mask_lt(from & 63);
  while (!w) {
    b++;
    if (b >= nblocks) return xnslots;
    w = shk_g_occ(t, b);
  }
  uint64_t nx = 64 * b + (uint64_t)(__ffsll((long long)w) - 1);
  return nx >= xnslots ? xnslots : nx;
}

// The reference's deNoise walks work ranges of >= min_len slots that end on a cluster end
// (CQF_mt.h:888-895) and, inside a range, never visits a cluster that STARTS on the range's
// last slot (`while(start < end_bucket_id)`, CQF_mt.h:1024, gqf.c:2881). Such a cluster is
// a single count-1 slot; it survives the round. This kernel reproduces the walk on table A
// and marks those slots by setting their traveled bit (zero otherwise during a build); the
// merge kernels in denoise mode keep marked entries. One thread: the walk is a dependent chain.
__global__ void k_denoise_marks(uint8_t *tab, uint64_t nslots, uint64_t xnslots, uint64_t nblocks, uint64_t min_len,
                                unsigned long long *nmarked) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint64_t cur = shk_g_first_nonempty(tab, 0, nblocks, xnslots);
  unsigned long long marked = 0;
  while (cur < nslots) {
    uint64_t end = cur + min_len > nslots ? nslots : cur + min_len;
    end = shk_g_first_empty(tab, end, nblocks) - 1;
    // one-slot cluster exactly at `end`: slot end in use, slot end-1 empty
    bool used = (end >> 6) < nblocks && shk_g_offset_lower_bound(tab, end) != 0;
    bool prev_empty = end == 0 || shk_g_offset_lower_bound(tab, end - 1) == 0;
    if (used && prev_empty) {
      uint8_t *p = tab + (end >> 6) * SHK_BLOCK_BYTES + SHK_OFF_TRAV + ((end & 63) >> 3);
      *p = (uint8_t)(*p | (1u << (end & 7)));
      marked++;
    }
    cur = shk_g_first_nonempty(tab, end + 1, nblocks, xnslots);
  }
  *nmarked = marked;
}

// The same range walk on a table that has not been written: the INTERMEDIATE table of a one-pass deNoise point, known
// through its free pointer at every region start (fin) and its run length per quotient (lens, one byte each). Slot s is
// in use iff the free pointer behind quotient s lies beyond s; "first non-empty" is the next quotient with a run
// (find_first_nonempty_slot looks at the occupieds bits, gqf.c:751-774). Protected singletons are reported as quotients
// (a one-slot cluster sits on its home slot). The ranges follow one another (each starts where the previous one ended),
// but inside a step ONE WAVE works on a whole region at a time: lane i owns quotients 4i..4i+3 of the region, a
// max-plus scan over the lanes gives every lane the free pointer in front of its quotients (the region's own anchor is
// fin[r], so regions are independent), ballots find the first empty slot / the next run.
struct ShkVRegion {            // what a lane knows of region r after shk_vregion()
  uint64_t q0;                 // region's first quotient
  uint64_t fpre;               // free pointer in front of my first quotient
  uint32_t l4;                 // my four run lengths (one byte each)
};
__device__ __forceinline__ ShkVRegion shk_vregion(const uint64_t *fin, const uint8_t *lens, uint64_t nslots, uint64_t r, unsigned lane) {
  ShkVRegion R;
  R.q0 = r << SHK_REGION_LOG2;
  const uint64_t myq = R.q0 + 4 * lane;
  R.l4 = myq < nslots ? reinterpret_cast<const uint32_t *>(lens + R.q0)[lane] : 0u;   // (regions are whole: nslots is a multiple of 64, lens of absent quotients are 0)
  // (relative to the region's first quotient: 32 bits, scanned with DPP moves -- this kernel is one wave and a chain of
  // dependent steps, every shuffle through the LDS crossbar is latency on that chain)
  ShkMPw mine; mine.a = 0; mine.b = SHK_NEG_INF_W;
#pragma unroll
  for (unsigned j = 0; j < 4; j++) {
    const uint32_t l = (R.l4 >> (8 * j)) & 255u;
    if (l) { ShkMPw m; m.a = (int)l; m.b = (int)(4 * lane + j + l); mine = shk_mpw_compose(mine, m); }
  }
  const ShkMPw incl = shk_mpw_wave_scan(mine);
  ShkMP pre;
  pre.a = __shfl_up(incl.a, 1);
  { const int pb = __shfl_up(incl.b, 1); pre.b = pb > 0 ? pb : SHK_NEG_INF; }
  if (lane == 0) { pre.a = 0; pre.b = SHK_NEG_INF; }
  R.fpre = (uint64_t)(shk_mp_apply(pre, (long long)fin[r] - (long long)R.q0) + (long long)R.q0);
  return R;
}
// A filter sharded by quotient range is walked shard by shard (ShkWalkShard): fin is then laid out with the free pointer
// the earlier shards carry over the border, so local coordinates + the shard's first quotient = coordinates of the single
// table; a shard stops where it would have to look at slots of the next one and leaves the state of the walk behind:
// state[0] = 0: between two ranges, the next one starts at the first run at or behind local slot state[1];
// state[0] = 1: inside a range whose minimum end is local slot state[1] (of the NEXT shard when written, of this one when
// read). The single table is the one-shard case (last = 1, state_in = {0, 0}).
struct ShkWalkShard {
  long long prev_fp;        // free pointer of everything in front of this shard, local (negative: ends before the border); shard 0: -1
  uint64_t cap_local;       // slots from this shard's first to the end of the whole filter (range ends are capped there)
  int last;                 // the filter ends with this shard
  int next_first_used;      // quotient 0 of the next shard has a run (its slot is the first one behind this shard)
};
__global__ void __launch_bounds__(SHK_WAVE) k_denoise_marks_virtual(const uint64_t *fin, const uint8_t *lens, const uint32_t *isum, uint64_t nslots,
                                                                    uint64_t xnslots, uint64_t min_len, uint64_t *prot, uint32_t cap,
                                                                    unsigned long long *nprot, ShkWalkShard W, const uint64_t *state_in,
                                                                    uint64_t *state_out) {
  if (blockIdx.x != 0) return;
  const unsigned lane = threadIdx.x & (SHK_WAVE - 1);
  const uint64_t nregions = (nslots + SHK_REGION - 1) / SHK_REGION;
  // free pointer in front of quotient x and behind it (wave-uniform results)
  auto fp_at = [&](uint64_t x, uint64_t *behind) -> uint64_t {
    const uint64_t r = x >> SHK_REGION_LOG2;
    if (r >= nregions) { *behind = fin[nregions]; return fin[nregions]; }
    const ShkVRegion R = shk_vregion(fin, lens, nslots, r, lane);
    uint64_t f = R.fpre, fb = R.fpre;
    const unsigned owner = (unsigned)((x - R.q0) >> 2), jx = (unsigned)(x & 3);
    for (unsigned j = 0; j <= jx; j++) {
      const uint32_t l = (R.l4 >> (8 * j)) & 255u;
      const uint64_t q = R.q0 + 4 * lane + j;
      if (j == jx) { fb = l ? (f > q ? f : q) + l : f; break; }
      if (l) f = (f > q ? f : q) + l;
    }
    *behind = __shfl(fb, (int)owner);
    return __shfl(f, (int)owner);
  };
  // first empty slot >= x
  // (~0: every own slot from x on is in use -- the answer lies behind this shard)
  auto first_empty = [&](uint64_t x) -> uint64_t {
    for (uint64_t r = x >> SHK_REGION_LOG2;; r++) {
      if (r >= nregions) return ~0ULL;
      const ShkVRegion R = shk_vregion(fin, lens, nslots, r, lane);
      uint64_t f = R.fpre, found = ~0ULL;
#pragma unroll
      for (unsigned j = 0; j < 4; j++) {
        const uint32_t l = (R.l4 >> (8 * j)) & 255u;
        const uint64_t q = R.q0 + 4 * lane + j;
        if (l) f = (f > q ? f : q) + l;                 // pointer behind quotient q
        if (q >= x && q < nslots && f <= q && found == ~0ULL) found = q;
      }
      const unsigned long long m = __ballot(found != ~0ULL);
      if (m) return __shfl(found, __ffsll((long long)m) - 1);
      if (x < R.q0 + SHK_REGION) x = R.q0 + SHK_REGION;
    }
  };
  // next quotient >= x that has a run; xnslots when there is none
  auto first_nonempty = [&](uint64_t x) -> uint64_t {
    for (uint64_t r = x >> SHK_REGION_LOG2; r < nregions; r++) {
      const uint64_t q0 = r << SHK_REGION_LOG2;
      if (isum[2 * r] != 0) {
        const uint64_t myq = q0 + 4 * lane;
        const uint32_t l4 = myq < nslots ? reinterpret_cast<const uint32_t *>(lens + q0)[lane] : 0u;
        uint64_t found = ~0ULL;
#pragma unroll
        for (unsigned j = 0; j < 4; j++)
          if (((l4 >> (8 * j)) & 255u) && myq + j >= x && found == ~0ULL) found = myq + j;
        const unsigned long long m = __ballot(found != ~0ULL);
        if (m) return __shfl(found, __ffsll((long long)m) - 1);
      }
    }
    return xnslots;
  };
  unsigned long long n = 0;
  uint64_t kind = state_in[0], x = state_in[1];
  uint64_t out0 = 0, out1 = 0;
  const bool boundary_used = fin[nregions] > nslots || W.next_first_used;     // the first slot behind this shard
  for (;;) {
    uint64_t end0;
    if (kind == 0) {
      const uint64_t cur = first_nonempty(x);
      if (cur >= nslots) break;                       // no further range starts in this shard ({0, 0} for the next one)
      end0 = cur + min_len > W.cap_local ? W.cap_local : cur + min_len;
    } else end0 = x;
    kind = 0;
    uint64_t e;
    if (!W.last && end0 > nslots) { out0 = 1; out1 = end0 - nslots; break; }
    if (end0 >= nslots) e = ~0ULL;
    else e = first_empty(end0);
    if (e == ~0ULL) {
      if (W.last) { const uint64_t fe = fin[nregions]; e = fe > end0 ? fe : end0; }
      else if (boundary_used) { out0 = 1; out1 = 0; break; }
      else e = nslots;
    }
    const uint64_t end = e - 1;
    // one-slot cluster exactly at `end`: slot end in use, slot end - 1 empty
    uint64_t f1;
    const uint64_t f0 = fp_at(end, &f1);
    const bool used = f1 > end;
    const bool prev_empty = end == 0 ? W.prev_fp <= -1 : f0 <= end - 1;
    if (used && prev_empty) {
      if (n < cap && lane == 0) prot[n] = end;
      n++;
    }
    x = end + 1;
  }
  if (lane == 0) { *nprot = n; state_out[0] = out0; state_out[1] = out1; }
}

// ---------------------------------------------------------------- lookups
// qf_count_key_value (gqf.c:2442-2469) and qf_count_key_value_set_traveled (:3092-3128):
// one thread per query. The traveled bit belongs to the first slot of the entry; it is set
// with a 32-bit atomic OR on the aligned word holding that bit (the reference uses a
// plain, racy |=, gqf.c:3078).
// mark: 0 read the traveled bit, 1 set it and return the previous value, 2 leave it alone
__device__ __forceinline__ uint64_t shk_lookup_one(uint8_t *tab, uint64_t key, uint64_t q_lo, uint64_t nslots, int mark,
                                                   uint8_t *trav_out) {
  const unsigned rem = key & 0xff;
  const uint64_t q = (key >> 8) - q_lo;
  uint64_t cnt = 0;
  uint8_t trav = 0;
  if (q < nslots && ((shk_g_occ(tab, q >> 6) >> (q & 63)) & 1)) {
    uint64_t rs = q == 0 ? 0 : shk_g_run_end(tab, q - 1) + 1;
    if (rs < q) rs = q;
    for (;;) {
      // decode_counter on the table in HBM
      const unsigned r0 = shk_g_slot(tab, rs);
      uint64_t c = 1, e = rs;
      const bool re0 = (shk_g_run(tab, rs >> 6) >> (rs & 63)) & 1;
      if (!re0) {
        unsigned d = shk_g_slot(tab, rs + 1);
        if (d <= r0) {
          e = rs + 1;
          uint64_t cc = 0;
          if (d == 0) { e++; d = shk_g_slot(tab, e); }
          while (d & 0x80) { cc = cc * 128 + (d & 0x7f); e++; d = shk_g_slot(tab, e); }
          cc = cc * 128 + d;
          c = cc + 1;
        }
      }
      if (r0 == rem) {
        cnt = c;
        uint8_t *tp = tab + (rs >> 6) * SHK_BLOCK_BYTES + SHK_OFF_TRAV + ((rs & 63) >> 3);
        const unsigned bit = 1u << (rs & 7);
        if (mark == 1) {
          uintptr_t ai = reinterpret_cast<uintptr_t>(tp);
          uint32_t *w = reinterpret_cast<uint32_t *>(ai & ~(uintptr_t)3);
          uint32_t old = atomicOr(w, bit << ((ai & 3) << 3));
          trav = (old >> ((ai & 3) << 3)) & bit ? 1 : 0;
        } else if (mark == 0) {
          trav = (*tp & bit) ? 1 : 0;
        }
        break;
      }
      if ((shk_g_run(tab, e >> 6) >> (e & 63)) & 1) break;
      rs = e + 1;
    }
  }
  *trav_out = trav;
  return cnt;
}

__global__ void k_lookup(uint8_t *tab, const uint64_t *keys, uint64_t n, uint64_t q_lo, uint64_t nslots, int mark,
                         uint64_t *counts, uint8_t *was_traveled) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t trav = 0;
  counts[i] = shk_lookup_one(tab, keys[i], q_lo, nslots, mark, &trav);
  if (was_traveled) was_traveled[i] = trav;
}

// free pointer at every region start of an existing table (after shk_import): end of the
// last run with a smaller quotient, plus one (any value <= the region start means "no spill")
__global__ void k_build_fin(const uint8_t *tab, uint64_t nslots, uint32_t nregions, uint64_t *fin) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > nregions) return;
  uint64_t q0 = (uint64_t)r * SHK_REGION;
  if (q0 > nslots) q0 = nslots;
  fin[r] = q0 == 0 ? 0 : shk_g_run_end(tab, q0 - 1) + 1;
}
