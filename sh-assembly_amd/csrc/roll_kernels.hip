// FASTQ text -> key words that leave the kernel ALREADY PARTITIONED by the first region digit.
//
// Replaces, for shk_count_chunks, the wave-per-read closed-form hash (k_hash_reads: ~300 vector operations per k-mer for
// its two 64-lane XOR scans, 6.5 ms per 832 M k-mers) plus the first partition level (write 8 B, read 8 B, write 8 B per
// key: k_rp_scatter, 2.7 ms) by ONE THREAD PER READ that restates reads_to_kmers (cqf/CQF_mt.h:610-731) literally --
// the serial roll of base/nthash.hpp:305-309, the un-inspected first window, the restart behind an 'N' -- at about 30
// vector operations per k-mer, twice:
//   k_roll_hist      hashes every k-mer and counts the first-level digits (nothing is written but 128 counters);
//   k_roll_scatter   hashes every k-mer again and sends it straight to its digit's bucket through the same LDS staging
//                    the partition's scatter uses (16384-key windows, one LDS atomic per key for count and rank, digit
//                    runs written contiguously).
// Hashing twice costs less than one round trip of the keys through HBM. Order inside a bucket is free.
//
// The roll in table form, T = the base's row (A, C, G, T; anything else = seed 0, nthash.hpp:120-153):
//   fh' = rol1(fh) ^ rol(seed[out], k) ^ seed[in]            rh' = ror1(rh) ^ ror1(seedc[out]) ^ rol(seedc[in], k-1)
// and the first window of a (sub)read is the same recurrence from fh = rh = 0 with the `out` terms left away
// (k steps: fh = XOR rol(seed[c_j], k-1-j), rh = XOR rol(seedc[c_j], j), nthash.hpp:295-302).
#include "shk_device.h"

#define SHK_ROLL_STEPS 16          // bases per round and thread = one 16-byte load per stream

struct ShkRollArgs {
  const uint8_t *text;
  uint64_t safe_end;               // bytes of `text` that may be read (text_bytes rounded up to 16, see shk.h)
  const uint64_t *rd_start, *rd_end, *nreads_p;
  const uint16_t *rd_chunk;
  uint32_t chunk_first, chunk_mul, k, hb;
  uint64_t q_lo;
  uint32_t dig_shift, dig_bits;    // first partition level: digit = (region >> dig_shift) & (2^dig_bits - 1)
  uint32_t hist_shift, hist_bits;  // k_roll_hist: bin = (region >> hist_shift) & (2^hist_bits - 1)
  uint64_t *hist;                  // k_roll_hist: 2^hist_bits counters
  uint64_t *cursor;                // k_roll_scatter: next free position of every digit's bucket (starts at its base)
  uint64_t *out;
  uint64_t cap;                    // words `out` holds
  uint32_t *err;
};

// The two tables in LDS, indexed by the raw BYTE (no code lookup in between): row c of `in` = {seed[c], rol(seedc[c], k-1)},
// row c of `out` = {rol(seed[c], k), ror1(seedc[c])}; every byte that is not a base (either case) has zero rows.
struct ShkRollRow { uint64_t f, r; };
struct ShkRollTabs {
  ShkRollRow in[256], out[256];
};
__device__ __forceinline__ void shk_roll_tabs_init(ShkRollTabs *t, uint32_t k) {
  const uint64_t sf[4] = {0x3c8bfbb395c60474ULL, 0x3193c18562a02b4cULL, 0x20323ed082572324ULL, 0x295549f54be24456ULL};  // A C G T, nthash.hpp:24-27
  for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
    const uint32_t u = i & 0xDF;
    const int b = u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : -1;
    const uint64_t f = b >= 0 ? sf[b] : 0, c = b >= 0 ? sf[3 - b] : 0;   // complement: A<->T, C<->G (cpOff, nthash.hpp:15)
    t->in[i].f = f; t->in[i].r = shk_rol64(c, k - 1);
    t->out[i].f = shk_rol64(f, k); t->out[i].r = shk_ror64(c, 1);
  }
}
// rotations by one as funnel shifts of the halves (v_alignbit_b32 each; the 64-bit shifts run at a quarter of the rate)
__device__ __forceinline__ uint64_t shk_rol1(uint64_t x) {
  const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  return ((uint64_t)((hi << 1) | (lo >> 31)) << 32) | ((lo << 1) | (hi >> 31));
}
__device__ __forceinline__ uint64_t shk_ror1(uint64_t x) {
  const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  return ((uint64_t)((hi >> 1) | (lo << 31)) << 32) | ((lo >> 1) | (hi << 31));
}

// 16 bytes at `p` (any alignment); bytes at or behind safe_end read as 0
struct ShkQuad { uint32_t x, y, z, w; };      // (a plain aggregate: HIP's uint4 is a union inside, which keeps the state out of registers)
__device__ __forceinline__ ShkQuad shk_quad(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { ShkQuad q = {x, y, z, w}; return q; }
__device__ __forceinline__ ShkQuad shk_load16(const uint8_t *text, uint64_t at, uint64_t safe_end) {
  ShkQuad v;
  if (at + 16 <= safe_end) { __builtin_memcpy(&v, text + at, 16); return v; }
  uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 16; j++)
    if (at + j < safe_end) w[j >> 2] |= (uint32_t)text[at + j] << (8 * (j & 3));
  return shk_quad(w[0], w[1], w[2], w[3]);
}

// One thread's place in its read (reads_to_kmers' loop state). The text is fetched 64 bytes per stream at a time (four
// 16-byte loads issued together, every fourth round): with one load per round each 128-byte line came up from L2 once
// per 16 bytes used -- the lanes of a wave read 64 different lines, and no line survives in the CU's L1 until its
// owner's next round -- and the kernels ran at the speed of that traffic, not of their arithmetic.
#define SHK_ROLL_QUADS 4
struct ShkRollState {
  uint64_t st;        // text offset of the read's first base
  uint32_t len;       // bases
  uint32_t i;         // next base to take
  uint32_t fill;      // bases of the current (sub)read taken so far, saturating at k
  uint64_t fh, rh;
  ShkQuad i0_, i1_, i2_, i3_, o0_, o1_, o2_, o3_;      // bytes [i0b, i0b + 64) of the read and the 64 bytes k further back
};
// quad p of four (by masks: a chain of selects over neighbouring fields is turned into an indexed load, and the whole
// state then lives in scratch memory)
__device__ __forceinline__ ShkQuad shk_pick_quad(const ShkQuad &a, const ShkQuad &b, const ShkQuad &c, const ShkQuad &d, uint32_t p) {
  const uint32_t m0 = 0u - (uint32_t)(p == 0), m1 = 0u - (uint32_t)(p == 1), m2 = 0u - (uint32_t)(p == 2), m3 = 0u - (uint32_t)(p == 3);
  ShkQuad v;
  v.x = (a.x & m0) | (b.x & m1) | (c.x & m2) | (d.x & m3);
  v.y = (a.y & m0) | (b.y & m1) | (c.y & m2) | (d.y & m3);
  v.z = (a.z & m0) | (b.z & m1) | (c.z & m2) | (d.z & m3);
  v.w = (a.w & m0) | (b.w & m1) | (c.w & m2) | (d.w & m3);
  return v;
}

// Takes the next (at most 16) bases of the read; calls emit(j, key) for the k-mer that step j (0..15, a compile-time
// constant after unrolling) completes. Returns false when the read is used up.
// Two forms of the same recurrence. The STRAIGHT one runs when the thread is in the middle of a (sub)read: its window is
// full, the 16 bytes ahead hold no 'N', and the base that leaves lies in the second load -- no branches, the state is
// updated unconditionally (behind the end of the read it is never looked at again) and only the emission is predicated.
// The GENERAL one is reads_to_kmers step by step (first window, restart behind an 'N').
// NQ = 16-byte quads fetched at a time per stream (4 in the histogram pass; the scatter pass, whose 16 keys per thread
// wait in registers for their window, has room for 2)
template <int NQ, typename Emit>
__device__ __forceinline__ bool shk_roll_round(const ShkRollTabs *T, const uint8_t *text, uint64_t safe_end, ShkRollState &s,
                                               uint32_t k, uint64_t mask, Emit emit) {
  if (s.i >= s.len) return false;
  const uint32_t ph = (s.i / SHK_ROLL_STEPS) % NQ;
  if (ph == 0) {
    // the base that leaves the window lies k steps behind; it is needed only once a window is full, i.e. never in
    // front of the read
#define SHK_ROLL_LOAD(Q, IN, OUT) { const uint32_t at = s.i + SHK_ROLL_STEPS * Q; \
      IN = at < s.len ? shk_load16(text, s.st + at, safe_end) : shk_quad(0, 0, 0, 0); \
      OUT = (at >= k && at < s.len) ? shk_load16(text, s.st + at - k, safe_end) : shk_quad(0, 0, 0, 0); }
    SHK_ROLL_LOAD(0, s.i0_, s.o0_)
    if (NQ > 1) SHK_ROLL_LOAD(1, s.i1_, s.o1_)
    if (NQ > 2) { SHK_ROLL_LOAD(2, s.i2_, s.o2_) SHK_ROLL_LOAD(3, s.i3_, s.o3_) }
#undef SHK_ROLL_LOAD
  }
  ShkQuad vin, vout;
  if (NQ == 1) { vin = s.i0_; vout = s.o0_; }
  else if (NQ == 2) { vin = shk_pick_quad(s.i0_, s.i1_, s.i0_, s.i1_, ph); vout = shk_pick_quad(s.o0_, s.o1_, s.o0_, s.o1_, ph); }
  else { vin = shk_pick_quad(s.i0_, s.i1_, s.i2_, s.i3_, ph); vout = shk_pick_quad(s.o0_, s.o1_, s.o2_, s.o3_, ph); }
  const uint32_t win[4] = {vin.x, vin.y, vin.z, vin.w}, wout[4] = {vout.x, vout.y, vout.z, vout.w};
  const uint32_t i0 = s.i;
  uint32_t anyN = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint32_t x = win[q] ^ 0x4E4E4E4Eu;              // a zero byte = an 'N'
    anyN |= (x - 0x01010101u) & ~x & 0x80808080u;
  }
  if (s.fill >= k && i0 >= k && !anyN) {
    uint64_t fh = s.fh, rh = s.rh;
    const auto step = [&](int j, bool whole) {
      const int sh = 8 * (j & 3) - 4;                      // byte j of its word, times sizeof(ShkRollRow)
      const uint32_t oi = (sh < 0 ? win[j >> 2] << 4 : win[j >> 2] >> sh) & 0xFF0u;
      const uint32_t oo = (sh < 0 ? wout[j >> 2] << 4 : wout[j >> 2] >> sh) & 0xFF0u;
      const ShkRollRow ri = *reinterpret_cast<const ShkRollRow *>(reinterpret_cast<const uint8_t *>(T->in) + oi);
      const ShkRollRow ro = *reinterpret_cast<const ShkRollRow *>(reinterpret_cast<const uint8_t *>(T->out) + oo);
      fh = shk_rol1(fh) ^ ri.f ^ ro.f;
      rh = shk_ror1(rh) ^ ri.r ^ ro.r;
      if (whole || i0 + j < s.len) emit(j, (fh < rh ? fh : rh) & mask);
    };
    if (i0 + SHK_ROLL_STEPS <= s.len) {                    // (all but a read's last round: nothing is predicated)
#pragma unroll
      for (int j = 0; j < SHK_ROLL_STEPS; j++) step(j, true);
    } else {
#pragma unroll
      for (int j = 0; j < SHK_ROLL_STEPS; j++) step(j, false);
    }
    s.fh = fh; s.rh = rh;
  } else if (s.fill + SHK_ROLL_STEPS < k && i0 + SHK_ROLL_STEPS <= s.len) {
    // inside the first window of a (sub)read, which the reference hashes without looking at it: sixteen bases go in,
    // nothing comes out
    uint64_t fh = s.fh, rh = s.rh;
#pragma unroll
    for (int j = 0; j < SHK_ROLL_STEPS; j++) {
      const int sh = 8 * (j & 3) - 4;
      const uint32_t oi = (sh < 0 ? win[j >> 2] << 4 : win[j >> 2] >> sh) & 0xFF0u;
      const ShkRollRow ri = *reinterpret_cast<const ShkRollRow *>(reinterpret_cast<const uint8_t *>(T->in) + oi);
      fh = shk_rol1(fh) ^ ri.f;
      rh = shk_ror1(rh) ^ ri.r;
    }
    s.fh = fh; s.rh = rh; s.fill += SHK_ROLL_STEPS;
  } else {
#pragma unroll
    for (int j = 0; j < SHK_ROLL_STEPS; j++) {
      if (i0 + j < s.len) {
        const uint32_t cin = (win[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        if (s.fill >= k && cin == 'N') {
          // the reference restarts behind an 'N' it meets at an index >= k of the (sub)read (CQF_mt.h:672-676)
          s.fill = 0; s.fh = 0; s.rh = 0;
        } else {
          uint64_t f = shk_rol1(s.fh) ^ T->in[cin].f, r = shk_ror1(s.rh) ^ T->in[cin].r;
          if (s.fill >= k) {
            // (i0 + j >= k here: the base k back lies in vout when i0 >= k, else within this read's first bytes)
            const uint32_t cout = i0 >= k ? (wout[j >> 2] >> (8 * (j & 3))) & 0xFFu : text[s.st + i0 + j - k];
            f ^= T->out[cout].f; r ^= T->out[cout].r;
          } else s.fill++;
          s.fh = f; s.rh = r;
          if (s.fill >= k) emit(j, (f < r ? f : r) & mask);
        }
      }
    }
  }
  s.i = i0 + SHK_ROLL_STEPS < s.len ? i0 + SHK_ROLL_STEPS : s.len;
  return true;
}

// first-level digit of a key (masked to hb bits). The context's first quotient is a multiple of the region size, so
// region = ((key >> 8) - q_lo) >> 8 = (key >> 16) - (q_lo >> 8), and its low 32 bits are all there is (regions < 2^25)
__device__ __forceinline__ uint32_t shk_roll_digit(uint64_t key, const ShkRollArgs &A) {
  const uint32_t region = (uint32_t)(key >> 16) - (uint32_t)(A.q_lo >> 8);
  return (region >> A.dig_shift) & ((1u << A.dig_bits) - 1);
}

// pass 1: digit histogram (and, as its sum, the number of keys). LB = log2 of the LDS bins: 10 for the first level's
// digits alone; 14 when the first TWO levels' digits fit (7 + 7 bits for a qb-29 filter): the histogram of the second
// partition level then comes out of this pass too and its own pass over the keys (k_rp_hist, 1.1 ms) is not needed
// (hist_shift / hist_bits describe the combined digit; k_roll_fold sums it down to the first level's counts).
template <int LB, int THREADS>
__global__ void __launch_bounds__(THREADS) k_roll_hist(ShkRollArgs A) {
  __shared__ ShkRollTabs T;
  __shared__ uint32_t lh[1u << LB];
  shk_roll_tabs_init(&T, A.k);
  const uint32_t P = 1u << A.hist_bits;
  for (uint32_t d = threadIdx.x; d < P; d += blockDim.x) lh[d] = 0;
  __syncthreads();
  const uint64_t nreads = *A.nreads_p;
  const uint64_t mask = A.hb >= 64 ? ~0ULL : ((1ULL << A.hb) - 1);
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint32_t qlo16 = (uint32_t)(A.q_lo >> 8);
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += stride) {
    const uint64_t st = A.rd_start[r], en = A.rd_end[r];
    if (en - st > 65535) { atomicOr(A.err, SHK_E_BAD_FASTQ); continue; }     // SHK_MAX_READ
    if (en - st < A.k) continue;
    ShkRollState s;
    s.st = st; s.len = (uint32_t)(en - st); s.i = 0; s.fill = 0; s.fh = 0; s.rh = 0;
    while (shk_roll_round<4>(&T, A.text, A.safe_end, s, A.k, mask, [&](int, uint64_t key) {
      atomicAdd(&lh[(((uint32_t)(key >> 16) - qlo16) >> A.hist_shift) & (P - 1)], 1u); })) {}
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < P; d += blockDim.x)
    if (lh[d]) atomicAdd((unsigned long long *)&A.hist[d], (unsigned long long)lh[d]);
}
// first-level counts from the two-level histogram: hist0[d] = sum of hist2[d * P1 .. (d + 1) * P1)
__global__ void k_roll_fold(const uint64_t *hist2, uint32_t p0, uint32_t p1, uint64_t *hist0) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= p0) return;
  uint64_t t = 0;
  for (uint32_t j = 0; j < p1; j++) t += hist2[(uint64_t)d * p1 + j];
  hist0[d] = t;
}

// pass 2: the keys again, straight into their buckets. THREADS x 16 keys per window.
template <int THREADS, int NQ>
__global__ void __launch_bounds__(THREADS) k_roll_scatter(ShkRollArgs A) {
  constexpr uint32_t TILE = THREADS * SHK_ROLL_STEPS;
  __shared__ ShkRollTabs T;
  __shared__ uint32_t lh[1024], lbase[1024];
  __shared__ uint64_t gbase[1024];
  __shared__ uint64_t stage[TILE];
  __shared__ uint32_t scratch[SHK_MAX_WAVES + 1];
  __shared__ uint32_t any_left;
  shk_roll_tabs_init(&T, A.k);
  const uint32_t P = 1u << A.dig_bits;
  const uint64_t nreads = *A.nreads_p;
  const uint64_t mask = A.hb >= 64 ? ~0ULL : ((1ULL << A.hb) - 1);
  const uint64_t stride = (uint64_t)gridDim.x * THREADS;
  uint64_t r = (uint64_t)blockIdx.x * THREADS + threadIdx.x;
  ShkRollState s;
  s.st = 0; s.len = 0; s.i = 0; s.fill = 0; s.fh = 0; s.rh = 0;
  uint64_t tag = 0;
  bool have = false;
  for (;;) {
    // a thread whose read is used up takes its next one
    while (!have && r < nreads) {
      const uint64_t st = A.rd_start[r], en = A.rd_end[r];
      if (en - st <= 65535 && en - st >= A.k) {
        s.st = st; s.len = (uint32_t)(en - st); s.i = 0; s.fill = 0; s.fh = 0; s.rh = 0;
        tag = (uint64_t)(A.chunk_first + A.rd_chunk[r] * A.chunk_mul) << A.hb;
        have = true;
      }
      r += stride;
    }
    for (uint32_t d = threadIdx.x; d < P; d += THREADS) lh[d] = 0;
    if (threadIdx.x == 0) any_left = 0;
    shk_lds_barrier();             // (the previous window's runs keep draining to HBM while this one is hashed)
    uint64_t w[SHK_ROLL_STEPS];
    uint32_t vm = 0;               // steps that completed a k-mer
    if (have) {
      have = shk_roll_round<NQ>(&T, A.text, A.safe_end, s, A.k, mask, [&](int j, uint64_t key) {
        w[j] = key | tag;
        atomicAdd(&lh[shk_roll_digit(key, A)], 1u);
        vm |= 1u << j;
      });
      if (have && s.i >= s.len) have = false;
    }
    if (have || r < nreads) any_left = 1;
    __syncthreads();
    const bool more = any_left != 0;
    // exclusive scan of the digit counts, one reservation per digit and window. lbase[d] then serves as the digit's
    // cursor inside the window (the keys' ranks are handed out when they are staged: no rank waits in a register),
    // gbase[d] = where the digit's run goes, minus its place in the window
    uint32_t carry = 0;
    for (uint32_t d0 = 0; d0 < P; d0 += THREADS) {
      const uint32_t d = d0 + threadIdx.x;
      uint32_t v = d < P ? lh[d] : 0, tot;
      const uint32_t ex = shk_block_exscan(v, &tot, scratch);
      if (d < P) {
        lbase[d] = carry + ex;
        gbase[d] = (v ? atomicAdd((unsigned long long *)&A.cursor[d], (unsigned long long)v) : 0) - (carry + ex);
      }
      carry += tot;
    }
    __syncthreads();
    const uint32_t cnt = carry;
#pragma unroll
    for (int u = 0; u < SHK_ROLL_STEPS; u++)
      if ((vm >> u) & 1u) stage[atomicAdd(&lbase[shk_roll_digit(A.hb >= 64 ? w[u] : (w[u] & mask), A)], 1u)] = w[u];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cnt; i += THREADS) {
      const uint64_t x = stage[i];
      const uint32_t d = shk_roll_digit(A.hb >= 64 ? x : (x & mask), A);
      const uint64_t at = gbase[d] + i;
      if (at < A.cap) A.out[at] = x; else atomicOr(A.err, SHK_E_KEYS_FULL);
    }
    shk_lds_barrier();
    if (!more) break;
  }
}
