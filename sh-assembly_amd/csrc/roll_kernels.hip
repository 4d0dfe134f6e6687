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
  uint64_t *hist;                  // k_roll_hist: 2^dig_bits counters
  uint64_t *cursor;                // k_roll_scatter: next free position of every digit's bucket (starts at its base)
  uint64_t *out;
  uint64_t cap;                    // words `out` holds
  uint32_t *err;
};

// rows of the two tables in LDS: in[code] = {seed, rol(seedc, k-1)}, out[code] = {rol(seed, k), ror1(seedc)}; code 4 = 0
struct ShkRollTabs {
  uint64_t in_f[8], in_r[8], out_f[8], out_r[8];
  uint8_t code[256];
};
__device__ __forceinline__ void shk_roll_tabs_init(ShkRollTabs *t, uint32_t k) {
  const uint64_t sf[4] = {0x3c8bfbb395c60474ULL, 0x3193c18562a02b4cULL, 0x20323ed082572324ULL, 0x295549f54be24456ULL};  // A C G T, nthash.hpp:24-27
  for (uint32_t i = threadIdx.x; i < 8; i += blockDim.x) {
    const uint64_t f = i < 4 ? sf[i] : 0, c = i < 4 ? sf[3 - i] : 0;     // complement: A<->T, C<->G (cpOff, nthash.hpp:15)
    t->in_f[i] = f; t->in_r[i] = shk_rol64(c, k - 1);
    t->out_f[i] = shk_rol64(f, k); t->out_r[i] = shk_ror64(c, 1);
  }
  for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
    const uint32_t u = i & 0xDF;
    t->code[i] = u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : 4;
  }
}

// 16 bytes at `p` (any alignment); bytes at or behind safe_end read as 0
__device__ __forceinline__ uint4 shk_load16(const uint8_t *text, uint64_t at, uint64_t safe_end) {
  uint4 v;
  if (at + 16 <= safe_end) { __builtin_memcpy(&v, text + at, 16); return v; }
  uint32_t w[4] = {0, 0, 0, 0};
  for (int j = 0; j < 16; j++)
    if (at + j < safe_end) w[j >> 2] |= (uint32_t)text[at + j] << (8 * (j & 3));
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// One thread's place in its read (reads_to_kmers' loop state)
struct ShkRollState {
  uint64_t st;        // text offset of the read's first base
  uint32_t len;       // bases
  uint32_t i;         // next base to take
  uint32_t fill;      // bases of the current (sub)read taken so far, saturating at k
  uint64_t fh, rh;
};

// Takes the next (at most 16) bases of the read; calls emit(j, key) for the k-mer that step j (0..15, a compile-time
// constant after unrolling) completes. Returns false when the read is used up.
template <typename Emit>
__device__ __forceinline__ bool shk_roll_round(const ShkRollTabs *T, const uint8_t *text, uint64_t safe_end, ShkRollState &s,
                                               uint32_t k, uint64_t mask, Emit emit) {
  if (s.i >= s.len) return false;
  const uint4 vin = shk_load16(text, s.st + s.i, safe_end);
  // the base that leaves the window k steps behind; needed only once a window is full, i.e. never in front of the read
  const uint4 vout = s.i >= k ? shk_load16(text, s.st + s.i - k, safe_end) : make_uint4(0, 0, 0, 0);
  const uint32_t win[4] = {vin.x, vin.y, vin.z, vin.w}, wout[4] = {vout.x, vout.y, vout.z, vout.w};
  const uint32_t i0 = s.i;
#pragma unroll
  for (int j = 0; j < SHK_ROLL_STEPS; j++) {
    if (i0 + j < s.len) {
      const uint32_t cin = (win[j >> 2] >> (8 * (j & 3))) & 0xFFu;
      if (s.fill >= k && cin == 'N') {
        // the reference restarts behind an 'N' it meets at an index >= k of the (sub)read (CQF_mt.h:672-676)
        s.fill = 0; s.fh = 0; s.rh = 0;
      } else {
        const uint32_t ci = T->code[cin];
        uint64_t f = shk_rol64(s.fh, 1) ^ T->in_f[ci], r = shk_ror64(s.rh, 1) ^ T->in_r[ci];
        if (s.fill >= k) {
          // (i0 + j >= k here: the out byte lies k bases back, in vout when i0 >= k, else in vin itself)
          const uint32_t cout = i0 >= k ? (wout[j >> 2] >> (8 * (j & 3))) & 0xFFu : text[s.st + i0 + j - k];
          const uint32_t co = T->code[cout];
          f ^= T->out_f[co]; r ^= T->out_r[co];
        } else s.fill++;
        s.fh = f; s.rh = r;
        if (s.fill >= k) emit(j, (f < r ? f : r) & mask);
      }
    }
  }
  s.i = i0 + SHK_ROLL_STEPS < s.len ? i0 + SHK_ROLL_STEPS : s.len;
  return true;
}

__device__ __forceinline__ uint32_t shk_roll_digit(uint64_t key, const ShkRollArgs &A) {
  const uint32_t region = (uint32_t)(((key >> 8) - A.q_lo) >> SHK_REGION_LOG2);
  return (region >> A.dig_shift) & ((1u << A.dig_bits) - 1);
}

// pass 1: digit histogram (and, as its sum, the number of keys)
__global__ void __launch_bounds__(256) k_roll_hist(ShkRollArgs A) {
  __shared__ ShkRollTabs T;
  __shared__ uint32_t lh[1024];
  shk_roll_tabs_init(&T, A.k);
  const uint32_t P = 1u << A.dig_bits;
  for (uint32_t d = threadIdx.x; d < P; d += blockDim.x) lh[d] = 0;
  __syncthreads();
  const uint64_t nreads = *A.nreads_p;
  const uint64_t mask = A.hb >= 64 ? ~0ULL : ((1ULL << A.hb) - 1);
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += stride) {
    const uint64_t st = A.rd_start[r], en = A.rd_end[r];
    if (en - st > 65535) { atomicOr(A.err, SHK_E_BAD_FASTQ); continue; }     // SHK_MAX_READ
    if (en - st < A.k) continue;
    ShkRollState s = {st, (uint32_t)(en - st), 0, 0, 0, 0};
    while (shk_roll_round(&T, A.text, A.safe_end, s, A.k, mask, [&](int, uint64_t key) { atomicAdd(&lh[shk_roll_digit(key, A)], 1u); })) {}
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < P; d += blockDim.x)
    if (lh[d]) atomicAdd((unsigned long long *)&A.hist[d], (unsigned long long)lh[d]);
}

// pass 2: the keys again, straight into their buckets. THREADS x 16 keys per window.
template <int THREADS>
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
  ShkRollState s = {0, 0, 0, 0, 0, 0};
  uint64_t tag = 0;
  bool have = false;
  for (;;) {
    // a thread whose read is used up takes its next one
    while (!have && r < nreads) {
      const uint64_t st = A.rd_start[r], en = A.rd_end[r];
      if (en - st <= 65535 && en - st >= A.k) {
        s = {st, (uint32_t)(en - st), 0, 0, 0, 0};
        tag = (uint64_t)(A.chunk_first + A.rd_chunk[r] * A.chunk_mul) << A.hb;
        have = true;
      }
      r += stride;
    }
    for (uint32_t d = threadIdx.x; d < P; d += THREADS) lh[d] = 0;
    if (threadIdx.x == 0) any_left = 0;
    __syncthreads();
    uint64_t w[SHK_ROLL_STEPS];
    uint32_t dr[SHK_ROLL_STEPS];
    uint32_t vm = 0;               // steps that completed a k-mer
    if (have) {
      have = shk_roll_round(&T, A.text, A.safe_end, s, A.k, mask, [&](int j, uint64_t key) {
        const uint32_t d = shk_roll_digit(key, A);
        w[j] = key | tag;
        dr[j] = (d << 16) | atomicAdd(&lh[d], 1u);
        vm |= 1u << j;
      });
      if (have && s.i >= s.len) have = false;
    }
    if (have || r < nreads) any_left = 1;
    __syncthreads();
    const bool more = any_left != 0;
    // exclusive scan of the digit counts, one reservation per digit and window
    uint32_t carry = 0;
    for (uint32_t d0 = 0; d0 < P; d0 += THREADS) {
      const uint32_t d = d0 + threadIdx.x;
      uint32_t v = d < P ? lh[d] : 0, tot;
      const uint32_t ex = shk_block_exscan(v, &tot, scratch);
      if (d < P) {
        lbase[d] = carry + ex;
        gbase[d] = v ? atomicAdd((unsigned long long *)&A.cursor[d], (unsigned long long)v) : 0;
      }
      carry += tot;
    }
    __syncthreads();
    const uint32_t cnt = carry;
#pragma unroll
    for (int u = 0; u < SHK_ROLL_STEPS; u++)
      if ((vm >> u) & 1u) stage[lbase[dr[u] >> 16] + (dr[u] & 0xFFFFu)] = w[u];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cnt; i += THREADS) {
      const uint64_t x = stage[i];
      const uint32_t d = shk_roll_digit(A.hb >= 64 ? x : (x & ((1ULL << A.hb) - 1)), A);
      const uint64_t at = gbase[d] + (i - lbase[d]);
      if (at < A.cap) A.out[at] = x; else atomicOr(A.err, SHK_E_KEYS_FULL);
    }
    __syncthreads();
    if (!more) break;
  }
}
