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
  // 2-bit staging (k_pack_reads), null = none: read r's bases, 64 per 16-byte unit, start at unit pk_base[r];
  // pk_flag[r] = its number of units (0 = too short or too long a read), bit 31 set = holds a byte that is no base
  const struct ShkQuad *pk;
  const uint64_t *pk_base;
  const uint32_t *pk_flag;
};

// The two tables in LDS, indexed by the raw BYTE (no code lookup in between): row c of `in` = {seed[c], rol(seedc[c], k-1)},
// row c of `out` = {rol(seed[c], k), ror1(seedc[c])}; every byte that is not a base (either case) has zero rows.
struct ShkRollRow { uint64_t f, r; };
struct ShkRollTabs {
  ShkRollRow in[256], out[256];
  ShkRollRow in4[4], out4[4];      // the same rows by 2-bit code (A C G T = 0 1 2 3), for reads staged by k_pack_reads
};
__device__ __forceinline__ void shk_roll_tabs_init(ShkRollTabs *t, uint32_t k) {
  const uint64_t sf[4] = {0x3c8bfbb395c60474ULL, 0x3193c18562a02b4cULL, 0x20323ed082572324ULL, 0x295549f54be24456ULL};  // A C G T, nthash.hpp:24-27
  for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
    // seedTab (nthash.hpp:120-153): the bases in either case, and the bytes 1 3 4 7 = their images under `& cpOff`
    // (nthash.hpp:15). The strand of the complement is looked up with seedTab[c & 7] for EVERY byte (nthash.hpp:245,
    // 299, 307): 'A' & 7 = 1 -> T, 'C' -> 3 -> G, 'G' -> 7 -> C, 'T' -> 4 -> A -- and a byte that is no base still
    // contributes there when its low bits are one of those ('Y', 'K', 'S', 'W', 'D' ... of the IUPAC codes do; 'N' does not)
    const uint32_t u = i & 0xDF, lo = i & 7;
    const int b = (u == 'A' || i == 4) ? 0 : (u == 'C' || i == 7) ? 1 : (u == 'G' || i == 3) ? 2 : (u == 'T' || i == 1) ? 3 : -1;
    const int bc = lo == 1 ? 3 : lo == 3 ? 2 : lo == 7 ? 1 : lo == 4 ? 0 : -1;
    const uint64_t f = b >= 0 ? sf[b] : 0, c = bc >= 0 ? sf[bc] : 0;
    t->in[i].f = f; t->in[i].r = shk_rol64(c, k - 1);
    t->out[i].f = shk_rol64(f, k); t->out[i].r = shk_ror64(c, 1);
    if (i < 4) {
      t->in4[i].f = sf[i]; t->in4[i].r = shk_rol64(sf[3 - i], k - 1);
      t->out4[i].f = shk_rol64(sf[i], k); t->out4[i].r = shk_ror64(sf[3 - i], 1);
    }
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

// ---- 2-bit staging -------------------------------------------------------------------------------------------------
// The two passes above re-read the FASTQ text thread by thread: a wave's 64 reads lie ~330 bytes apart, every lane pulls
// its own 128-byte lines, and with half a million threads in flight no line survives in L2 until its owner's next round
// (measured: 18 GB fetched per 2.6 GB of text, profiles/r03_a). k_pack_reads reads the text ONCE, coalesced, and leaves
// every read that is all bases as 2 bits per base, 64 bases per 16-byte unit, units of consecutive reads adjacent: the
// passes then take 16 bytes per four rounds from lines their neighbours use too, and a read that holds anything else
// (an 'N', any other byte: rare) stays on the text path above, which treats it exactly as reads_to_kmers does.
// code = A C G T -> 0 1 2 3 (either case): ((c >> 1) ^ (c >> 2)) & 3.
__global__ void k_pack_count(const uint64_t *rd_start, const uint64_t *rd_end, const uint64_t *nreads_p, uint32_t k, uint32_t *units) {
  const uint64_t nreads = *nreads_p;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += stride) {
    const uint64_t len = rd_end[r] - rd_start[r];
    units[r] = (len >= k && len <= 65535) ? (uint32_t)((len + 63) >> 6) : 0;
  }
}
// bytes of v that are zero -> 0x80 (exactly those)
__device__ __forceinline__ uint32_t shk_zero_bytes(uint32_t v) { return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu); }
// four lanes per read, one 64-base unit per lane and turn
__global__ void k_pack_reads(const uint8_t *text, uint64_t safe_end, const uint64_t *rd_start, const uint64_t *rd_end, const uint64_t *nreads_p,
                             const uint64_t *pk_base, uint32_t *flag, ShkQuad *pk, uint64_t cap_units) {
  const uint64_t nreads = *nreads_p;
  const uint64_t stride = ((uint64_t)gridDim.x * blockDim.x) >> 2;
  for (uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; r < nreads; r += stride) {
    const uint32_t units = flag[r] & 0x7FFFFFFFu;
    if (!units) continue;
    const uint64_t base = pk_base[r];
    if (base + units > cap_units) { if ((threadIdx.x & 3) == 0) atomicOr(&flag[r], 0x80000000u); continue; }   // (overlapping chunks: more bases than text)
    const uint64_t st = rd_start[r];
    const uint32_t len = (uint32_t)(rd_end[r] - st);
    uint32_t bad = 0;
    for (uint32_t u = threadIdx.x & 3; u < units; u += 4) {
      uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t pos = 64 * u + 16 * q;
        if (pos < len) {
          const ShkQuad v = shk_load16(text, st + pos, safe_end);
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int b = 0; b < 4; b++) {
            const uint32_t y = w[b] & 0xDFDFDFDFu;
            const uint32_t ok = shk_zero_bytes(y ^ 0x41414141u) | shk_zero_bytes(y ^ 0x43434343u) | shk_zero_bytes(y ^ 0x47474747u) |
                                shk_zero_bytes(y ^ 0x54545454u);
            const uint32_t left = len - pos > 4u * b ? len - pos - 4u * b : 0;      // bytes of this word inside the read
            const uint32_t need = left >= 4 ? 0x80808080u : (0x80808080u & ((1u << (8 * left)) - 1));
            bad |= need & ~ok;
            const uint32_t x = ((w[b] >> 1) ^ (w[b] >> 2)) & 0x03030303u;          // four codes, one per byte
            o[q] |= ((x * 0x01041040u) >> 24) << (8 * b);                          // ... gathered into one byte
          }
        }
      }
      pk[base + u] = shk_quad(o[0], o[1], o[2], o[3]);
    }
    if (bad) atomicOr(&flag[r], 0x80000000u);
  }
}

__device__ __forceinline__ uint32_t shk_pick_word(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t p) {
  return (a & (0u - (uint32_t)(p == 0))) | (b & (0u - (uint32_t)(p == 1))) | (c & (0u - (uint32_t)(p == 2))) | (d & (0u - (uint32_t)(p == 3)));
}
// shk_roll_round for a staged read (`pk` = its first unit; no 'N', so no restart: one (sub)read, fill = min(i, k)).
// A round's 16 incoming bases are one 32-bit word of the unit held in s.i0_; the 16 that leave start 2 * (i - k) bits
// into the read's stream: two neighbouring words, taken from s.o0_ = 16 bytes fetched (at any 4-byte offset) every third
// round. Same register budget as the text form with one quad per stream.
template <typename Emit>
__device__ __forceinline__ bool shk_roll_round_pk(const ShkRollTabs *T, const ShkQuad *pk, ShkRollState &s, uint32_t k, uint64_t mask, Emit emit) {
  if (s.i >= s.len) return false;
  const uint32_t i0 = s.i, ph = (i0 >> 4) & 3;
  if (ph == 0) s.i0_ = pk[i0 >> 6];
  const uint32_t zin = shk_pick_word(s.i0_.x, s.i0_.y, s.i0_.z, s.i0_.w, ph);
  uint64_t fh = s.fh, rh = s.rh;
  if (i0 >= k) {
    const uint32_t n = (i0 >> 4) - ((k + 15) >> 4);         // rounds of this form so far (< 4096)
    const uint32_t n3 = n - 3 * ((n * 0xAAABu) >> 17);      // n mod 3
    const uint32_t ob = 2 * (i0 - k);                       // bit at which the leaving bases start
    if (n3 == 0) __builtin_memcpy(&s.o0_, reinterpret_cast<const uint8_t *>(pk) + 4 * (uint64_t)(ob >> 5), 16);
    const uint32_t a = shk_pick_word(s.o0_.x, s.o0_.y, s.o0_.z, 0, n3), b = shk_pick_word(s.o0_.y, s.o0_.z, s.o0_.w, 0, n3);
    const uint32_t zout = (uint32_t)((((uint64_t)b << 32) | a) >> (ob & 31));
    const auto step = [&](int j, bool whole) {
      const uint32_t oi = (j < 2 ? zin << (4 - 2 * j) : zin >> (2 * j - 4)) & 0x30u;       // code times sizeof(ShkRollRow)
      const uint32_t oo = (j < 2 ? zout << (4 - 2 * j) : zout >> (2 * j - 4)) & 0x30u;
      const ShkRollRow ri = *reinterpret_cast<const ShkRollRow *>(reinterpret_cast<const uint8_t *>(T->in4) + oi);
      const ShkRollRow ro = *reinterpret_cast<const ShkRollRow *>(reinterpret_cast<const uint8_t *>(T->out4) + oo);
      fh = shk_rol1(fh) ^ ri.f ^ ro.f;
      rh = shk_ror1(rh) ^ ri.r ^ ro.r;
      if (whole || i0 + j < s.len) emit(j, (fh < rh ? fh : rh) & mask);
    };
    if (i0 + SHK_ROLL_STEPS <= s.len) {
#pragma unroll
      for (int j = 0; j < SHK_ROLL_STEPS; j++) step(j, true);
    } else {
#pragma unroll
      for (int j = 0; j < SHK_ROLL_STEPS; j++) step(j, false);
    }
  } else if (i0 + SHK_ROLL_STEPS < k) {
    // inside the first window: sixteen bases go in, nothing comes out
#pragma unroll
    for (int j = 0; j < SHK_ROLL_STEPS; j++) {
      const uint32_t oi = (j < 2 ? zin << (4 - 2 * j) : zin >> (2 * j - 4)) & 0x30u;
      const ShkRollRow ri = *reinterpret_cast<const ShkRollRow *>(reinterpret_cast<const uint8_t *>(T->in4) + oi);
      fh = shk_rol1(fh) ^ ri.f;
      rh = shk_ror1(rh) ^ ri.r;
    }
  } else {
    // the round in which the first window fills: the bases that leave in it are among the read's first sixteen
    const uint32_t z0 = (i0 >> 6) == 0 ? s.i0_.x : pk[0].x;
#pragma unroll
    for (int j = 0; j < SHK_ROLL_STEPS; j++) {
      const uint32_t idx = i0 + j;
      const ShkRollRow ri = T->in4[(zin >> (2 * j)) & 3u];
      uint64_t f = shk_rol1(fh) ^ ri.f, r = shk_ror1(rh) ^ ri.r;
      if (idx >= k) {
        const ShkRollRow ro = T->out4[(z0 >> (2 * (idx - k))) & 3u];
        f ^= ro.f; r ^= ro.r;
      }
      fh = f; rh = r;
      if (idx + 1 >= k && idx < s.len) emit(j, (f < r ? f : r) & mask);
    }
  }
  s.fh = fh; s.rh = rh;
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
    const auto count = [&](int, uint64_t key) { atomicAdd(&lh[(((uint32_t)(key >> 16) - qlo16) >> A.hist_shift) & (P - 1)], 1u); };
    if (A.pk && !(A.pk_flag[r] >> 31)) {
      const ShkQuad *up = A.pk + A.pk_base[r];
      while (shk_roll_round_pk(&T, up, s, A.k, mask, count)) {}
    } else {
      while (shk_roll_round<4>(&T, A.text, A.safe_end, s, A.k, mask, count)) {}
    }
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
        if (A.pk && !(A.pk_flag[r] >> 31)) s.st = A.pk_base[r] | (1ULL << 63);     // staged: its first unit instead of its text offset
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
      const auto keep = [&](int j, uint64_t key) {
        w[j] = key | tag;
        atomicAdd(&lh[shk_roll_digit(key, A)], 1u);
        vm |= 1u << j;
      };
      have = (s.st >> 63) ? shk_roll_round_pk(&T, A.pk + (s.st & ~(1ULL << 63)), s, A.k, mask, keep)
                          : shk_roll_round<NQ>(&T, A.text, A.safe_end, s, A.k, mask, keep);
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
