/* TEST INFRASTRUCTURE ONLY -- see cqf_oracle.h.
 *
 * Plain-C restatement of the reference's counting path at bits_per_slot = 8
 * (the only width CQF-deNoise uses for its main filter: hb = qb + 8,
 * src/CQF-deNoise.cpp:161). Each function names the reference lines it follows.
 */
#include "cqf_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ ntHash
 * base/nthash.hpp: seeds :24-28, msTab[c][i] = rol(seed[c], i) (:30-118),
 * complement via c & cpOff (:15): 'A'&7=1->T, 'C'&7=3->G, 'G'&7=7->C, 'T'&7=4->A. */
static const uint64_t SEED_A = 0x3c8bfbb395c60474ULL, SEED_C = 0x3193c18562a02b4cULL,
                      SEED_G = 0x20323ed082572324ULL, SEED_T = 0x295549f54be24456ULL;

static inline uint64_t rol64(uint64_t v, unsigned s) { s &= 63; return s ? (v << s) | (v >> (64 - s)) : v; }
static inline uint64_t ror64(uint64_t v, unsigned s) { s &= 63; return s ? (v >> s) | (v << (64 - s)) : v; }

/* seedTab, nthash.hpp:120-153 (indexes 1,3,4,7 are the "& cpOff" images) */
static uint64_t seed_of(unsigned char c) {
  switch (c) {
    case 'A': case 'a': case 4: return SEED_A;
    case 'C': case 'c': case 7: return SEED_C;
    case 'G': case 'g': case 3: return SEED_G;
    case 'T': case 't': case 1: return SEED_T;
    default: return 0;
  }
}

/* NTPC64(const char*, k, fh, rh), nthash.hpp:295-302 */
void orc_nthash(const char *seq, unsigned k, uint64_t *fh, uint64_t *rh) {
  uint64_t f = 0, r = 0;
  for (unsigned i = 0; i < k; i++) {
    f ^= rol64(seed_of((unsigned char)seq[i]), (k - 1 - i) % 64);
    r ^= rol64(seed_of((unsigned char)seq[i] & 7), i % 64);
  }
  *fh = f; *rh = r;
}
/* NTPC64(out, in, k, fh, rh), nthash.hpp:305-309 */
void orc_nthash_roll(unsigned char out, unsigned char in, unsigned k, uint64_t *fh, uint64_t *rh) {
  *fh = rol64(*fh, 1) ^ rol64(seed_of(out), k % 64) ^ seed_of(in);
  *rh = ror64(*rh, 1) ^ rol64(seed_of(out & 7), 63) ^ rol64(seed_of(in & 7), (k - 1) % 64);
}

/* ------------------------------------------------------------------ block access
 * qfblock, gqf.c:63-86: offset u8 | occupieds u64 | runends u64 | traveled u64 | slots[64] */
#define OFF(qf, b) ((qf)->blocks[(b) * ORC_BLOCK_BYTES])
static inline uint64_t ld64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline void st64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }
static inline uint64_t occw(const orc_qf *qf, uint64_t b) { return ld64(qf->blocks + b * ORC_BLOCK_BYTES + 1); }
static inline uint64_t runw(const orc_qf *qf, uint64_t b) { return ld64(qf->blocks + b * ORC_BLOCK_BYTES + 9); }
static inline uint64_t travw(const orc_qf *qf, uint64_t b) { return ld64(qf->blocks + b * ORC_BLOCK_BYTES + 17); }
static inline void set_occw(orc_qf *qf, uint64_t b, uint64_t v) { st64(qf->blocks + b * ORC_BLOCK_BYTES + 1, v); }
static inline void set_runw(orc_qf *qf, uint64_t b, uint64_t v) { st64(qf->blocks + b * ORC_BLOCK_BYTES + 9, v); }
static inline void set_travw(orc_qf *qf, uint64_t b, uint64_t v) { st64(qf->blocks + b * ORC_BLOCK_BYTES + 17, v); }
static inline uint8_t *slotp(const orc_qf *qf, uint64_t i) { return qf->blocks + (i / 64) * ORC_BLOCK_BYTES + 25 + (i % 64); }
static inline uint64_t get_slot(const orc_qf *qf, uint64_t i) { return *slotp(qf, i); }          /* gqf.c:542 */
/* writes behind the table happen only once an insert has found it full (qf->full is set; the result is void and the
 * tests discard it) -- the reference would overrun here; the oracle must stay inside its allocation */
static inline void set_slot(orc_qf *qf, uint64_t i, uint64_t v) { if (i / 64 <= qf->nblocks) *slotp(qf, i) = (uint8_t)v; } /* gqf.c:556 */
static inline int is_runend(const orc_qf *qf, uint64_t i) { return (runw(qf, i / 64) >> (i % 64)) & 1; }   /* :474 */
static inline int is_occupied(const orc_qf *qf, uint64_t i) { return (occw(qf, i / 64) >> (i % 64)) & 1; } /* :480 */
static inline void set_runend(orc_qf *qf, uint64_t i, int v) {
  if (i / 64 > qf->nblocks) return;
  uint64_t w = runw(qf, i / 64);
  if (v) w |= 1ULL << (i % 64); else w &= ~(1ULL << (i % 64));
  set_runw(qf, i / 64, w);
}
static inline void set_occupied(orc_qf *qf, uint64_t i, int v) {
  uint64_t w = occw(qf, i / 64);
  if (v) w |= 1ULL << (i % 64); else w &= ~(1ULL << (i % 64));
  set_occw(qf, i / 64, w);
}

#define BITMASK(n) ((n) == 64 ? 0xffffffffffffffffULL : (1ULL << (n)) - 1ULL)
static inline int popcnt(uint64_t v) { return __builtin_popcountll(v); }
static inline int popcntv(uint64_t v, int ignore) { return (ignore % 64) ? popcnt(v & ~BITMASK(ignore % 64)) : popcnt(v); } /* :300 */
static inline int bitrank(uint64_t v, int pos) { return popcnt(v & ((2ULL << pos) - 1)); }  /* :310 */
/* position of the rank-th (0-based) one, 64 if none: gqf.c:336-459 */
static inline uint64_t bitselect(uint64_t v, int rank) {
  for (int i = 0; i < rank; i++) v &= v - 1;
  return v ? (uint64_t)__builtin_ctzll(v) : 64;
}
static inline uint64_t bitselectv(uint64_t v, int ignore, int rank) { return bitselect(v & ~BITMASK(ignore % 64), rank); }

/* gqf.c:655-704 with the block's offset handed in */
static uint64_t run_end_at(const orc_qf *qf, uint64_t q, uint64_t boff) {
  uint64_t bi = q / 64, off = q % 64;
  uint64_t rank = bitrank(occw(qf, bi), off);
  if (rank == 0) {
    if (boff <= off) return q;
    return 64 * bi + boff - 1;
  }
  uint64_t rb = bi + boff / 64;
  uint64_t ignore = boff % 64;
  uint64_t rrank = rank - 1;
  uint64_t ro = bitselectv(runw(qf, rb), ignore, rrank);
  if (ro == 64) {
    if (boff == 0 && rank == 0) return q;
    do {
      rrank -= popcntv(runw(qf, rb), ignore);
      rb++;
      ignore = 0;
      ro = bitselectv(runw(qf, rb), ignore, rrank);
    } while (ro == 64);
  }
  uint64_t ri = 64 * rb + ro;
  return ri < q ? q : ri;
}

/* gqf.c:599-601: run_end(64 b - 1) - 64 b + 1. The reference evaluates the saturated case (stored offset 255) by
 * mutual recursion block_offset <-> run_end, one level per saturated block in front of b; on an over-full table that
 * chain is the whole table and the recursion overruns the stack (round-2 fuzz, qb 24). Same function, as a loop: go
 * back to the nearest block whose stored offset is exact (block 0's always is), then forward. */
static uint64_t block_offset_strict(const orc_qf *qf, uint64_t b) {
  if (b == 0) return 0; /* never asked for: block 0's offset is 0 by construction */
  uint64_t b0 = b - 1;
  while (b0 > 0 && OFF(qf, b0) >= 255) b0--;
  uint64_t off = b0 == 0 ? (OFF(qf, 0) < 255 ? OFF(qf, 0) : 0) : OFF(qf, b0);
  for (uint64_t bi = b0 + 1; bi <= b; bi++) off = run_end_at(qf, 64 * bi - 1, off) - 64 * bi + 1;
  return off;
}
/* gqf.c:580-591 */
static uint64_t block_offset(const orc_qf *qf, uint64_t b) {
  if (OFF(qf, b) < 255) return OFF(qf, b);
  return block_offset_strict(qf, b);
}
/* gqf.c:655-704 */
static uint64_t run_end(const orc_qf *qf, uint64_t q) { return run_end_at(qf, q, block_offset(qf, q / 64)); }

/* gqf.c:706-718 */
static int offset_lower_bound(const orc_qf *qf, uint64_t slot) {
  uint64_t b = slot / 64, so = slot % 64;
  uint64_t boff = OFF(qf, b);
  uint64_t occ = occw(qf, b) & BITMASK(so + 1);
  if (boff <= so) {
    uint64_t re = (runw(qf, b) & BITMASK(so)) >> boff;
    return popcnt(occ) - popcnt(re);
  }
  return (int)(boff - so) + popcnt(occ);
}
/* gqf.c:738-748 */
uint64_t orc_find_first_empty_slot(const orc_qf *qf, uint64_t from) {
  for (;;) {
    int t = offset_lower_bound(qf, from);
    if (t == 0) break;
    from += t;
  }
  return from;
}
/* gqf.c:751-774 */
uint64_t orc_find_first_nonempty_slot(const orc_qf *qf, uint64_t from) {
  if (is_occupied(qf, from)) return from;
  uint64_t bi = from / 64;
  uint64_t rank = bitrank(occw(qf, bi), from % 64);
  uint64_t next = bitselect(occw(qf, bi), rank);
  if (next == 64) {
    rank = 0;
    while (next == 64 && bi < qf->nblocks) {
      bi++;
      next = bitselect(occw(qf, bi), rank); /* reads the guard block at bi == nblocks */
    }
  }
  next = bi * 64 + next;
  if (bi > qf->nblocks || next >= qf->xnslots) return qf->xnslots;
  return next;
}

/* ------------------------------------------------------------------ counters
 * encode_counter, gqf.c:1225-1255 (base 128 digits of count-1, high bit on all but
 * the last digit, escape 0 when the leading digit exceeds the remainder) */
int orc_encode_counter(uint64_t rem, uint64_t counter, uint64_t *out) {
  uint64_t tmp[16]; int n = 0;
  if (counter == 0) return 0;
  if (counter == 1) { out[0] = rem; return 1; }
  counter--;
  uint64_t digit = counter % 128;
  counter /= 128;
  tmp[n++] = digit;
  while (counter) {
    digit = (counter % 128) | 0x80;
    tmp[n++] = digit;
    counter /= 128;
  }
  int m = 0;
  out[m++] = rem;
  if (digit > rem) out[m++] = 0;
  for (int i = n - 1; i >= 0; i--) out[m++] = tmp[i];
  return m;
}
/* decode_counter, gqf.c:1259-1299: returns index of the last slot of the entry */
static uint64_t decode_counter(const orc_qf *qf, uint64_t index, uint64_t *remainder, uint64_t *count) {
  uint64_t rem = get_slot(qf, index);
  *remainder = rem;
  if (is_runend(qf, index)) { *count = 1; return index; }
  uint64_t digit = get_slot(qf, index + 1);
  if (digit > rem) { *count = 1; return index; }
  uint64_t cnt = 0, end = index + 1;
  if (digit == 0) { end++; digit = get_slot(qf, end); }
  while (digit & 0x80) { cnt = cnt * 128 + (digit & 0x7f); end++; digit = get_slot(qf, end); }
  cnt = cnt * 128 + digit;
  *count = cnt + 1;
  return end;
}

/* ------------------------------------------------------------------ init / io
 * qf_init, gqf.c:2187-2290 */
orc_qf *orc_qf_new(uint64_t qb, uint64_t hb, uint32_t seed) {
  if (hb != qb + 8) return NULL; /* only the 8-bit-remainder geometry is restated */
  orc_qf *qf = (orc_qf *)calloc(1, sizeof(orc_qf));
  qf->qb = qb; qf->hb = hb; qf->seed = seed;
  qf->nslots = 1ULL << qb;
  qf->xnslots = qf->nslots + (uint64_t)(10 * sqrt((double)qf->nslots));
  qf->nblocks = (qf->xnslots + 63) / 64;
  qf->size = qf->nblocks * ORC_BLOCK_BYTES;
  qf->blocks = (uint8_t *)calloc(qf->size + 2 * ORC_BLOCK_BYTES, 1);
  return qf;
}
void orc_qf_free(orc_qf *qf) { if (qf) { free(qf->blocks); free(qf); } }
const uint8_t *orc_qf_blocks(const orc_qf *qf) { return qf->blocks; }
uint64_t orc_qf_size(const orc_qf *qf) { return qf->size; }
uint64_t orc_qf_nelts(const orc_qf *qf) { return qf->nelts; }
uint64_t orc_qf_ndistinct(const orc_qf *qf) { return qf->ndistinct; }
int orc_qf_full(const orc_qf *qf) { return qf->full; }

/* quotient_filter_metadata image, gqf.h:62-77 (offsets verified against the compiled
 * reference: size 0, seed 8, nslots 16, xnslots 24, key_bits 32, value_bits 40,
 * key_remainder_bits 48, bits_per_slot 56, range 64 (u128), nblocks 80, nelts 88,
 * ndistinct_elts 96, noccupied_slots 104, num_locks 112, sizeof 128) */
void orc_qf_header(const orc_qf *qf, uint8_t out[128]) {
  memset(out, 0, 128);
  uint64_t v;
#define PUT(off, val) do { v = (val); memcpy(out + (off), &v, 8); } while (0)
  PUT(0, qf->size);
  memcpy(out + 8, &qf->seed, 4);
  PUT(16, qf->nslots); PUT(24, qf->xnslots); PUT(32, qf->hb); PUT(40, 0);
  PUT(48, qf->hb - qf->qb); PUT(56, qf->hb - qf->qb);
  { unsigned __int128 range = (unsigned __int128)qf->nslots << (qf->hb - qf->qb); memcpy(out + 64, &range, 16); }
  PUT(80, qf->nblocks); PUT(88, qf->nelts); PUT(96, qf->ndistinct); PUT(104, 0);
  PUT(112, qf->xnslots / (1ULL << 16) + 2);
#undef PUT
}
/* qf_serialize, gqf.c:2379-2394 (+ CQF_mt.h:986-987 counter patch) */
int orc_qf_serialize(const orc_qf *qf, const char *path) {
  FILE *f = fopen(path, "wb+");
  if (!f) return -1;
  uint8_t hdr[128];
  orc_qf_header(qf, hdr);
  fwrite(hdr, 128, 1, f);
  fwrite(qf->blocks, qf->size, 1, f);
  fclose(f);
  return 0;
}
/* qf_deserialize, gqf.c:2396-2420 */
orc_qf *orc_qf_load(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  uint8_t hdr[128];
  if (fread(hdr, 128, 1, f) != 1) { fclose(f); return NULL; }
  uint64_t nslots, key_bits, bps; uint32_t seed;
  memcpy(&nslots, hdr + 16, 8); memcpy(&key_bits, hdr + 32, 8); memcpy(&bps, hdr + 56, 8); memcpy(&seed, hdr + 8, 4);
  uint64_t qb = 0; while ((1ULL << qb) < nslots) qb++;
  if (bps != 8 || key_bits != qb + 8) { fclose(f); return NULL; }
  orc_qf *qf = orc_qf_new(qb, key_bits, seed);
  if (fread(qf->blocks, qf->size, 1, f) != 1) { fclose(f); orc_qf_free(qf); return NULL; }
  memcpy(&qf->nelts, hdr + 88, 8); memcpy(&qf->ndistinct, hdr + 96, 8);
  fclose(f);
  return qf;
}

/* ------------------------------------------------------------------ insert
 * One-slot insertion step shared by every case of insert1_advance / insert_advance
 * (gqf.c:1614-1915, 2024-2136): find_first_empty_slot (:738), shift_remainders (:868),
 * shift_runends (:986), and `offset++` saturating at 255 on blocks home+1..empty
 * (:1695-1700). `pos` is where the new slot goes; slots [pos, empty) move right by one. */
static void insert_one_slot(orc_qf *qf, uint64_t q, uint64_t pos, uint64_t value) {
  uint64_t e = orc_find_first_empty_slot(qf, pos);
  if (e >= qf->xnslots) { qf->full = 1; return; } /* the reference runs off the end here (undetected) */
  /* a shift over more than 2^20 slots: one cluster 64 times the CLUSTER_SIZE = 2^14 the reference's region locks assume
   * (gqf.c:53, 174-199) -- the table is over-full for every purpose and each further insert would move megabytes;
   * treated like the overrun (full is sticky, the tests discard such a table) */
  if (e - pos > (1ULL << 20)) { qf->full = 1; return; }
  for (uint64_t i = e; i > pos; i--) {
    set_slot(qf, i, get_slot(qf, i - 1));
    set_runend(qf, i, is_runend(qf, i - 1));
  }
  set_runend(qf, pos, 0);
  set_slot(qf, pos, value);
  for (uint64_t b = q / 64 + 1; b <= e / 64; b++)
    if (OFF(qf, b) < 255) OFF(qf, b)++;
}

/* qf_insert_advance, gqf.c:2432-2440: increments the counter of `key` by `count`.
 * Case analysis follows insert1_advance (:1627-1907): empty home slot, new run for an
 * occupied-by-shift home, new largest / inner remainder, existing remainder. */
int orc_qf_insert(orc_qf *qf, uint64_t key, uint64_t count) {
  uint64_t r = key & 0xff, q = key >> 8;
  uint64_t enc[16];
  int isNew;
  if (count == 0 || qf->full) return 0;
  if (!is_occupied(qf, q)) {
    /* :1627-1637 / :1887-1907: the run starts right after the previous run's end */
    uint64_t pos = q == 0 ? 0 : run_end(qf, q - 1) + 1;
    if (pos < q) pos = q;
    int n = orc_encode_counter(r, count, enc);
    for (int i = 0; i < n; i++) {
      insert_one_slot(qf, q, pos + i, enc[i]);
      if (i > 0) set_runend(qf, pos + i - 1, 0);
      set_runend(qf, pos + i, 1);
      if (i == 0) set_occupied(qf, q, 1);
    }
    isNew = 1;
  } else {
    uint64_t re = run_end(qf, q);
    uint64_t rs = q == 0 ? 0 : run_end(qf, q - 1) + 1;
    if (rs < q) rs = q;
    /* walk the run (:1653-1675) */
    uint64_t cur = rs, crem = 0, ccnt = 0, cend = 0;
    int found = 0, past_end = 0;
    for (;;) {
      cend = decode_counter(qf, cur, &crem, &ccnt);
      if (crem >= r) { found = (crem == r); break; }
      if (cend == re) { past_end = 1; break; }
      cur = cend + 1;
    }
    if (past_end) {
      /* new largest remainder (:1681-1701): append after the run end */
      int n = orc_encode_counter(r, count, enc);
      for (int i = 0; i < n; i++) {
        insert_one_slot(qf, q, re + 1 + i, enc[i]);
        set_runend(qf, re + i, 0);
        set_runend(qf, re + 1 + i, 1);
      }
      isNew = 1;
    } else if (!found) {
      /* new remainder inside the run (:1870-1885) */
      int n = orc_encode_counter(r, count, enc);
      for (int i = 0; i < n; i++) insert_one_slot(qf, q, cur + i, enc[i]);
      isNew = 0 + 1;
    } else {
      /* existing remainder (:1704-1869): re-encode count+old in place, growing the
       * entry by inserting slots behind its last slot */
      int oldn = (int)(cend - cur + 1);
      int n = orc_encode_counter(r, ccnt + count, enc);
      for (int i = oldn; i < n; i++) {
        uint64_t p = cur + i;
        int at_end = (p - 1 == re);
        insert_one_slot(qf, q, p, 0);
        if (at_end) { set_runend(qf, re, 0); set_runend(qf, p, 1); }
        re++;
      }
      for (int i = 0; i < n; i++) set_slot(qf, cur + i, enc[i]);
      isNew = 0;
    }
  }
  if (isNew) qf->ndistinct++;
  qf->nelts += count;
  return isNew;
}

/* ------------------------------------------------------------------ lookups
 * qf_count_key_value, gqf.c:2442-2469 */
static int find_entry(const orc_qf *qf, uint64_t key, uint64_t *start, uint64_t *count) {
  uint64_t r = key & 0xff, q = key >> 8;
  if (!is_occupied(qf, q)) return 0;
  uint64_t rs = q == 0 ? 0 : run_end(qf, q - 1) + 1;
  if (rs < q) rs = q;
  uint64_t crem, ccnt, cend;
  do {
    cend = decode_counter(qf, rs, &crem, &ccnt);
    if (crem == r) { *start = rs; *count = ccnt; return 1; }
    rs = cend + 1;
  } while (!is_runend(qf, cend));
  return 0;
}
uint64_t orc_qf_count(const orc_qf *qf, uint64_t key) {
  uint64_t s, c;
  return find_entry(qf, key, &s, &c) ? c : 0;
}
/* qf_count_key_value_set_traveled, gqf.c:3092-3128: traveled bit of the entry's first slot */
int orc_qf_count_set_traveled(orc_qf *qf, uint64_t key, uint64_t *count) {
  uint64_t s, c;
  if (!find_entry(qf, key, &s, &c)) { *count = 0; return 0; }
  *count = c;
  uint64_t w = travw(qf, s / 64);
  if ((w >> (s % 64)) & 1) return 1;
  set_travw(qf, s / 64, w | (1ULL << (s % 64)));
  return 0;
}
/* qf_count_key_value_is_traveled, gqf.c:3132-3163 */
int orc_qf_count_is_traveled(const orc_qf *qf, uint64_t key, uint64_t *count) {
  uint64_t s, c;
  if (!find_entry(qf, key, &s, &c)) { *count = 0; return 0; }
  *count = c;
  return (int)((travw(qf, s / 64) >> (s % 64)) & 1);
}

/* (key,count) walk in table order; what qf_iterator/qfi_get/qfi_next (gqf.c:2474-2601) yield */
uint64_t orc_qf_dump(const orc_qf *qf, uint64_t *keys, uint64_t *counts, uint64_t cap) {
  uint64_t n = 0, freep = 0;
  for (uint64_t b = 0; b < qf->nblocks; b++) {
    uint64_t w = occw(qf, b);
    while (w) {
      uint64_t q = b * 64 + (uint64_t)__builtin_ctzll(w);
      w &= w - 1;
      uint64_t cur = freep > q ? freep : q;
      uint64_t crem, ccnt, cend;
      do {
        cend = decode_counter(qf, cur, &crem, &ccnt);
        if (n < cap) { keys[n] = (q << 8) | crem; counts[n] = ccnt; }
        n++;
        cur = cend + 1;
      } while (!is_runend(qf, cend));
      freep = cend + 1;
    }
  }
  return n;
}

/* check_offset, gqf.c:3056-3069 */
int orc_qf_check_offset(const orc_qf *qf) {
  for (uint64_t x = 1; x < qf->nblocks; x++) {
    uint64_t real = block_offset_strict(qf, x);
    if (real <= 255 && OFF(qf, x) != real) return 0;
  }
  return 1;
}

/* ------------------------------------------------------------------ deNoise
 * qf_clean_singleton, gqf.c:2792-2876: one cluster [start_bucket_id, end_bucket_id] */
static void clean_singleton(orc_qf *qf, uint64_t start_bucket_id, uint64_t end_bucket_id, uint64_t *removed) {
  uint64_t last_empty_slot, bucket_idx, run_start, run_e, insert_idx;
  last_empty_slot = bucket_idx = run_start = insert_idx = start_bucket_id;
  uint64_t remainder, tmp;
  while (bucket_idx <= end_bucket_id) {
    last_empty_slot = insert_idx;
    run_e = run_start;
    for (;;) {
      if (is_runend(qf, run_e)) { (*removed)++; break; }
      remainder = get_slot(qf, run_e);
      tmp = get_slot(qf, ++run_e);
      if (remainder >= tmp) {
        if (insert_idx + 1 == run_e) {
          if (tmp == 0) run_e++;
          tmp = get_slot(qf, run_e);
          while (tmp & 0x80) tmp = get_slot(qf, ++run_e);
          insert_idx = run_e + 1;
        } else {
          set_slot(qf, insert_idx++, remainder);
          if (tmp == 0) { set_slot(qf, insert_idx++, 0); run_e++; }
          tmp = get_slot(qf, run_e);
          while (tmp & 0x80) { set_slot(qf, insert_idx++, tmp); tmp = get_slot(qf, ++run_e); }
          set_slot(qf, insert_idx++, tmp);
        }
        if (is_runend(qf, run_e)) break;
        run_e++;
      } else {
        (*removed)++;
      }
    }
    if (last_empty_slot == insert_idx) {
      set_occupied(qf, bucket_idx, 0);
      set_runend(qf, run_e, 0);
    } else if (run_e + 1 != insert_idx) {
      set_runend(qf, run_e, 0);
      set_runend(qf, insert_idx - 1, 1);
    }
    run_start = run_e + 1;
    bucket_idx++;
    while (!is_occupied(qf, bucket_idx) && bucket_idx <= end_bucket_id) bucket_idx++;
    while (insert_idx < bucket_idx) { set_slot(qf, insert_idx, 0); insert_idx++; }
  }
  for (uint64_t b = start_bucket_id / 64 + 1; b <= end_bucket_id / 64; b++) {
    uint64_t off = block_offset_strict(qf, b);
    OFF(qf, b) = off > 255 ? 255 : (uint8_t)off;
  }
}
/* qf_clean_singleton_discrete, gqf.c:2878-2886 */
static void clean_discrete(orc_qf *qf, uint64_t start_bucket_id, uint64_t end_bucket_id, uint64_t *removed) {
  uint64_t start = start_bucket_id, end;
  while (start < end_bucket_id) {
    end = orc_find_first_empty_slot(qf, start + 1) - 1;
    clean_singleton(qf, start, end, removed);
    start = orc_find_first_nonempty_slot(qf, end + 1);
  }
}
/* qf_clean_singleton_with_lock, CQF_mt.h:999-1039 (one thread: the _atStart variant,
 * gqf.c:2888-3040, differs from qf_clean_singleton only in when it takes locks) */
static void clean_with_lock(orc_qf *qf, uint64_t start_bucket_id, uint64_t end_bucket_id) {
  uint64_t start, end, sb = start_bucket_id / 64, eb = end_bucket_id / 64, removed = 0;
  if (sb == eb) {
    clean_discrete(qf, start_bucket_id, end_bucket_id, &removed);
  } else {
    start = start_bucket_id;
    end = orc_find_first_empty_slot(qf, start + 1) - 1;
    while ((end / 64) == sb) {
      clean_singleton(qf, start, end, &removed);
      start = orc_find_first_nonempty_slot(qf, end + 1);
      end = orc_find_first_empty_slot(qf, start + 1) - 1;
    }
    if (start / 64 == sb) {
      clean_singleton(qf, start, end, &removed);
      start = orc_find_first_nonempty_slot(qf, end + 1);
      end = orc_find_first_empty_slot(qf, start + 1) - 1;
    }
    while (start < end_bucket_id) {
      if (end / 64 == eb) { clean_discrete(qf, start, end_bucket_id, &removed); break; }
      clean_singleton(qf, start, end, &removed);
      start = orc_find_first_nonempty_slot(qf, end + 1);
      end = orc_find_first_empty_slot(qf, start + 1) - 1;
    }
  }
  qf->nelts -= removed;
  qf->ndistinct -= removed;
}
/* one DeNoise phase for one thread, CQF_mt.h:866, 884-901 */
uint64_t orc_denoise_round_t1(orc_qf *qf, uint64_t min_len) {
  uint64_t before = qf->ndistinct;
  uint64_t cur = orc_find_first_nonempty_slot(qf, 0);
  while (cur < qf->nslots) {
    uint64_t start = cur, end;
    end = (start + min_len) > qf->nslots ? qf->nslots : (start + min_len);
    end = orc_find_first_empty_slot(qf, end) - 1;
    cur = orc_find_first_nonempty_slot(qf, end + 1);
    clean_with_lock(qf, start, end);
  }
  return before - qf->ndistinct;
}

/* ------------------------------------------------------------------ reads -> k-mers
 * reads_to_kmers, CQF_mt.h:610-731. `emit` receives min(fh,rh) % range per k-mer. */
typedef void (*emit_fn)(void *ctx, uint64_t key);
static void walk_chunk(const char *chunk, uint64_t size, unsigned k, uint64_t hb, emit_fn emit, void *ctx) {
  const char *fs = chunk, *fe, *end = chunk + size;
  uint64_t mask = hb >= 64 ? ~0ULL : ((1ULL << hb) - 1);
  while (fs && fs != end) {
    fs = (const char *)memchr(fs, '\n', end - fs); if (!fs) break; /* reference would crash here */
    fs++;
    fe = (const char *)memchr(fs, '\n', end - fs); if (!fe) break;
    const char *read = fs; uint64_t len = (uint64_t)(fe - fs);
    uint64_t fh, rh;
    for (;;) { /* start_read: */
      if (len < k) break;
      orc_nthash(read, k, &fh, &rh);
      emit(ctx, (fh < rh ? fh : rh) & mask);
      uint64_t i; int restart = 0;
      for (i = k; i < len; i++) {
        if (read[i] == 'N') { read += i + 1; len -= i + 1; restart = 1; break; }
        orc_nthash_roll((unsigned char)read[i - k], (unsigned char)read[i], k, &fh, &rh);
        emit(ctx, (fh < rh ? fh : rh) & mask);
      }
      if (!restart) break;
    }
    fs = fe + 1;
    fs = (const char *)memchr(fs, '\n', end - fs); if (!fs) break;
    fs++;
    fs = (const char *)memchr(fs, '\n', end - fs); if (!fs) break;
    fs++;
  }
}
static void emit_insert(void *ctx, uint64_t key) { orc_qf_insert((orc_qf *)ctx, key, 1); }
void orc_reads_to_kmers(orc_qf *qf, const char *chunk, uint64_t size, unsigned k) {
  walk_chunk(chunk, size, k, qf->hb, emit_insert, qf);
}
struct keybuf { uint64_t *keys; uint64_t cap, n; };
static void emit_store(void *ctx, uint64_t key) {
  struct keybuf *b = (struct keybuf *)ctx;
  if (b->n < b->cap) b->keys[b->n] = key;
  b->n++;
}
uint64_t orc_chunk_keys(const char *chunk, uint64_t size, unsigned k, uint64_t hb, uint64_t *keys, uint64_t cap) {
  struct keybuf b = {keys, cap, 0};
  walk_chunk(chunk, size, k, hb, emit_store, &b);
  return b.n;
}

/* ------------------------------------------------------------------ chunker
 * skip_next_eol, CQF_mt.h:573-585 */
static int skip_next_eol(const char *part, int64_t *pos, int64_t max_pos) {
  int64_t i;
  for (i = *pos; i < max_pos - 2; ++i)
    if ((part[i] == '\n' || part[i] == '\r') && !(part[i + 1] == '\n' || part[i + 1] == '\r')) break;
  if (i >= max_pos - 2) return 0;
  *pos = i + 1;
  return 1;
}
typedef struct { FILE *in; char *carry; uint64_t filled; } orc_file;
/* fastq_read_parts, CQF_mt.h:735-816 (plain text). Returns 0 at end of file, -1 on
 * "Wrong input file"; the chunk is malloc'ed into *out / *out_size. */
static int read_part(orc_file *fp, uint64_t part_size, uint32_t overhead, char **out, uint64_t *out_size) {
  char *part = (char *)malloc(part_size + overhead);
  memcpy(part, fp->carry, fp->filled);
  if (feof(fp->in)) { free(part); return 0; }
  uint64_t readed = fread(part + fp->filled, 1, part_size, fp->in);
  int64_t total = (int64_t)(fp->filled + readed);
  if (fp->filled >= overhead) { free(part); return -1; }
  if (feof(fp->in)) { *out = part; *out_size = (uint64_t)total; return 1; }
  uint64_t size;
  int64_t ls[9]; int j; int64_t i = total - overhead / 2;
  for (j = 0; j < 9; ++j) { if (!skip_next_eol(part, &i, total)) break; ls[j] = i; }
  if (j < 9) size = 0;
  else {
    int k;
    for (k = 0; k < 4; ++k) {
      if (part[ls[k]] == '@' && part[ls[k + 2]] == '+') {
        if (part[ls[k + 2] + 1] == '\n' || part[ls[k + 2] + 1] == '\r') break;
        if (ls[k + 1] - ls[k] == ls[k + 3] - ls[k + 2] &&
            memcmp(part + ls[k] + 1, part + ls[k + 2] + 1, (size_t)(ls[k + 3] - ls[k + 2] - 1)) == 0) break;
      }
    }
    size = (k == 4) ? 0 : (uint64_t)ls[k];
  }
  memcpy(fp->carry, part + size, (size_t)(total - (int64_t)size));
  fp->filled = (uint64_t)total - size;
  *out = part; *out_size = size;
  return 1;
}
uint64_t orc_chunk_sizes(const char *path, uint64_t part_size, uint32_t overhead, uint64_t *sizes, uint64_t cap) {
  orc_file f; f.in = fopen(path, "rb"); if (!f.in) return 0;
  f.carry = (char *)malloc(part_size + overhead); f.filled = 0;
  uint64_t n = 0; char *c; uint64_t sz;
  while (read_part(&f, part_size, overhead, &c, &sz) == 1) { if (n < cap) sizes[n] = sz; n++; free(c); }
  fclose(f.in); free(f.carry);
  return n;
}

/* ExtractKmer / DeNoise / Idle with one worker, CQF_mt.h:821-931; files are served
 * round-robin, one part each (:828-830). stats: rounds, removed, chunks. */
void orc_build_t1(orc_qf *qf, const char **files, int nfiles, unsigned k,
                  uint64_t ndistinct_for_denoise, uint32_t num_denoise, int end_denoise,
                  uint64_t part_size, uint32_t overhead, uint64_t min_denoise_len, uint64_t *stats) {
  orc_file **queue = (orc_file **)calloc((size_t)nfiles + 1, sizeof(orc_file *));
  int head = 0, qn = 0;
  for (int i = 0; i < nfiles; i++) {
    FILE *in = fopen(files[i], "rb");
    if (!in) continue;
    orc_file *f = (orc_file *)calloc(1, sizeof(orc_file));
    f->in = in; f->carry = (char *)malloc(part_size + overhead);
    queue[qn++] = f;
  }
  int num_files = qn;
  stats[0] = stats[1] = stats[2] = 0;
  int mode = 0; /* 0 ExtractKmer, 1 DeNoise, 2 Idle */
  while (mode != 2) {
    if (mode == 0) {
      while (num_files) {
        orc_file *fp = queue[head];
        /* pop front */
        for (int i = 0; i + 1 < num_files; i++) queue[i] = queue[i + 1];
        char *c; uint64_t sz;
        int rc = read_part(fp, part_size, overhead, &c, &sz);
        if (rc == 1) {
          queue[num_files - 1] = fp; /* push back */
          orc_reads_to_kmers(qf, c, sz, k);
          free(c);
          stats[2]++;
          if (num_denoise && qf->ndistinct >= ndistinct_for_denoise) break;
        } else {
          fclose(fp->in); free(fp->carry); free(fp); num_files--;
        }
      }
      if ((num_denoise && qf->ndistinct >= ndistinct_for_denoise) || (end_denoise && !num_files)) {
        if (num_denoise) num_denoise--;
        mode = 1;
      } else if (!num_files) mode = 2;
    } else {
      stats[1] += orc_denoise_round_t1(qf, min_denoise_len);
      stats[0]++;
      mode = num_files ? 0 : 2;
    }
  }
  free(queue);
}

/* ------------------------------------------------------------------ sizing
 * Poisson CDF (boost::math::cdf(poisson) in the reference, CQF_mt.h:79-101): summed pmf. */
static double poisson_cdf(double mean, double x) {
  if (x < 0) return 0;
  long kmax = (long)floor(x);
  long double s = 0;
  for (long i = 0; i <= kmax; i++) s += expl(-(long double)mean + i * logl((long double)mean) - lgammal((long double)i + 1));
  return (double)(s > 1 ? 1 : s);
}
/* mean_CDF2deNoise, CQF_mt.h:94-133 */
int orc_mean_cdf2denoise(double mean, double cdf_desired) {
  int start = 0, end = (int)(mean + 1), mid;
  double cdf0 = poisson_cdf(mean, 0);
#define CDFP(x) ((poisson_cdf(mean, (x)) - cdf0) / (1 - cdf0))
  while (CDFP(end) < cdf_desired) end *= 2;
  while (start <= end) {
    if (start == end) return start;
    else if (start + 1 == end) {
      double t1 = CDFP(start), t2 = CDFP(end);
      if (t2 <= cdf_desired) return end;
      else if (t1 <= cdf_desired) return start;
      else return start - 1 > 0 ? start - 1 : 0;
    }
    mid = (start + end) / 2;
    double cdf = CDFP(mid);
    if (cdf < cdf_desired) start = mid + 1;
    else if (cdf > cdf_desired) end = mid - 1;
    else return start;
  }
#undef CDFP
  return start;
}
/* true2falseKmer_DP, cqf/true2falseKmer_DP.cpp:12-50 */
double orc_true2false_dp(const double *e, size_t seq_len, size_t K) {
  double *DP = (double *)calloc(K + 1, sizeof(double)), *nDP = (double *)calloc(K + 1, sizeof(double));
  double tmp = 1, trueP;
  for (size_t x = 0; x < K; x++) tmp *= (1 - e[x]);
  DP[0] = tmp;
  for (size_t x = 1; x <= K; x++) {
    tmp = e[x - 1];
    for (size_t y = x; y < K; y++) tmp *= (1 - e[y]);
    DP[x] = tmp;
  }
  trueP = DP[0];
  for (size_t x = K; x < seq_len; x++) {
    nDP[0] = DP[0] * (1 - e[x]);
    for (size_t y = 1; y <= K; y++) nDP[y - 1] += DP[y] * (1 - e[x]);
    nDP[K] = e[x];
    trueP += nDP[0];
    memcpy(DP, nDP, (K + 1) * sizeof(double));
    memset(nDP, 0, (K + 1) * sizeof(double));
  }
  free(DP); free(nDP);
  return trueP / ((double)(seq_len - K + 1) - trueP);
}
/* src/CQF-deNoise.cpp:96-161. alpha < 0 selects the error-profile ratio `true2false`. */
void orc_size_filter(int K, uint64_t n_true, uint64_t N_total, double alpha, double true2false,
                     int num_denoise_opt, double fr, orc_sizing *o) {
  uint64_t num_true, num_false, num_slots;
  if (alpha == -1) num_true = (uint64_t)(N_total * true2false / (1 + true2false));
  else num_true = (uint64_t)(N_total * pow(1 - alpha, K));
  num_false = N_total - num_true;
  int nd = num_denoise_opt;
  if (nd < 0) {
    if (!fr) fr = 1.0 / n_true;
    nd = orc_mean_cdf2denoise((double)(num_true / n_true), fr);
  }
  int enc = 0;
  uint64_t tmp = num_true / n_true + 1;
  while (tmp) { tmp >>= 7; enc++; }
#define NSLOTS(d) ((uint64_t)(n_true * (enc + (double)3 / 2) + num_false * 10 / (((uint64_t)(d) + 1) * 9)))
  num_slots = NSLOTS(nd);
  uint64_t qb = 1, base = 2;
  while (base < num_slots) { qb++; base <<= 1; }
  int ub, lb;
  ub = lb = nd;
  uint64_t st = num_slots;
  while (nd && st < (1ULL << qb)) { nd--; st = NSLOTS(nd); }
  if (st >= (1ULL << qb)) nd++;
  o->ndistinct_for_denoise = n_true + num_false / ((uint64_t)nd + 1);
  lb = nd;
  st = (uint64_t)(n_true * (enc + (double)3 / 2));
  if (st > (1ULL << (qb - 1))) ub = 0;
  else {
    st = num_slots;
    while (st >= (1ULL << (qb - 1))) { ub++; st = NSLOTS(ub); }
    if (st < (1ULL << (qb - 1))) ub--;
  }
#undef NSLOTS
  o->num_true_kmers = num_true; o->num_false_kmers = num_false;
  o->qb = qb; o->hb = qb + 8; o->num_denoise = nd; o->lower_bound = lb; o->upper_bound = ub;
}
