#include "stitch.hpp"

#include <math.h>
#include <string.h>

namespace shk {

static const int BLK = 89, OFF_OCC = 1, OFF_RUN = 9, OFF_SLOTS = 25;

static inline uint64_t ld64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline void or_bit(uint8_t *blocks, uint64_t slot, int field) {
  blocks[(slot >> 6) * BLK + field + ((slot & 63) >> 3)] |= (uint8_t)(1u << (slot & 7));
}

int stitch_shards(const uint8_t *const *shards, const uint64_t *shard_blocks, uint32_t nshards, uint32_t qb,
                  uint8_t *out, uint64_t out_bytes) {
  const uint64_t nslots = 1ULL << qb;
  const uint64_t xnslots = nslots + (uint64_t)(10 * sqrt((double)nslots));   // qf_init, gqf.c:2197
  const uint64_t nblocks = (xnslots + 63) / 64;
  if (out_bytes < nblocks * BLK) return -1;
  memset(out, 0, nblocks * BLK);
  const uint64_t per = nslots / nshards;
  uint64_t free_ptr = 0;                  // first slot after the last placed run
  std::vector<uint64_t> block_free(nblocks + 1, 0);  // free pointer when entering block b (for the offsets)
  uint64_t next_block = 0;
  for (uint32_t g = 0; g < nshards; g++) {
    const uint8_t *sh = shards[g];
    const uint64_t q_lo = per * g;
    uint64_t lfree = 0;                   // local free pointer inside the shard (to find run starts)
    for (uint64_t lb = 0; lb < per / 64; lb++) {
      uint64_t occ = ld64(sh + lb * BLK + OFF_OCC);
      while (occ) {
        const unsigned bit = (unsigned)__builtin_ctzll(occ);
        occ &= occ - 1;
        const uint64_t lq = lb * 64 + bit;
        // the run of local quotient lq starts at max(lq, lfree) and ends at its runend bit
        uint64_t ls = lq > lfree ? lq : lfree;
        uint64_t le = ls;
        for (;;) {
          if (le / 64 >= shard_blocks[g]) return -5;
          if ((ld64(sh + (le >> 6) * BLK + OFF_RUN) >> (le & 63)) & 1) break;
          le++;
        }
        lfree = le + 1;
        const uint64_t q = q_lo + lq, len = le - ls + 1;
        // offsets of the blocks this quotient passes
        while (next_block <= q / 64) block_free[next_block++] = free_ptr;
        const uint64_t st = q > free_ptr ? q : free_ptr;
        if (st + len > xnslots) return -3;
        for (uint64_t i = 0; i < len; i++)
          out[((st + i) >> 6) * BLK + OFF_SLOTS + ((st + i) & 63)] = sh[((ls + i) >> 6) * BLK + OFF_SLOTS + ((ls + i) & 63)];
        or_bit(out, q, OFF_OCC);
        or_bit(out, st + len - 1, OFF_RUN);
        free_ptr = st + len;
      }
    }
  }
  while (next_block < nblocks) block_free[next_block++] = free_ptr;
  for (uint64_t b = 0; b < nblocks; b++) {
    const uint64_t f = block_free[b], s = b * 64;
    const uint64_t o = f > s ? f - s : 0;
    out[b * BLK] = (uint8_t)(o > 255 ? 255 : o);
  }
  return 0;
}

// ------------------------------------------------------------------ the same, one rank at a time
// What stitch_shards does for all shards in one place, split so that every rank lays out ITS OWN blocks of the single
// table and nobody gathers the others' tables:
//   1. shard_summary: the shard as a function on the free pointer, f -> max(f + a, b) (a = slots its runs take, b = where
//      its last run ends when nothing is carried in). The ranks exchange (a, b) -- 16 bytes each -- and every rank folds
//      the pairs of the ranks in front of it into the free pointer F it starts from.
//   2. shard_layout: the rank's runs re-placed from F into its own block range (quotient range / 64; the last rank also
//      owns the overflow tail), with occupieds, runends, slots and the block offsets (which only depend on F and on the
//      rank's own earlier runs). Whatever lands behind its range is its SPILL: one contiguous stretch of slots.
//   3. the spills (a few hundred slots, normally) are exchanged; shard_apply_spill ORs the part of each that falls into
//      the rank's range into its blocks. Slots never collide: a later run always starts behind the earlier runs' end.
namespace {
struct RunWalker {     // the runs of a shard in quotient order: (local quotient, local start, length)
  const uint8_t *sh; uint64_t nblocks_own, sh_blocks; uint64_t lb = 0, occ = 0, lfree = 0;
  RunWalker(const uint8_t *s, uint64_t own, uint64_t all) : sh(s), nblocks_own(own), sh_blocks(all) {
    if (nblocks_own) occ = ld64(sh + OFF_OCC);
  }
  // 1 = a run, 0 = done, -5 = a run without its end bit (corrupt shard)
  int next(uint64_t *lq, uint64_t *ls, uint64_t *len) {
    while (!occ) {
      lb++;
      if (lb >= nblocks_own) return 0;
      occ = ld64(sh + lb * BLK + OFF_OCC);
    }
    const unsigned bit = (unsigned)__builtin_ctzll(occ);
    occ &= occ - 1;
    *lq = lb * 64 + bit;
    // the run of local quotient lq starts at max(lq, local free pointer) and ends at its runend bit
    const uint64_t s0 = *lq > lfree ? *lq : lfree;
    uint64_t e = s0;
    for (;;) {
      if (e / 64 >= sh_blocks) return -5;
      if ((ld64(sh + (e >> 6) * BLK + OFF_RUN) >> (e & 63)) & 1) break;
      e++;
    }
    lfree = e + 1;
    *ls = s0; *len = e - s0 + 1;
    return 1;
  }
};
}  // namespace

int shard_summary(const uint8_t *shard, uint64_t shard_blocks, uint32_t g, uint32_t nshards, uint32_t qb, uint64_t *a, uint64_t *b) {
  const uint64_t per = (1ULL << qb) / nshards, q_lo = per * g;
  RunWalker w(shard, per / 64, shard_blocks);
  uint64_t lq, ls, len, used = 0, fp = 0;
  int r;
  while ((r = w.next(&lq, &ls, &len)) == 1) {
    const uint64_t q = q_lo + lq;
    used += len;
    fp = (q > fp ? q : fp) + len;
  }
  if (r < 0) return r;
  *a = used; *b = fp;
  return 0;
}

int shard_layout(const uint8_t *shard, uint64_t shard_blocks, uint32_t g, uint32_t nshards, uint32_t qb, uint64_t free_in,
                 uint8_t *own_blocks, uint64_t own_bytes, std::vector<uint8_t> *spill_slots, std::vector<uint8_t> *spill_runends,
                 uint64_t *spill_start, uint64_t *free_out) {
  const uint64_t nslots = 1ULL << qb;
  const uint64_t xnslots = nslots + (uint64_t)(10 * sqrt((double)nslots));
  const uint64_t nblocks = (xnslots + 63) / 64;
  const uint64_t per = nslots / nshards, q_lo = per * g;
  const uint64_t b_lo = q_lo / 64, b_hi = g + 1 == nshards ? nblocks : (q_lo + per) / 64;    // my blocks
  if (own_bytes < (b_hi - b_lo) * BLK) return -1;
  memset(own_blocks, 0, (b_hi - b_lo) * BLK);
  const uint64_t s_lo = b_lo * 64, s_hi = b_hi * 64;                                        // my slots
  spill_slots->clear(); spill_runends->clear();
  *spill_start = free_in > s_hi ? free_in : s_hi;
  RunWalker w(shard, per / 64, shard_blocks);
  uint64_t lq, ls, len, free_ptr = free_in, next_block = b_lo;
  auto put_slot = [&](uint64_t at, uint8_t v, bool runend) {
    if (at < s_hi) {
      own_blocks[((at - s_lo) >> 6) * BLK + OFF_SLOTS + (at & 63)] = v;
      if (runend) or_bit(own_blocks, at - s_lo, OFF_RUN);
    } else {
      const uint64_t i = at - *spill_start;
      if (spill_slots->size() <= i) { spill_slots->resize(i + 1, 0); spill_runends->resize(i / 8 + 1, 0); }
      (*spill_slots)[i] = v;
      if (runend) (*spill_runends)[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
  };
  int r;
  while ((r = w.next(&lq, &ls, &len)) == 1) {
    const uint64_t q = q_lo + lq;
    // offsets of my blocks this quotient passes: the free pointer on entering the block
    while (next_block <= q / 64 && next_block < b_hi) {
      const uint64_t s = next_block * 64, o = free_ptr > s ? free_ptr - s : 0;
      own_blocks[(next_block - b_lo) * BLK] = (uint8_t)(o > 255 ? 255 : o);
      next_block++;
    }
    const uint64_t st = q > free_ptr ? q : free_ptr;
    if (st + len > xnslots) return -3;
    for (uint64_t i = 0; i < len; i++)
      put_slot(st + i, shard[((ls + i) >> 6) * BLK + OFF_SLOTS + ((ls + i) & 63)], i + 1 == len);
    or_bit(own_blocks, q - s_lo, OFF_OCC);
    free_ptr = st + len;
  }
  if (r < 0) return r;
  while (next_block < b_hi) {
    const uint64_t s = next_block * 64, o = free_ptr > s ? free_ptr - s : 0;
    own_blocks[(next_block - b_lo) * BLK] = (uint8_t)(o > 255 ? 255 : o);
    next_block++;
  }
  *free_out = free_ptr;
  return 0;
}

int shard_apply_spill(uint8_t *own_blocks, uint32_t g, uint32_t nshards, uint32_t qb, uint64_t spill_start, const uint8_t *slots,
                      const uint8_t *runends, uint64_t n) {
  const uint64_t nslots = 1ULL << qb;
  const uint64_t xnslots = nslots + (uint64_t)(10 * sqrt((double)nslots));
  const uint64_t nblocks = (xnslots + 63) / 64;
  const uint64_t per = nslots / nshards, q_lo = per * g;
  const uint64_t s_lo = (q_lo / 64) * 64, s_hi = (g + 1 == nshards ? nblocks : (q_lo + per) / 64) * 64;
  for (uint64_t i = 0; i < n; i++) {
    const uint64_t at = spill_start + i;
    if (at < s_lo || at >= s_hi) continue;
    own_blocks[((at - s_lo) >> 6) * BLK + OFF_SLOTS + (at & 63)] = slots[i];
    if ((runends[i >> 3] >> (i & 7)) & 1) or_bit(own_blocks, at - s_lo, OFF_RUN);
  }
  return 0;
}

}  // namespace shk
