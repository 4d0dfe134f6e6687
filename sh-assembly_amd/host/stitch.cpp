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

}  // namespace shk
