// Stitch quotient-range shards into the single filter the reference would hold.
//
// With G GPUs every context owns the quotients [g, g+1) * 2^qb / G and lays its runs out in
// its own table (own overflow tail). In ONE table the last cluster of shard g can spill
// into the slots of shard g+1 and push that shard's first runs to the right. The stitch
// walks the shards left to right, re-places every run at max(q, previous run end + 1)
// (the layout rule of cqf/gqf.c insert1_advance :1614-1915) and recomputes the block offsets
// (block_offset_strict, gqf.c:599-601, clamped to 255) -- the result is the table a single
// context holds for the same key multiset, i.e. what qf_serialize would write.
#pragma once
#include <stdint.h>
#include <vector>

namespace shk {

// shards[g] points to the block bytes of shard g (as returned by shk_export_blocks);
// shard_blocks[g] = number of 89-byte blocks in it. `out` must hold nblocks(qb) * 89 bytes
// and is fully overwritten. Returns 0, or -3 when the stitched runs pass xnslots.
int stitch_shards(const uint8_t *const *shards, const uint64_t *shard_blocks, uint32_t nshards, uint32_t qb,
                  uint8_t *out, uint64_t out_bytes);

// The same layout, rank by rank (no rank sees another's table; see stitch.cpp): the shard as a free-pointer function
// f -> max(f + a, b); the rank's own blocks of the single table for the free pointer it starts from, plus what its runs
// spill behind its block range; a received spill ORed into the rank's blocks.
int shard_summary(const uint8_t *shard, uint64_t shard_blocks, uint32_t g, uint32_t nshards, uint32_t qb, uint64_t *a, uint64_t *b);
int shard_layout(const uint8_t *shard, uint64_t shard_blocks, uint32_t g, uint32_t nshards, uint32_t qb, uint64_t free_in,
                 uint8_t *own_blocks, uint64_t own_bytes, std::vector<uint8_t> *spill_slots, std::vector<uint8_t> *spill_runends,
                 uint64_t *spill_start, uint64_t *free_out);
int shard_apply_spill(uint8_t *own_blocks, uint32_t g, uint32_t nshards, uint32_t qb, uint64_t spill_start, const uint8_t *slots,
                      const uint8_t *runends, uint64_t n);

}  // namespace shk
