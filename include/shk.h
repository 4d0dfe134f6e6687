/* shk.h -- C ABI of libshk.so: the MI355X-native k-mer counting path
 * (CQF-deNoise stage of SH-assembly) as a drop-in for the reference's link-time
 * boundary. Plain pointers and sizes only.
 *
 * What each entry point stands in for in the reference (/root/reference):
 *   shk_create            CQF_mt::CQF_mt(qb,hb,t,seed) -> qf_init          cqf/CQF_mt.h:427-445, cqf/gqf.c:2187-2290
 *                         + CQF_runtime_mt ctor (deNoise trigger/rounds)   cqf/CQF_mt.h:292-305
 *   shk_count_chunks      per-chunk body of fastq_to_uint64kmers_prod:
 *                         reads_to_kmers -> NTPC64 + qf_insert_advance,
 *                         trigger test, DeNoise mode (t = 1 schedule)      cqf/CQF_mt.h:610-731, 821-931; gqf.c:2432
 *   shk_hash_chunks       reads_to_kmers' hashing half only (multi-GPU
 *                         routing: keys go to their owner before insert)   cqf/CQF_mt.h:630-718; base/nthash.hpp:295-309
 *   shk_count_words       qf_insert_advance over a batch of routed keys    cqf/gqf.c:2432-2440
 *   shk_denoise           one DeNoise phase (also --endDeNoise)            cqf/CQF_mt.h:860-914, 999-1039; gqf.c:2792-3040
 *   shk_export_cqf        CQF_mt::save -> qf_serialize                     cqf/CQF_mt.h:521, 986-987; gqf.c:2379-2394
 *   shk_import_cqf        CQF_mt::load -> qf_deserialize                   cqf/CQF_mt.h:514-519; gqf.c:2396-2420
 *   shk_lookup            qf_count_key_value /
 *                         qf_count_key_value_{is,set}_traveled             cqf/gqf.c:2442-2469, 3092-3163
 *   shk_extend_forward    get_unitig_forward, the walk that meets no other
 *                         unitig (Contiger, first slice)                   src/contig_assembly.cpp:3028-3218
 *   shk_unitigs_from_seeds its two calls per seed + median abundance        src/contig_assembly.cpp:1886-1904; base/Utility.cpp:27-40
 *   shk_select_seeds      processDataChunk's seed rule                     src/contig_assembly.cpp:1856-1876
 *   shk_find_unitigs,     find_unitigs_mt_master/worker: seeds walked both ways, work queue of branch neighbours,
 *   shk_unitigs_add_*,    startKmer2unitig with "smaller id wins", known-node stops, pure circles; check_unitig,
 *   shk_unitig_set_write  track_kmer_worker, build_graph_worker, writer -- all on the device (csrc/unitig_kernels.hip)
 *                                                                         src/contig_assembly.cpp:2034-2269, 3018-3218, 935-1084, 600-629
 *   shk_insert_counted    qf_insert_advance(count > 1) -> insert_advance    cqf/gqf.c:2024-2136 (local-QF flush, CQF_mt.h:588-607)
 *   shk_dump              qf_iterator / qfi_get / qfi_next / qfi_end        cqf/gqf.c:2474-2601
 *   shk_merge             qf_merge                                          cqf/gqf.c:2614-2655
 *   shk_multi_merge       qf_multi_merge                                    cqf/gqf.c:2660-2704
 *   shk_import_shards     (no counterpart: the reference is one process) quotient-range shards -> the single table
 *   shk_stats             runtime->nelts / ndistinct_elts / num_deNoise    cqf/CQF_mt.h:277-288
 *   shk_destroy           CQF_mt::~CQF_mt -> qf_destroy                    cqf/CQF_mt.h:547-557; gqf.c:2306
 *
 * Errors: every call returns 0 or a negative SHK_ERR_* code; the library never exits
 * the process (the reference perror()+exit()s, gqf.c:2244-2247) and, unlike the
 * reference, detects a full table (SHK_ERR_TABLE_FULL) before the pass that would overflow writes anything.
 * Granularity of an error inside shk_count_chunks / shk_count_words: the call works through its chunks in ranges (a
 * range ends where a deNoise round fires); ranges and rounds completed before the failing range stay committed and
 * `stats` reports them (stats->chunks = chunks consumed), the failing range leaves the table as it was.
 * A context is not re-entrant; use one context per thread / per GPU.
 */
#ifndef SHK_H
#define SHK_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct shk_ctx shk_ctx;

enum {
  SHK_OK = 0,
  SHK_ERR_ARG = -1,          /* bad argument / unsupported geometry */
  SHK_ERR_HIP = -2,          /* a HIP runtime call failed (no GPU, out of memory, ...) */
  SHK_ERR_TABLE_FULL = -3,   /* runs would pass xnslots */
  SHK_ERR_REGION = -4,       /* one 256-quotient region exceeded the kernel's LDS image or hash */
  SHK_ERR_CORRUPT = -5,      /* table metadata inconsistent / key outside this context's range */
  SHK_ERR_FASTQ = -6,        /* malformed input: read longer than 65535, too many reads */
  SHK_ERR_BATCH = -7,        /* batch larger than the capacities given at create time */
  SHK_ERR_IO = -8
};

typedef struct shk_config {
  uint32_t qb;                     /* log2(#slots) of the whole filter (all shards) */
  uint32_t hb;                     /* hash bits, must be qb + 8 (src/CQF-deNoise.cpp:161) */
  uint32_t seed;                   /* stored in the .cqf header (src/CQF-deNoise.cpp:83) */
  uint32_t k;                      /* k-mer size, 1..191 */
  uint64_t ndistinct_for_denoise;  /* deNoise trigger (src/CQF-deNoise.cpp:125) */
  uint32_t num_denoise;            /* rounds available (runtime->num_deNoise) */
  uint32_t reserved0;
  uint64_t min_denoise_len;        /* 0 = reference value NUM_SLOTS_TO_LOCK<<4 = 1<<20 (CQF_mt.h:964) */
  uint64_t max_batch_bytes;        /* capacity: FASTQ text bytes per shk_count_chunks call */
  uint64_t max_batch_keys;         /* capacity: k-mers per batch */
  uint64_t max_batch_reads;        /* capacity: reads per batch (0 = max_batch_bytes/16) */
  int32_t device;                  /* HIP device ordinal */
  uint32_t shard_index;            /* this context owns quotients [shard_index, shard_index+1) * 2^qb / num_shards */
  uint32_t num_shards;             /* 0 or 1 = whole filter; otherwise a power of two */
  uint32_t threads_per_group;      /* 0 = 512; tests may lower it */
  uint32_t hash_groups;            /* 0 = auto; grid of the grid-stride kernels */
  uint32_t max_level_bits;         /* 0 = 10; digit bits per partition level (tests lower it) */
} shk_config;

typedef struct shk_batch_stats {
  uint64_t kmers;            /* k-mers presented to the filter by this call (qf_insert_advance calls) */
  uint64_t new_distinct;     /* keys that were new when inserted (isNew) */
  uint64_t removed;          /* entries removed by deNoise rounds fired inside this call */
  uint32_t denoise_rounds;   /* rounds fired inside this call */
  uint32_t chunks;           /* chunks consumed */
} shk_batch_stats;

typedef struct shk_totals {
  uint64_t nelts;            /* runtime->nelts */
  uint64_t ndistinct;        /* runtime->ndistinct_elts */
  uint32_t rounds_left;      /* runtime->num_deNoise */
  uint32_t rounds_done;
  uint64_t nslots, xnslots, nblocks, table_bytes;  /* this context's (shard's) geometry */
  uint64_t free_pointer;     /* first slot after the last run */
} shk_totals;

int shk_create(const shk_config *cfg, shk_ctx **out);
void shk_destroy(shk_ctx *ctx);

/* Count every k-mer of `nchunks` FASTQ chunks. chunk_off/chunk_len index into `text`
 * (text_on_device != 0: `text` is a device pointer, 16-byte aligned, that stays valid until the call returns; the
 * kernels read the text in aligned 16-byte units, so the allocation must be readable up to the next multiple of 16
 * behind text + text_bytes -- hipMalloc'ed buffers always are).
 * Chunks are the units after which the reference tests its deNoise trigger; rounds fire
 * inside the call exactly where the t = 1 reference would fire them. */
int shk_count_chunks(shk_ctx *ctx, const void *text, int text_on_device, uint64_t text_bytes,
                     const uint64_t *chunk_off, const uint64_t *chunk_len, uint32_t nchunks,
                     shk_batch_stats *stats);

/* The same in two halves, so that the front end of later batches (parse, hash, partition: its own stream and buffers)
 * runs WHILE an earlier batch is rebuilt into the table -- the overlapped form of the reference's producer threads
 * (cqf/CQF_mt.h:821-931 interleaves reading, hashing and inserting across threads). shk_prepare_chunks starts the front
 * end of a batch and returns at once; shk_count_prepared takes the OLDEST prepared batch through the rebuild (deNoise
 * rounds fire inside it exactly as in shk_count_chunks) and returns its statistics. At most two batches may be prepared
 * ahead (SHK_ERR_BATCH beyond). `text` (host or device) must stay valid until the batch has been counted. A failure of
 * the front end is returned by the shk_count_prepared of that batch; the table is untouched then. Results are those of
 * shk_count_chunks on the same batches in the same order. Not for sharded contexts (their words go through
 * shk_hash_chunks and the exchange). */
int shk_prepare_chunks(shk_ctx *ctx, const void *text, int text_on_device, uint64_t text_bytes,
                       const uint64_t *chunk_off, const uint64_t *chunk_len, uint32_t nchunks);
int shk_count_prepared(shk_ctx *ctx, shk_batch_stats *stats);
/* Allocates the front end's own buffers and stream now instead of inside the first shk_prepare_chunks (which does it
 * otherwise), and -- for a context with deNoise rounds -- the records of a deNoise point instead of in the first pass
 * that needs them: for callers that want context set-up and steady state apart, e.g. when timing. Idempotent. */
int shk_prepare_reserve(shk_ctx *ctx);

/* Overlapped ingest: start copying host text (pinned memory for full PCIe rate) for a LATER call into one of two
 * context-owned device buffers; the copy runs on its own stream while the context computes. Pass the returned
 * pointer as `text` with text_on_device = 1; that call waits for the copy. Upload batch s+1, then count batch s. */
int shk_upload_text(shk_ctx *ctx, const void *host_text, uint64_t nbytes, void **d_text);
/* page-locked host memory for the text handed to shk_upload_text (pageable memory works, at a fraction of the rate) */
int shk_host_alloc(uint64_t nbytes, void **p);
void shk_host_free(void *p);

/* Hash only: leaves `*nwords` key words (key | chunk_index << hb, reference emission
 * order) in a context-owned device buffer `*d_words`, valid until the next call.
 * In a context that is one of G shards, chunk i of the call is labelled i * G + shard_index:
 * the ranks' chunks interleave like the parts of the reference's round-robin file queue
 * (cqf/CQF_mt.h:364-390, 828-830), and nchunks * G must stay below 4096. */
int shk_hash_chunks(shk_ctx *ctx, const void *text, int text_on_device, uint64_t text_bytes,
                    const uint64_t *chunk_off, const uint64_t *chunk_len, uint32_t nchunks,
                    uint64_t **d_words, uint64_t *nwords);

/* Insert key words that are already on the device (all keys must belong to this context's
 * quotient range). nchunks = 1 + the largest chunk index present. */
int shk_count_words(shk_ctx *ctx, const uint64_t *d_words, uint64_t nwords, uint32_t nchunks,
                    shk_batch_stats *stats);

/* ---- staged form of shk_count_words for several GPUs. The deNoise trigger is a property of
 * the WHOLE filter, so with quotient-range shards the host reduces each shard's statistics
 * (all-reduce over RCCL/gloo) between these calls and takes the same decisions on every rank
 * (sh-assembly_amd/shk/dist.py mirrors the single-GPU logic of csrc/shk_api.hip merge_stage).
 *   shk_route_words    bin this rank's key words by owner (send buffer of the all-to-all)
 *   shk_stage_words    copy + partition the routed key words (chunk ids are global)
 *   shk_stage_summary  statistics of inserting the words of chunks [lo, hi]; nothing is written
 *   shk_stage_commit   write them; must follow a summary over exactly [lo, hi]
 *   shk_denoise        one round on this shard alone: its range walk (CQF_mt.h:888-895) starts at the shard's first
 *                      slot. The sharded driver uses it only as a last resort -- rounds normally go through
 *                      shk_stage_point_* / shk_stage_round_try below, which continue the walk from shard to shard
 *                      over the single table's layout */
typedef struct shk_summary {
  uint64_t new_distinct, added, removed, before;
  uint64_t hist[32];
  uint32_t err_bits;       /* raw kernel flags; SHK_SOFT_BITS are meaningless for a speculative range */
  uint32_t reserved;
} shk_summary;
#define SHK_SOFT_BITS 0x0Au     /* table full | region image exceeded: only final for a committed range */
#define SHK_HASH_FULL_BIT 0x04u /* too many distinct new keys in one region: summarise fewer chunks */
/* Group the key words left by shk_hash_chunks by owner shard (owner = top log2(nshards) bits of
 * the quotient): `*d_out` (context-owned, valid until the next call) holds them owner by owner,
 * counts[o] words for owner o -- the send buffer of the all-to-all. Two send buffers alternate: the result of one
 * call stays valid until the SECOND-next call, so batch s can be on the wire while batch s+1 is hashed and routed. */
int shk_route_words(shk_ctx *ctx, uint64_t nwords, uint32_t nshards, uint64_t **d_out, uint64_t *counts);

/* shk_hash_chunks + shk_route_words in one pass over the text: every k-mer is hashed and sent straight to its owner's
 * bin of one of the two alternating send buffers (no key word goes to HBM and back in between). Same results:
 * `*d_out` holds the words binned by owner, counts[s] of them for shard s, `*nwords` in total; chunk i of the call is
 * labelled i * num_shards + shard_index. */
int shk_hash_route_chunks(shk_ctx *ctx, const void *text, int text_on_device, uint64_t text_bytes,
                          const uint64_t *chunk_off, const uint64_t *chunk_len, uint32_t nchunks, uint32_t nshards,
                          uint64_t **d_out, uint64_t *counts, uint64_t *nwords);
int shk_stage_words(shk_ctx *ctx, const uint64_t *d_words, uint64_t nwords);
/* The same from TWO device buffers (a shard's own words, where shk_hash_route_chunks left them, and the words it received):
 * no copy that brings them together first. Neither may lie in the buffer shk_hash_chunks returns (the first partition
 * level writes there): SHK_ERR_ARG. */
/* Allocates the two send buffers of shk_route_words / shk_hash_route_chunks now instead of inside their first two calls,
 * and the records of a deNoise point (for callers that keep set-up and steady state apart). Idempotent. */
int shk_route_reserve(shk_ctx *ctx);
int shk_stage_words_pair(shk_ctx *ctx, const uint64_t *d_words_a, uint64_t nwords_a, const uint64_t *d_words_b, uint64_t nwords_b);
int shk_stage_summary(shk_ctx *ctx, uint32_t chunk_lo, uint32_t chunk_hi, uint32_t hist_base, uint32_t hist_shift,
                      int want_hist, shk_summary *out);
int shk_stage_commit(shk_ctx *ctx, uint32_t chunk_lo, uint32_t chunk_hi, const shk_summary *s);
/* try/accept form (what sh-assembly_amd/shk/dist.py uses): shk_stage_try computes everything about
 * inserting [lo, hi] that does not depend on the other ranks' decision -- statistics, run lengths and
 * encodings (kept on the device), free pointers, error flags -- and writes nothing to the live table;
 * shk_stage_accept then places the runs into the spare table and makes it the live one (only after a
 * clean try over the same range); a try that is not accepted is simply superseded by the next call */
int shk_stage_try(shk_ctx *ctx, uint32_t chunk_lo, uint32_t chunk_hi, uint32_t hist_base, uint32_t hist_shift,
                  int want_hist, shk_summary *out);
int shk_stage_accept(shk_ctx *ctx, const shk_summary *s);
/* One deNoise round on this shard fused with the insertion of the staged chunks [lo, hi] that lie behind the deNoise
 * point (one pass over the table instead of two): statistics only; shk_stage_accept writes. out->removed = singletons
 * dropped; a dropped key that reappears in [lo, hi] counts in out->new_distinct. If any rank reports err_bits, or the
 * trigger would be reached again inside [lo, hi], do not accept: run shk_denoise and go on as usual. */
int shk_stage_try_denoise(shk_ctx *ctx, uint32_t chunk_lo, uint32_t chunk_hi, shk_summary *out);
/* want_hist = 2 in shk_stage_summary / shk_stage_try additionally records the first chunk of every
 * new key; this call returns the exact histogram of that last pass: out[i] = new keys first seen in
 * chunk i, for i < n (n <= chunk_hi + 1 of the pass). With it the ranks find the chunk of a deNoise
 * point in one pass instead of refining the 32-bin histogram. SHK_ERR_ARG when the last pass has none.
 * (The first want_hist = 2 pass allocates nregions KiB of device memory for the first-chunk records.) */
int shk_stage_chunk_hist(shk_ctx *ctx, uint64_t *out, uint32_t n);

/* ---- a deNoise point inside the staged batch in ONE rebuild per shard (the sharded form of what shk_count_chunks does
 * on a single table; replaces try(lo..point) + accept + try_denoise + accept). The ranks take the steps together
 * (sh-assembly_amd/shk/dist.py: _one_pass_point); nothing is written before shk_stage_accept.
 *   shk_stage_sample        statistics pass over every n-th region: hist[i] = keys new to the filter first seen in chunk
 *                           i, within the sample (i <= chunk_hi); summed over the shards and scaled by regions / sampled
 *                           it predicts the chunk at which the filter-wide distinct count reaches the trigger
 *   shk_stage_point_try     the rebuild with two counts per key (chunks <= split / behind it): entries whose count at the
 *                           split is 1 are dropped (qf_remove_singletons as run by CQF_mt.h:860-869 after chunk `split`),
 *                           the later chunks added on top. Also records the exact first-chunk histogram
 *                           (shk_stage_chunk_hist) with which the ranks check that `split` IS the chunk of the point.
 *                           out->islots / ifin / first_used describe the shard's table at the split (slots in use, free
 *                           pointer behind its last quotient, run on quotient 0): what the next shard needs to lay its
 *                           own part out as in the single table
 *   shk_stage_point_walk    the round's range walk (CQF_mt.h:888-895) over this shard, continuing the previous shard's:
 *                           carry = slots the earlier shards spill over this shard's first quotient (>= 0), prev_fp = their
 *                           free pointer relative to it (negative when they end before it; -1 for shard 0), state_in/out
 *                           = {0, 0} between two ranges / {1, minimum end of the open range relative to the next shard};
 *                           *nprot singletons on range ends survive the round (kept on the device for _finish)
 *   shk_stage_point_finish  rebuilds the regions holding those; `out` = final statistics, `accept` = what to hand to
 *                           shk_stage_accept when every rank is clean and the trigger is not reached again in the rest */
typedef struct shk_point {
  uint64_t new_after, added_after;   /* keys new to the table after the round / occurrences from the chunks behind the split */
  uint64_t removed, added_before;    /* singletons dropped / occurrences from the chunks up to the split */
  uint64_t islots, ifin;
  uint32_t first_used;
  uint32_t err_bits;                 /* any bit: do not go on with this point (take the three-pass path) */
} shk_point;
int shk_stage_sample(shk_ctx *ctx, uint32_t chunk_lo, uint32_t chunk_hi, uint64_t *hist, uint32_t *regions, uint32_t *sampled,
                     uint32_t *err_bits);
int shk_stage_point_try(shk_ctx *ctx, uint32_t chunk_lo, uint32_t split, uint32_t chunk_hi, shk_point *out);
/* the same for a round on its own (no words: the reference's --endDeNoise round, or a round the batch ends on);
 * followed by shk_stage_point_walk / _finish / shk_stage_accept like a point. split = chunk_lo - 1 in
 * shk_stage_point_try puts the round in front of all the staged chunks [chunk_lo, chunk_hi]. */
int shk_stage_round_try(shk_ctx *ctx, shk_point *out);
int shk_stage_point_walk(shk_ctx *ctx, int64_t carry, int64_t prev_fp, int last, int next_first_used, const uint64_t state_in[2],
                         uint64_t state_out[2], uint64_t *nprot, uint32_t *err_bits);
int shk_stage_point_finish(shk_ctx *ctx, shk_point *out, shk_summary *accept);

/* One deNoise round now (the reference's --endDeNoise round; does not use up num_denoise). */
int shk_denoise(shk_ctx *ctx, uint64_t *removed);

/* ---- filter-to-filter utilities
 * shk_insert_counted: add `counts[i]` occurrences of keys[i] (keys < 2^hb inside this context's quotient range), the
 *   batched form of qf_insert_advance(qf, key, 0, count, ...). No deNoise round fires inside (the reference tests its
 *   trigger after a chunk of reads only). stats->kmers = occurrences added, stats->new_distinct = keys that were new.
 * shk_dump: (key, count) of every entry, in the order qf_iterator/qfi_next visit them (ascending key); *n_out = number
 *   of entries; at most `cap` pairs are written; keys == NULL only counts. key = (quotient << 8) | remainder (qfi_get).
 *   The reference's qfi_next ends the iteration when it steps inside a run onto a slot behind nslots (gqf.c:2537-2539),
 *   so entries in the overflow tail that do not open their run, and all behind them, are invisible to its iterator and
 *   to qf_merge: ref_iterator_end != 0 makes *n_out that (smaller) number; 0 = every entry.
 * shk_merge: dst := dst + src -- what qf_merge(a, b, c) leaves in c for a = dst, b = src: the canonical table of the
 *   summed multiset (same geometry, same device). stats->kmers = occurrences added, new_distinct = keys new to dst.
 * shk_multi_merge: the same for several sources (qf_multi_merge).
 * shk_import_shards: replace the table (whole-filter context, num_shards <= 1) by the union of `nshards` quotient-range
 *   shards, each in the layout shk_export_blocks gives for a context with num_shards = nshards (its nslots / nshards
 *   quotients plus its own overflow tail). Device tables need 16 readable bytes behind them. nelts / ndistinct:
 *   the runtime counters for the header (0 = take the sums found in the shards). */
int shk_insert_counted(shk_ctx *ctx, const uint64_t *keys, const uint64_t *counts, uint64_t n, int on_device,
                       shk_batch_stats *stats);
int shk_dump(shk_ctx *ctx, uint64_t *keys, uint64_t *counts, uint64_t cap, int on_device, int ref_iterator_end,
             uint64_t *n_out);
int shk_merge(shk_ctx *dst, shk_ctx *src, shk_batch_stats *stats);
int shk_multi_merge(shk_ctx *dst, shk_ctx *const *srcs, uint32_t n, shk_batch_stats *stats);
int shk_import_shards(shk_ctx *ctx, const void *const *shard_blocks, const uint64_t *shard_bytes, uint32_t nshards,
                      int on_device, uint64_t nelts, uint64_t ndistinct);

int shk_stats(shk_ctx *ctx, shk_totals *out);
/* 128-byte quotient_filter_metadata image (gqf.h:62-77) for this context */
int shk_header(shk_ctx *ctx, uint8_t out[128]);
/* table bytes (nblocks * 89) to host memory */
int shk_export_blocks(shk_ctx *ctx, void *host_dst, uint64_t cap);
/* device pointer and size of the live table (valid until the next call that rebuilds the table) */
int shk_table_ptr(shk_ctx *ctx, void **d_table, uint64_t *nbytes);
/* header + blocks, byte-identical to qf_serialize */
int shk_export_cqf(shk_ctx *ctx, const char *path);
/* replace the table by one read from a .cqf (whole filter; num_shards must be 1) */
int shk_import_cqf(shk_ctx *ctx, const char *path);
int shk_import_blocks(shk_ctx *ctx, const void *host_src, uint64_t nbytes, uint64_t nelts, uint64_t ndistinct);

/* mode 0: count + is_traveled, 1: count + set_traveled (returns the bit before), 2: count only.
 * keys/counts/was_traveled are host pointers unless on_device != 0. was_traveled may be NULL. */
int shk_lookup(shk_ctx *ctx, const uint64_t *keys, uint64_t n, int on_device, int mode,
               uint64_t *counts, uint8_t *was_traveled);

/* ---- Contiger, first slice: forward extension of many open unitig ends at once.
 * For end i: cur_kmers[i*k ..] is the contig's last k-mer, first_kmers[i*k ..] its first (upper-case ACGT). Per
 * step the kernel looks up the 4 successors and the 3 siblings of the current k-mer and applies the reference's
 * rule (stop on a solid sibling or > 1 solid successor; extend on exactly one; stop on none or on a pure
 * circle), for a walk that meets no other unitig (no startKmer2unitig hits). out_bases/out_counts hold up to
 * max_ext appended bases and the filter counts of the k-mers they complete, out_n their number, out_stop why
 * the walk ended (SHK_STOP_*), out_branch (may be NULL) at a SHK_STOP_BRANCH the solid neighbours of the last
 * k-mer s0 s1..s(k-1): bit x = successor s1..s(k-1)+x, bit 4+z = sibling z+s1..s(k-1) (x, z index "ACGT"); out_ncount (may be NULL, 8 per end) their filter counts. mark_traveled != 0 sets the traveled bit of every k-mer looked up, as the
 * reference's count_key_value_set_traveled does. k <= 64. All pointers are host pointers. */
#define SHK_STOP_BRANCH 1
#define SHK_STOP_DEAD_END 2
#define SHK_STOP_CIRCLE 3
#define SHK_STOP_BUFFER 4
#define SHK_STOP_BAD_SEED 5
int shk_extend_forward(shk_ctx *ctx, const char *cur_kmers, const char *first_kmers, uint32_t n, uint32_t k,
                       uint64_t abundance_min, int mark_traveled, uint32_t max_ext, char *out_bases,
                       uint32_t *out_counts, uint32_t *out_n, uint8_t *out_stop, uint8_t *out_branch,
                       uint32_t *out_ncount);
/* One maximal unitig per seed k-mer: extend, reverse-complement, extend again; out_seq[i*max_len ..] holds
 * out_len[i] bases, out_median[i] the contig's median abundance as the reference stores it (int),
 * out_stop[2*i], out_stop[2*i+1] the two stop reasons. seed_counts[i] = the seed's filter count. */
int shk_unitigs_from_seeds(shk_ctx *ctx, const char *seeds, const uint32_t *seed_counts, uint32_t n, uint32_t k,
                           uint64_t abundance_min, uint32_t max_len, char *out_seq, uint32_t *out_len,
                           int32_t *out_median, uint8_t *out_stop);

/* All unitigs reachable from the seeds, written as FASTA in the reference's record grammar
 * (`>i LN:i:len KC:i:median*(len-k+1) km:f:median L:+:j:+ ... L:-:j:- ...`, successors in A,C,G,T order, predecessors
 * in T,G,C,A order, src/contig_assembly.cpp:606-626, 1012-1084): seeds are extended in both directions, every solid
 * neighbour met at a branch starts a new contig (the reference's work queue), a unitig found more than once is
 * kept once. The reference produces the same set of sequences up to reverse complement; ids and order are its
 * thread schedule's and are not reproduced. */
typedef struct shk_unitig_stats {
  uint64_t unitigs, total_len;   /* kept unitigs and their summed length */
  uint64_t rounds, extensions;   /* launches of the walk kernel; bases appended by all walks (duplicates included) */
  uint64_t duplicates, truncated;/* unitigs found again and dropped; walks cut at max_len */
} shk_unitig_stats;
int shk_find_unitigs(shk_ctx *ctx, const char *seeds, const uint32_t *seed_counts, uint32_t n, uint32_t k,
                     uint64_t abundance_min, uint32_t max_len, const char *out_path, shk_unitig_stats *stats);
/* The same in pieces, for callers that feed seeds batch by batch (the Contiger command line): a unitig set that
 * accumulates over calls. With mark_traveled != 0 every k-mer an extension looks up is marked, so that
 * shk_select_seeds(use_traveled = 1) on later reads skips seeds inside unitigs that are already known -- the
 * reference's own pruning (contig_assembly.cpp:1871-1873). */
typedef struct shk_unitig_set shk_unitig_set;
shk_unitig_set *shk_unitig_set_new(void);
void shk_unitig_set_free(shk_unitig_set *u);
int shk_unitigs_add_seeds(shk_ctx *ctx, shk_unitig_set *u, const char *seeds, const uint32_t *seed_counts, uint32_t n,
                          uint32_t k, uint64_t abundance_min, uint32_t max_len, int mark_traveled);
int shk_unitig_set_write(shk_unitig_set *u, uint32_t k, const char *out_path, shk_unitig_stats *stats);
/* The command line's inner loop without the seeds leaving the device: the seeds of the reads in the given FASTQ chunks
 * (rule of shk_select_seeds with use_traveled = 1) become contigs and are walked at once, every lookup marking the
 * traveled bit as the reference's does. *nseeds (may be NULL) = seeds taken from this batch. */
int shk_unitigs_add_reads(shk_ctx *ctx, shk_unitig_set *u, const void *text, int text_on_device, uint64_t text_bytes,
                          const uint64_t *chunk_off, const uint64_t *chunk_len, uint32_t nchunks, uint32_t k,
                          uint64_t abundance_min, uint64_t count_min, uint64_t count_max, uint32_t max_len, uint64_t *nseeds);
/* Seeds of the reads in the given FASTQ chunks (processDataChunk, contig_assembly.cpp:1856-1876): the k-mer at
 * len/2 - k/2 of every read, upper-cased, without 'N', whose filter count lies in [count_min, count_max]; with
 * use_traveled != 0 the lookup marks the k-mer and a k-mer that was already marked gives no seed. out_seeds
 * receives *n_out * k bases (capacity cap seeds), out_counts their counts. */
int shk_select_seeds(shk_ctx *ctx, const void *text, int text_on_device, uint64_t text_bytes, const uint64_t *chunk_off,
                     const uint64_t *chunk_len, uint32_t nchunks, uint32_t k, uint64_t count_min, uint64_t count_max,
                     int use_traveled, char *out_seeds, uint32_t *out_counts, uint32_t cap, uint32_t *n_out);

/* per-kernel device time measured with HIP events on the context's stream */
typedef struct shk_kernel_time {
  const char *name;
  uint64_t launches;
  double ms;
} shk_kernel_time;
int shk_profile_enable(shk_ctx *ctx, int on);
int shk_profile_get(shk_ctx *ctx, shk_kernel_time *out, int cap);  /* returns #entries */
int shk_profile_reset(shk_ctx *ctx);

const char *shk_strerror(int code);
/* last kernel error bits (diagnostics) */
uint32_t shk_last_error_bits(shk_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
