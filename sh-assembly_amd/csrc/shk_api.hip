// Host side of libshk.so: context, buffers, launch sequences. See include/shk.h for the
// reference interface each entry point replaces. Everything here runs on one HIP stream
// per context; the only host<->device synchronisations in a batch are the one that reads
// the merge statistics (needed to decide where a deNoise round fires) and the final one.
#include "../../include/shk.h"
#include <thread>

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
#include <algorithm>
#include <unordered_set>
#include <unordered_map>

#include "kmer_kernels.hip"
#include "roll_kernels.hip"
#include "partition_kernels.hip"
#include "cqf_kernels.hip"
#include "merge2_kernels.hip"
#include "walk_kernels.hip"
#include "unitig_kernels.hip"

#define SHK_SLACK 256  // bytes of slack behind buffers read with wide loads

enum {
  KP_COUNT_LINES, KP_SCAN_CHUNKS, KP_EMIT_READS, KP_COUNT_KEYS, KP_HASH, KP_SCAN, KP_RP_PREP, KP_RP_HIST,
  KP_RP_SCATTER, KP_MERGE_SUM, KP_REGION_SCAN, KP_MERGE_WRITE, KP_MERGE_SINGLE, KP_MERGE_SPILL, KP_PLACE, KP_MARKS, KP_LOOKUP, KP_WALK, KP_UG_WALK, KP_UG_FINISH, KP_MERGE_FUSED, KP_MERGE_SAMPLE, KP_MISC, KP_ROLL_HIST, KP_ROLL_SCATTER, KP_PACK, KP_RP_SLOTS, KP_N
};
static const char *kp_names[KP_N] = {
  "k_count_lines", "k_scan_chunks", "k_emit_reads", "k_count_keys", "k_hash_reads", "k_scan_*", "k_rp_prep",
  "k_rp_hist", "k_rp_scatter", "k_region_merge<summary>", "k_region_scan", "k_region_merge<write>", "k_region_merge<single>",
  "k_region_merge<spill>", "k_region_place", "k_denoise_marks", "k_lookup", "k_extend_forward+k_select_seeds", "k_ug_walk", "k_ug_check/emit/median/links", "k_region_merge<fused>", "k_region_merge<sample>", "misc", "k_roll_hist", "k_roll_scatter", "k_pack_reads", "k_rp_slot_cursors"};

struct PendingEvent { int id; hipEvent_t a, b; };

struct shk_ctx {
  shk_config cfg;
  int dev;
  hipStream_t stream;
  // geometry of this context (shard)
  uint64_t g_nslots;            // whole filter
  uint64_t q_lo, nslots, xnslots, nblocks, table_bytes;
  uint32_t nregions, rbits;     // regions and ceil(log2(nregions))
  uint32_t nlevels;
  ShkRpLevel lv[4];
  uint32_t threads, hash_groups;
  // state
  uint64_t nelts, ndistinct;
  uint32_t rounds_left, rounds_done;
  // device buffers
  uint8_t *tab[2];
  uint64_t *fin[2];
  int cur;                      // which of tab[]/fin[] is live
  uint8_t *d_text;
  uint8_t *d_up[2];             // shk_upload_text: two alternating buffers filled on a copy stream
  hipStream_t copy_stream;
  hipEvent_t up_done[2];
  int up_next, up_pending[2];
  uint64_t *d_chunk_off, *d_chunk_len, *d_nlines, *d_reads_base;
  uint64_t *d_rd_start, *d_rd_end;
  uint16_t *d_rd_chunk;         // chunk (within the call) of every read
  uint32_t *d_nkeys;
  uint64_t *d_key_base;
  uint64_t *d_words[2];
  uint64_t *d_scalars;          // [0] nreads, [1] nwords, [2] scan total scratch, [3] marks
  uint64_t *d_block_sums;
  uint64_t *d_hist[4];          // per level: nbuckets*P (first level: times its window groups)
  uint64_t *d_base_sub;         // first level with window groups: scanned bases of the (digit, group) sub-buckets
  uint64_t *d_base[5];          // base[l]: bucket bases entering level l (base[nlevels] = region_base)
  uint64_t *d_cursor;
  uint32_t *d_tfb;
  uint32_t *d_summary;
  long long *d_tile_a, *d_tile_b, *d_tile_f;
  unsigned long long *d_lb_agg, *d_lb_incl;
  int big_image;                // 1: rebuild kernels run with the SHK_IMG_BLOCKS_BIG image (set after a cluster outgrew the small one)
  int single_ok;                // 1: single-launch rebuild with look-back (SHK_SINGLE=1); 0 after it had to give up once
  uint32_t merge_group;         // threads per region workgroup (one wave rebuilds; the others help staging and folding)
  int use_spill;                // 1 (default): summary launch spills lengths + encodings, k_region_place writes table B
  uint8_t *d_spill;
  uint32_t *d_over_list;
  // what the spill records currently describe (a write pass may use them only for the same request)
  int spill_valid; const uint64_t *spill_words; uint32_t spill_lo, spill_hi; int spill_denoise, spill_big;
  uint64_t spill_nover;
  uint16_t *d_newchunks;        // first chunks of new keys per region (exact deNoise point in one pass); null without deNoise rounds
  unsigned long long *d_chist;  // [SHK_MAX_CHUNKS]
  uint64_t *h_chist;            // pinned
  uint32_t chist_n;             // entries of h_chist valid from the last summary (0 = none)
  uint32_t sample_stride;       // sampled statistics pass before a deNoise point: every n-th region (<= 1: off)
  uint32_t region_cap;          // how the partitioned words lie: 0 = d_base[nlevels] holds exact offsets; else region r owns the slot
                                // [r * region_cap, ...) and d_base[nlevels][r] is its END (ShkRpLevel::slot_cap)
  uint32_t slot_overflows;      // consecutive batches whose slotted last level overflowed; at 2 the slots are switched off
  const uint64_t *stage2_b;     // shk_stage_words_pair: the second source of the first partition level (null = one source)
  uint64_t stage2_na, stage2_nb;
  int slots_off;
  uint32_t pt_lo, pt_split, pt_hi; int pt_valid; uint64_t pt_nprot;   // one-pass deNoise point in progress (shk_stage_point_*)
  const uint64_t *pt_words;     // its words (null: a round on its own, shk_stage_round_try)
  unsigned long long *d_counters;  // 4 counters + 32 hist bins
  uint32_t *d_err;
  uint64_t *h_pinned;           // pinned mirror: counters(4) hist(32) err(1) scalars(4)
  uint64_t max_reads;
  // profiling
  int prof_on;
  double prof_ms[KP_N];
  uint64_t prof_n[KP_N];
  std::vector<PendingEvent> pending;
  std::vector<hipEvent_t> evpool;
  uint32_t last_err_bits;
  double new_frac;              // new distinct keys per presented k-mer in the last committed range (predicts crossings)
  int staged;                   // which d_words[] holds the partitioned words of shk_stage_words
  // one-pass deNoise point (denoise_fused): the intermediate table's (T, c), run lengths and free pointers; protected quotients
  uint32_t *d_isum; uint8_t *d_ilens; uint64_t *d_fin_i; uint64_t *d_prot;
  int counted;                  // 1 while shk_insert_counted runs: the words' chunk field is a multiplicity
  uint64_t *d_send[2];          // shk_route_words: two alternating send buffers (allocated on first use), so that the
  int send_next;                // exchange of one batch can run while the next batch is hashed and routed
  struct ShkFront *front;       // shk_prepare_chunks: the front end (parse, hash, partition) of later batches on its own stream
};

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "libshk: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); return SHK_ERR_HIP; } } while (0)

static hipEvent_t ev_get(shk_ctx *c) {
  if (!c->evpool.empty()) { hipEvent_t e = c->evpool.back(); c->evpool.pop_back(); return e; }
  hipEvent_t e; hipEventCreate(&e); return e;
}
struct ProfScope {
  shk_ctx *c; int id; hipEvent_t a, b;
  ProfScope(shk_ctx *c_, int id_) : c(c_), id(id_) {
    if (c->prof_on) { a = ev_get(c); b = ev_get(c); hipEventRecord(a, c->stream); }
  }
  ~ProfScope() {
    if (c->prof_on) { hipEventRecord(b, c->stream); PendingEvent p = {id, a, b}; c->pending.push_back(p); }
  }
};
static void prof_collect(shk_ctx *c) {
  for (size_t i = 0; i < c->pending.size(); i++) {
    float ms = 0;
    hipEventSynchronize(c->pending[i].b);
    hipEventElapsedTime(&ms, c->pending[i].a, c->pending[i].b);
    c->prof_ms[c->pending[i].id] += ms;
    c->prof_n[c->pending[i].id]++;
    c->evpool.push_back(c->pending[i].a);
    c->evpool.push_back(c->pending[i].b);
  }
  c->pending.clear();
}

static int map_err_bits(uint32_t bits) {
  if (!bits) return SHK_OK;
  if (bits & SHK_E_TABLE_FULL) return SHK_ERR_TABLE_FULL;
  if (bits & (SHK_E_OLD_EXTENT | SHK_E_NEW_EXTENT | SHK_E_HASH_FULL | SHK_E_RUN_TOO_LONG | SHK_E_LOOKBACK)) return SHK_ERR_REGION;
  if (bits & SHK_E_CORRUPT) return SHK_ERR_CORRUPT;
  if (bits & SHK_E_BAD_FASTQ) return SHK_ERR_FASTQ;
  if (bits & SHK_E_KEYS_FULL) return SHK_ERR_BATCH;
  return SHK_ERR_CORRUPT;
}

extern "C" const char *shk_strerror(int code) {
  switch (code) {
    case SHK_OK: return "ok";
    case SHK_ERR_ARG: return "bad argument or unsupported geometry";
    case SHK_ERR_HIP: return "HIP runtime error (is a GPU present?)";
    case SHK_ERR_TABLE_FULL: return "counting quotient filter is full";
    case SHK_ERR_REGION: return "a 256-quotient region exceeds the kernel's on-chip image or hash";
    case SHK_ERR_CORRUPT: return "table metadata inconsistent or key outside this context's range";
    case SHK_ERR_FASTQ: return "malformed FASTQ input";
    case SHK_ERR_BATCH: return "batch exceeds the capacities given to shk_create";
    case SHK_ERR_IO: return "file I/O error";
  }
  return "unknown error";
}
extern "C" uint32_t shk_last_error_bits(shk_ctx *c) { return c ? c->last_err_bits : 0; }

template <typename T> static int dmalloc(T **p, uint64_t n) {
  void *v = nullptr;
  HIPCHK(hipMalloc(&v, n * sizeof(T) + SHK_SLACK));
  *p = (T *)v;
  return 0;
}

// ------------------------------------------------------------------ create / destroy
static int ensure_chist(shk_ctx *c);
static int point_alloc(shk_ctx *c);
extern "C" int shk_create(const shk_config *cfg, shk_ctx **out) {
  if (!cfg || !out) return SHK_ERR_ARG;
  if (cfg->qb < 6 || cfg->qb > 40 || cfg->hb != cfg->qb + 8 || cfg->k < 1 || cfg->k > SHK_MAX_K) return SHK_ERR_ARG;
  uint32_t ns = cfg->num_shards ? cfg->num_shards : 1;
  if (ns & (ns - 1)) return SHK_ERR_ARG;
  if (cfg->shard_index >= ns) return SHK_ERR_ARG;
  if (cfg->hb + SHK_CHUNK_BITS > 64) return SHK_ERR_ARG;
  shk_ctx *c = new shk_ctx();
  c->cfg = *cfg;
  c->dev = cfg->device;
  HIPCHK(hipSetDevice(c->dev));
  HIPCHK(hipStreamCreate(&c->stream));
  c->g_nslots = 1ULL << cfg->qb;
  if (c->g_nslots / ns < 64) { delete c; return SHK_ERR_ARG; }
  c->nslots = c->g_nslots / ns;
  c->q_lo = c->nslots * cfg->shard_index;
  // qf_init geometry, gqf.c:2197-2198 (every shard keeps a full-size overflow tail)
  c->xnslots = c->nslots + (uint64_t)(10 * sqrt((double)c->g_nslots));
  c->nblocks = (c->xnslots + 63) / 64;
  c->table_bytes = c->nblocks * SHK_BLOCK_BYTES;
  c->nregions = (uint32_t)((c->nslots + SHK_REGION - 1) / SHK_REGION);
  c->rbits = 0;
  while ((1u << c->rbits) < c->nregions) c->rbits++;
  const uint32_t mlb = cfg->max_level_bits ? cfg->max_level_bits : 10;
  if (mlb > 10) { delete c; return SHK_ERR_ARG; }
  c->nlevels = (c->rbits + mlb - 1) / mlb;
  if (c->nlevels == 0) c->nlevels = 1;   // one region: a single pass still converts the words to 32-bit records
  if (c->nlevels > 4) { delete c; return SHK_ERR_ARG; }
  {
    uint32_t left = c->rbits, nb = 1;
    for (uint32_t l = 0; l < c->nlevels; l++) {
      uint32_t bits = (left + (c->nlevels - l) - 1) / (c->nlevels - l);
      left -= bits;
      c->lv[l].shift = left; c->lv[l].bits = bits; c->lv[l].nbuckets = nb; c->lv[l].hb = cfg->hb; c->lv[l].q_lo = c->q_lo;
      c->lv[l].nslots = c->nslots; c->lv[l].out32 = 0; c->lv[l].ablate = 0; c->lv[l].ng_log2 = 0; c->lv[l].slot_cap = 0;
      nb <<= bits;
    }
    c->lv[c->nlevels - 1].out32 = 1;
    // first level: one cursor per (digit, window group) instead of one per digit (ShkRpLevel::ng_log2)
    if (c->lv[0].bits >= 2 && c->lv[0].bits <= 7 && !getenv("SHK_RP_NO_GROUPS")) c->lv[0].ng_log2 = 3;
  }
  c->threads = cfg->threads_per_group ? cfg->threads_per_group : 512;
  if (c->threads < 64 || c->threads > 1024 || (c->threads & (c->threads - 1))) { delete c; return SHK_ERR_ARG; }
  c->hash_groups = cfg->hash_groups ? cfg->hash_groups : 2048;
  c->rounds_left = cfg->num_denoise;
  c->max_reads = cfg->max_batch_reads ? cfg->max_batch_reads : cfg->max_batch_bytes / 16 + 1024;
  const uint64_t capk = cfg->max_batch_keys;
  const uint32_t maxch = SHK_MAX_CHUNKS;
  for (int i = 0; i < 2; i++) {
    if (dmalloc(&c->tab[i], c->table_bytes)) return SHK_ERR_HIP;
    if (dmalloc(&c->fin[i], (uint64_t)c->nregions + 2)) return SHK_ERR_HIP;
    if (dmalloc(&c->d_words[i], capk + 1)) return SHK_ERR_HIP;
  }
  if (dmalloc(&c->d_text, cfg->max_batch_bytes + 64)) return SHK_ERR_HIP;
  if (dmalloc(&c->d_chunk_off, maxch) || dmalloc(&c->d_chunk_len, maxch) || dmalloc(&c->d_nlines, (uint64_t)maxch * SHK_PARSE_SEGS) ||
      dmalloc(&c->d_reads_base, maxch + 1)) return SHK_ERR_HIP;
  if (dmalloc(&c->d_rd_start, c->max_reads + 1) || dmalloc(&c->d_rd_end, c->max_reads + 1) ||
      dmalloc(&c->d_nkeys, c->max_reads + 1) || dmalloc(&c->d_rd_chunk, c->max_reads + 1) || dmalloc(&c->d_key_base, c->max_reads + 2)) return SHK_ERR_HIP;
  if (dmalloc(&c->d_scalars, 64)) return SHK_ERR_HIP;
  HIPCHK(hipMemsetAsync(c->d_scalars, 0, 64 * 8, c->stream));
  {
    uint64_t mx = capk > c->max_reads ? capk : c->max_reads;
    uint64_t pw = 1ULL << c->rbits;
    if (pw > mx) mx = pw;
    if (dmalloc(&c->d_block_sums, mx / SHK_SCAN_TILE + 4 + 8192)) return SHK_ERR_HIP;
  }
  {
    uint64_t nb = 1;
    if (dmalloc(&c->d_base[0], 2)) return SHK_ERR_HIP;
    for (uint32_t l = 0; l < c->nlevels; l++) {
      uint64_t n = nb << c->lv[l].bits;
      if (dmalloc(&c->d_hist[l], (n << c->lv[l].ng_log2) + 1) || dmalloc(&c->d_base[l + 1], n + 2)) return SHK_ERR_HIP;
      if (l == 0 && dmalloc(&c->d_base_sub, (n << c->lv[0].ng_log2) + 2)) return SHK_ERR_HIP;
      nb = n;
    }
    { const uint64_t first = (1ULL << (c->lv[0].bits + c->lv[0].ng_log2));
      if (dmalloc(&c->d_cursor, (nb > first ? nb : first) + 2)) return SHK_ERR_HIP; }
  }
  if (dmalloc(&c->d_tfb, capk / SHK_RP_TILE + 2)) return SHK_ERR_HIP;
  if (dmalloc(&c->d_summary, SHK_SUM_STRIDE * (uint64_t)c->nregions + 8)) return SHK_ERR_HIP;
  if (dmalloc(&c->d_lb_agg, (uint64_t)c->nregions + 2) || dmalloc(&c->d_lb_incl, (uint64_t)c->nregions + 2)) return SHK_ERR_HIP;
  c->single_ok = getenv("SHK_SINGLE") ? 1 : 0;
  c->merge_group = SHK_MERGE_GROUP;
  if (const char *mg = getenv("SHK_MERGE_GROUP")) { int v = atoi(mg); if (v == 64 || v == 128) c->merge_group = (uint32_t)v; }
  c->use_spill = (getenv("SHK_TWO_LAUNCH") || c->single_ok) ? 0 : 1;
  // the sampled location of a deNoise point needs enough regions for the sample to mean something
  c->region_cap = 0; c->slot_overflows = 0; c->slots_off = getenv("SHK_NO_SLOTS") ? 1 : 0;
  c->stage2_b = nullptr; c->stage2_na = c->stage2_nb = 0;
  // every 8th region; every 16th from 2^20 regions on (qb >= 28): a wrong guess costs one more one-pass point (18 ms at
  // qb 29), the sample 1.9 / 1.15 / 0.75 ms at stride 8 / 16 / 32; measured on the 12 points of the qb-29 bench: no wrong
  // guess at 8 and 16, one at 32 (its chance grows with sqrt(stride) / sqrt(new keys per batch))
  c->sample_stride = c->nregions >= (1u << 20) ? 16 : 8;
  if (const char *e = getenv("SHK_SAMPLE_STRIDE")) c->sample_stride = (uint32_t)atoi(e);
  else if (c->nregions < (1u << 14)) c->sample_stride = 0;
  if (dmalloc(&c->d_spill, (uint64_t)c->nregions * SHK_SPILL_STRIDE) || dmalloc(&c->d_over_list, (uint64_t)c->nregions + 1)) return SHK_ERR_HIP;
  { uint64_t nt = c->nregions / SHK_RSCAN_TILE + 2;
    if (dmalloc(&c->d_tile_a, nt) || dmalloc(&c->d_tile_b, nt) || dmalloc(&c->d_tile_f, nt)) return SHK_ERR_HIP; }
  if (dmalloc(&c->d_counters, 4 + SHK_HIST_BINS + 4)) return SHK_ERR_HIP;
  if (dmalloc(&c->d_err, 4)) return SHK_ERR_HIP;
  HIPCHK(hipHostMalloc((void **)&c->h_pinned, 64 * sizeof(uint64_t), hipHostMallocDefault));
  HIPCHK(hipMemsetAsync(c->tab[0], 0, c->table_bytes + SHK_SLACK, c->stream));
  HIPCHK(hipMemsetAsync(c->tab[1], 0, c->table_bytes + SHK_SLACK, c->stream));
  HIPCHK(hipMemsetAsync(c->fin[0], 0, ((uint64_t)c->nregions + 2) * 8, c->stream));
  HIPCHK(hipMemsetAsync(c->fin[1], 0, ((uint64_t)c->nregions + 2) * 8, c->stream));
  HIPCHK(hipMemsetAsync(c->d_err, 0, 16, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  // A context that is going to take deNoise rounds (its own, or as a shard of a filter that does) gets the buffers of the
  // first-chunk records and of the one-pass point now: allocating gigabytes inside a counting call is a synchronous trip
  // into the driver in the middle of the build (contexts without rounds never pay for them)
  if ((cfg->num_denoise > 0 || cfg->num_shards > 1) && c->use_spill && !getenv("SHK_COARSE_HIST")) {
    if (ensure_chist(c) || point_alloc(c)) { shk_destroy(c); return SHK_ERR_HIP; }
  }
  if (cfg->num_shards > 1 && !getenv("SHK_ROUTE_SINGLE_BUFFER")) {     // the two send buffers of shk_route_words, for the same reason
    for (int b = 0; b < 2; b++)
      if (dmalloc(&c->d_send[b], c->cfg.max_batch_keys + 1)) { shk_destroy(c); return SHK_ERR_HIP; }
  }
  *out = c;
  return SHK_OK;
}

static void front_destroy(shk_ctx *c);
extern "C" void shk_destroy(shk_ctx *c) {
  if (!c) return;
  hipSetDevice(c->dev);
  front_destroy(c);
  hipStreamSynchronize(c->stream);
  if (c->copy_stream) {
    hipStreamSynchronize(c->copy_stream);
    for (int b = 0; b < 2; b++) { if (c->d_up[b]) hipFree(c->d_up[b]); if (c->up_done[b]) hipEventDestroy(c->up_done[b]); }
    hipStreamDestroy(c->copy_stream);
  }
  if (getenv("SHK_STAMPS")) {
    unsigned long long st[16];
    hipMemcpy(st, c->d_scalars + 16, sizeof(st), hipMemcpyDeviceToHost);
    static const char *nm[9] = {"stage+init", "fold keys", "old rank/select", "count sort", "merge pass", "scan+stats", "look-back", "placement", "stores"};
    unsigned long long tot = 0; for (int i = 0; i < 9; i++) tot += st[i];
    for (int i = 0; i < 9; i++) fprintf(stderr, "SHK_STAMPS %-16s %6.2f %%\n", nm[i], tot ? 100.0 * st[i] / tot : 0.0);
  }
  prof_collect(c);
  for (size_t i = 0; i < c->evpool.size(); i++) hipEventDestroy(c->evpool[i]);
  for (int i = 0; i < 2; i++) { hipFree(c->tab[i]); hipFree(c->fin[i]); hipFree(c->d_words[i]); if (c->d_send[i]) hipFree(c->d_send[i]); }
  hipFree(c->d_text); hipFree(c->d_chunk_off); hipFree(c->d_chunk_len); hipFree(c->d_nlines); hipFree(c->d_reads_base);
  hipFree(c->d_rd_start); hipFree(c->d_rd_end); hipFree(c->d_rd_chunk); hipFree(c->d_nkeys); hipFree(c->d_key_base); hipFree(c->d_scalars);
  hipFree(c->d_block_sums);
  hipFree(c->d_base[0]); hipFree(c->d_base_sub);
  for (uint32_t l = 0; l < c->nlevels; l++) { hipFree(c->d_hist[l]); hipFree(c->d_base[l + 1]); }
  if (c->d_isum) { hipFree(c->d_isum); hipFree(c->d_ilens); hipFree(c->d_fin_i); hipFree(c->d_prot); }
  hipFree(c->d_spill); hipFree(c->d_over_list); if (c->d_newchunks) { hipFree(c->d_newchunks); hipFree(c->d_chist); hipHostFree(c->h_chist); } hipFree(c->d_cursor); hipFree(c->d_tfb); hipFree(c->d_summary); hipFree(c->d_lb_agg); hipFree(c->d_lb_incl); hipFree(c->d_tile_a); hipFree(c->d_tile_b); hipFree(c->d_tile_f); hipFree(c->d_counters); hipFree(c->d_err);
  hipHostFree(c->h_pinned);
  hipStreamDestroy(c->stream);
  delete c;
}

// ------------------------------------------------------------------ helpers
// exclusive scan of in[0..n) (n on the host, or *n_dev on the device with n_max as bound)
template <typename T>
static int run_scan(shk_ctx *c, const T *in, uint64_t n_max, const uint64_t *n_dev, uint64_t *out, uint64_t *sums = nullptr) {
  // sums: n_max / SHK_SCAN_TILE + 2 words of scratch; the context's own fits the batch sizes it was created for
  ProfScope ps(c, KP_SCAN);
  if (!sums) sums = c->d_block_sums;
  const uint32_t nb = (uint32_t)(n_max / SHK_SCAN_TILE + 1);
  hipLaunchKernelGGL((k_scan_reduce<T>), dim3(nb), dim3(c->threads), 0, c->stream, in, n_max, n_dev, sums);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(c->threads), 0, c->stream, sums, (uint64_t)nb, c->d_scalars + 2);
  hipLaunchKernelGGL((k_scan_apply<T>), dim3(nb), dim3(c->threads), 0, c->stream, in, n_max, n_dev, sums,
                     c->d_scalars + 2, out);
  HIPCHK(hipGetLastError());
  return 0;
}

static int fetch_err(shk_ctx *c, uint32_t *bits) {
  HIPCHK(hipMemcpyAsync(c->h_pinned + 40, c->d_err, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  *bits = *(uint32_t *)(c->h_pinned + 40);
  if (*bits) c->last_err_bits = *bits;
  if (*bits) hipMemsetAsync(c->d_err, 0, 16, c->stream);
  return 0;
}

// text + chunk table -> key words in d_words[0]; d_scalars[1] = #words
// text -> extents of every read (d_rd_start, d_rd_end, d_rd_chunk); *dtext_out = where the text is on the device
static int parse_stage(shk_ctx *c, const void *text, int on_device, uint64_t text_bytes, const uint64_t *chunk_off,
                       const uint64_t *chunk_len, uint32_t nchunks, const uint8_t **dtext_out, uint64_t *nreads_out) {
  for (uint32_t i = 0; i < nchunks; i++)
    if (chunk_off[i] + chunk_len[i] > text_bytes) return SHK_ERR_ARG;
  const uint8_t *dtext;
  if (on_device) {
    dtext = (const uint8_t *)text;
    for (int b = 0; b < 2; b++)   // a buffer of shk_upload_text whose copy may still be running
      if (c->d_up[b] && dtext == c->d_up[b] && c->up_pending[b]) { HIPCHK(hipStreamWaitEvent(c->stream, c->up_done[b], 0)); c->up_pending[b] = 0; }
  } else {
    if (text_bytes > c->cfg.max_batch_bytes) return SHK_ERR_BATCH;
    HIPCHK(hipMemcpyAsync(c->d_text, text, text_bytes, hipMemcpyHostToDevice, c->stream));
    dtext = c->d_text;
  }
  HIPCHK(hipMemcpyAsync(c->d_chunk_off, chunk_off, nchunks * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_chunk_len, chunk_len, nchunks * 8, hipMemcpyHostToDevice, c->stream));
  { ProfScope ps(c, KP_COUNT_LINES);
    hipLaunchKernelGGL(k_count_lines, dim3(nchunks * SHK_PARSE_SEGS), dim3(c->threads), 0, c->stream, dtext, c->d_chunk_off, c->d_chunk_len, c->d_nlines); }
  { ProfScope ps(c, KP_SCAN_CHUNKS);
    hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(c->threads), 0, c->stream, c->d_nlines, nchunks, c->d_reads_base, c->d_scalars + 0); }
  // the read arrays are sized by max_reads: the count is checked on the host below
  HIPCHK(hipMemcpyAsync(c->h_pinned + 41, c->d_scalars, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  const uint64_t nreads = c->h_pinned[41];
  if (nreads > c->max_reads) return SHK_ERR_BATCH;
  { ProfScope ps(c, KP_EMIT_READS);
    hipLaunchKernelGGL(k_emit_reads, dim3(nchunks * SHK_PARSE_SEGS), dim3(c->threads), 0, c->stream, dtext, c->d_chunk_off, c->d_chunk_len,
                       c->d_reads_base, c->d_nlines, c->d_rd_start, c->d_rd_end, c->d_rd_chunk); }
  *dtext_out = dtext;
  *nreads_out = nreads;
  return SHK_OK;
}

static int hash_stage(shk_ctx *c, const void *text, int on_device, uint64_t text_bytes, const uint64_t *chunk_off,
                      const uint64_t *chunk_len, uint32_t nchunks, uint32_t chunk_first, uint32_t chunk_mul, bool hist0 = false) {
  // chunk i of this call is labelled chunk_first + i * chunk_mul; hist0: the hash kernel also fills the first partition
  // level's histogram (d_hist[0]), see partition_stage
  if (hist0) HIPCHK(hipMemsetAsync(c->d_hist[0], 0, (1ULL << (c->lv[0].bits + c->lv[0].ng_log2)) * 8, c->stream));
  if (nchunks == 0 || nchunks > SHK_MAX_CHUNKS || chunk_first + (uint64_t)(nchunks - 1) * chunk_mul >= SHK_MAX_CHUNKS) return SHK_ERR_BATCH;
  const uint8_t *dtext;
  uint64_t nreads;
  { int rc = parse_stage(c, text, on_device, text_bytes, chunk_off, chunk_len, nchunks, &dtext, &nreads); if (rc) return rc; }
  uint32_t groups = c->hash_groups;
  { uint64_t need = nreads / (c->threads / SHK_WAVE) + 1; if (need < groups) groups = (uint32_t)need; }
  { ProfScope ps(c, KP_COUNT_KEYS);       // (one thread per read)
    const uint64_t blocks = nreads / 256 + 1;
    hipLaunchKernelGGL(k_count_keys, dim3((uint32_t)(blocks < (1u << 20) ? blocks : (1u << 20))), dim3(256), 0, c->stream, dtext, c->d_rd_start, c->d_rd_end,
                       c->d_scalars + 0, c->cfg.k, c->d_nkeys, c->d_err); }
  if (run_scan<uint32_t>(c, c->d_nkeys, nreads, nullptr, c->d_key_base)) return SHK_ERR_HIP;
  // total = key_base[nreads] -> d_scalars[1]
  HIPCHK(hipMemcpyAsync(c->d_scalars + 1, c->d_key_base + nreads, 8, hipMemcpyDeviceToDevice, c->stream));
  { ProfScope ps(c, KP_HASH);
    const uint32_t ht = c->threads < SHK_HASH_WAVES * SHK_WAVE ? c->threads : SHK_HASH_WAVES * SHK_WAVE;
    hipLaunchKernelGGL(k_hash_reads, dim3(groups * (c->threads / ht)), dim3(ht), 0, c->stream, dtext, c->d_rd_start, c->d_rd_end,
                       c->d_scalars + 0, c->d_rd_chunk, chunk_first, chunk_mul, c->d_key_base, c->cfg.k, c->cfg.hb,
                       c->d_words[0], c->cfg.max_batch_keys, c->d_err, hist0 ? c->d_hist[0] : nullptr, c->lv[0].shift, c->lv[0].bits, c->q_lo, c->lv[0].ng_log2); }
  HIPCHK(hipGetLastError());
  return SHK_OK;
}

// 2-bit staging of the batch's reads for the roll kernels (k_pack_reads), in `buf` = a buffer of max_batch_keys words that
// nothing else uses until the roll kernels are done. Leaves A.pk null (text path for every read) when the buffer cannot
// hold the batch's units or SHK_NO_PACK is set (measurement).
static int pack_stage(shk_ctx *c, ShkRollArgs &A, uint64_t nreads, uint64_t text_bytes, uint64_t *buf) {
  A.pk = nullptr; A.pk_base = nullptr; A.pk_flag = nullptr;
  // units <= sum over reads of (len / 64 + 1) <= text_bytes / 64 + nreads when no two chunks overlap (k_pack_reads leaves
  // the reads that do not fit on the text path); 16 bytes of slack for the roll kernels' 16-byte fetches
  const uint64_t cap_units = c->cfg.max_batch_keys / 2 > 1 ? c->cfg.max_batch_keys / 2 - 1 : 0;
  if (getenv("SHK_NO_PACK") || text_bytes / 64 + nreads + 1 > cap_units) return SHK_OK;
  ProfScope ps(c, KP_PACK);
  const uint32_t t = c->threads < 256 ? c->threads : 256;
  { const uint64_t blocks = nreads / t + 1;
    hipLaunchKernelGGL(k_pack_count, dim3((uint32_t)(blocks < 4096 ? blocks : 4096)), dim3(t), 0, c->stream, (const uint64_t *)c->d_rd_start,
                       (const uint64_t *)c->d_rd_end, (const uint64_t *)(c->d_scalars + 0), c->cfg.k, c->d_nkeys); }
  if (run_scan<uint32_t>(c, c->d_nkeys, nreads, nullptr, c->d_key_base)) return SHK_ERR_HIP;
  { const uint64_t blocks = nreads * 4 / t + 1;
    hipLaunchKernelGGL(k_pack_reads, dim3((uint32_t)(blocks < 16384 ? blocks : 16384)), dim3(t), 0, c->stream, A.text, A.safe_end,
                       (const uint64_t *)c->d_rd_start, (const uint64_t *)c->d_rd_end, (const uint64_t *)(c->d_scalars + 0),
                       (const uint64_t *)c->d_key_base, c->d_nkeys, (ShkQuad *)buf, cap_units); }
  A.pk = (const ShkQuad *)buf; A.pk_base = c->d_key_base; A.pk_flag = c->d_nkeys;
  return SHK_OK;
}

// text + chunk table -> key words in d_words[0], partitioned by the first region digit; d_base[1] = bucket bases,
// d_scalars[1] = #words (roll_kernels.hip). For contexts with at least two partition levels.
static bool roll_path(const shk_ctx *c) { return c->nlevels >= 2 && (c->q_lo & (SHK_REGION - 1)) == 0 && !getenv("SHK_NO_ROLL"); }
static int roll_stage(shk_ctx *c, const void *text, int on_device, uint64_t text_bytes, const uint64_t *chunk_off,
                      const uint64_t *chunk_len, uint32_t nchunks, uint32_t chunk_first, uint32_t chunk_mul, bool *level1_hist_ready) {
  if (nchunks == 0 || nchunks > SHK_MAX_CHUNKS || chunk_first + (uint64_t)(nchunks - 1) * chunk_mul >= SHK_MAX_CHUNKS) return SHK_ERR_BATCH;
  const uint8_t *dtext;
  uint64_t nreads;
  { int rc = parse_stage(c, text, on_device, text_bytes, chunk_off, chunk_len, nchunks, &dtext, &nreads); if (rc) return rc; }
  const uint64_t P = 1ULL << c->lv[0].bits;
  // the first two levels' digits together, when they fit the histogram pass's LDS bins
  const uint32_t cb = c->lv[0].bits + c->lv[1].bits;
  const bool two = cb <= 14 && !getenv("SHK_ROLL_HIST1");
  *level1_hist_ready = two;
  if (two) HIPCHK(hipMemsetAsync(c->d_hist[1], 0, (1ULL << cb) * 8, c->stream));
  else HIPCHK(hipMemsetAsync(c->d_hist[0], 0, P * 8, c->stream));
  ShkRollArgs A;
  A.text = dtext; A.safe_end = (text_bytes + 15) & ~15ULL;
  A.rd_start = c->d_rd_start; A.rd_end = c->d_rd_end; A.nreads_p = c->d_scalars + 0; A.rd_chunk = c->d_rd_chunk;
  A.chunk_first = chunk_first; A.chunk_mul = chunk_mul; A.k = c->cfg.k; A.hb = c->cfg.hb; A.q_lo = c->q_lo;
  A.dig_shift = c->lv[0].shift; A.dig_bits = c->lv[0].bits;
  A.hist = two ? c->d_hist[1] : c->d_hist[0]; A.hist_shift = two ? c->lv[1].shift : c->lv[0].shift; A.hist_bits = two ? cb : c->lv[0].bits;
  A.cursor = c->d_cursor; A.out = c->d_words[0]; A.cap = c->cfg.max_batch_keys; A.err = c->d_err;
  { int rc = pack_stage(c, A, nreads, text_bytes, c->d_words[1]); if (rc) return rc; }     // (d_words[1]: the partition's other buffer, idle until its second level)
  { ProfScope ps(c, KP_ROLL_HIST);
    if (two && c->threads >= 512) {
      const uint64_t blocks = nreads / 512 + 1;
      hipLaunchKernelGGL((k_roll_hist<14, 512>), dim3((uint32_t)(blocks < 1024 ? blocks : 1024)), dim3(512), 0, c->stream, A);
    } else if (two) {     // (small workgroups: the CPU emulator build of the tests)
      const uint64_t blocks = nreads / 64 + 1;
      hipLaunchKernelGGL((k_roll_hist<14, 64>), dim3((uint32_t)(blocks < 64 ? blocks : 64)), dim3(64), 0, c->stream, A);
    } else {
      const uint64_t blocks = nreads / 256 + 1;
      hipLaunchKernelGGL((k_roll_hist<10, 256>), dim3((uint32_t)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, c->stream, A);
    }
    if (two) hipLaunchKernelGGL(k_roll_fold, dim3((uint32_t)(P / 256 + 1)), dim3(256), 0, c->stream, (const uint64_t *)c->d_hist[1], (uint32_t)P,
                                1u << c->lv[1].bits, c->d_hist[0]); }
  // bucket bases = exclusive scan of the digit counts; its total is the number of key words
  if (run_scan<uint64_t>(c, c->d_hist[0], P, nullptr, c->d_base[1])) return SHK_ERR_HIP;
  HIPCHK(hipMemcpyAsync(c->d_scalars + 1, c->d_base[1] + P, 8, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_cursor, c->d_base[1], P * 8, hipMemcpyDeviceToDevice, c->stream));
  { ProfScope ps(c, KP_ROLL_SCATTER);
    const char *nqe = getenv("SHK_ROLL_NQ");            // (measurement: 16-byte quads fetched at a time per stream)
    const int nq = nqe ? atoi(nqe) : 1;
    const char *te = getenv("SHK_ROLL_T");             // (measurement: threads per workgroup = keys per window / 16)
    const int rt = te ? atoi(te) : 1024;
    if (c->threads >= 512 && rt == 512) {
      const uint64_t blocks = nreads / 512 + 1;
      hipLaunchKernelGGL((k_roll_scatter<512, 1>), dim3((uint32_t)(blocks < 1024 ? blocks : 1024)), dim3(512), 0, c->stream, A);
    } else if (c->threads >= 512 && rt == 256) {
      const uint64_t blocks = nreads / 256 + 1;
      hipLaunchKernelGGL((k_roll_scatter<256, 1>), dim3((uint32_t)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, c->stream, A);
    } else if (c->threads >= 512) {
      const uint64_t blocks = nreads / 1024 + 1;
      const dim3 grid((uint32_t)(blocks < 512 ? blocks : 512));
      if (nq == 1) hipLaunchKernelGGL((k_roll_scatter<1024, 1>), grid, dim3(1024), 0, c->stream, A);
      else hipLaunchKernelGGL((k_roll_scatter<1024, 2>), grid, dim3(1024), 0, c->stream, A);
    } else {          // (small workgroups: the CPU emulator build of the tests)
      const uint64_t blocks = nreads / 64 + 1;
      hipLaunchKernelGGL((k_roll_scatter<64, 4>), dim3((uint32_t)(blocks < 64 ? blocks : 64)), dim3(64), 0, c->stream, A);
    } }
  HIPCHK(hipGetLastError());
  return SHK_OK;
}

// words in d_words[src] (count in d_scalars[1], bound nmax) -> sorted by region in
// d_words[*dst]; region offsets in d_base[nlevels]
// (ext != null: the first level reads the caller's buffer instead of d_words[src])
// first_level = 1: the words in d_words[src] are partitioned by the first digit already and d_base[1] holds the bucket
// bases (roll_stage)
static int partition_stage(shk_ctx *c, int src, uint64_t nmax, int *dst, const uint64_t *ext = nullptr, bool hist0_ready = false,
                           uint32_t first_level = 0, bool hist1_ready = false) {
  const uint64_t *n_p = c->d_scalars + 1;
  { ProfScope ps(c, KP_RP_PREP);
    hipLaunchKernelGGL(k_rp_base1, dim3(1), dim3(64), 0, c->stream, n_p, c->d_base[0]); }
  const uint32_t nwin = (uint32_t)(nmax / SHK_RP_TILE + 1);
  const uint64_t *in = ext ? ext : c->d_words[src];
  int cur = ext ? 1 : src;   // the buffer `in` occupies (an external source leaves both free: write to d_words[0] first)
  c->region_cap = 0;
  // shk_stage_words_pair: the first level reads TWO buffers (the words a shard kept for itself, where the routing left them,
  // and the words it received): both are counted into one histogram and scattered with one set of cursors. Their
  // lengths and one-bucket base arrays live in d_scalars[40..45].
  const uint64_t *in2 = (ext && first_level == 0) ? c->stage2_b : nullptr;
  const uint64_t n_a = in2 ? c->stage2_na : nmax, n_b = in2 ? c->stage2_nb : 0;
  for (uint32_t l = first_level; l < c->nlevels; l++) {
    const uint64_t nb = c->lv[l].nbuckets, P = 1ULL << c->lv[l].bits;
    // the sources of this level: (words, their number on the device, their bucket bases, their number on the host)
    struct Src { const uint64_t *w, *n_p, *base; uint64_t n; } srcs[2] = {{in, n_p, c->d_base[l], nmax}, {nullptr, nullptr, nullptr, 0}};
    int nsrc = 1;
    if (l == 0 && in2) {
      srcs[0] = {in, c->d_scalars + 40, c->d_scalars + 42, n_a};
      srcs[1] = {in2, c->d_scalars + 41, c->d_scalars + 44, n_b};
      nsrc = 2;
    }
    // Last level: fixed-capacity region slots instead of a histogram pass over the keys + scan, when the output buffer
    // (max_batch_keys 8-byte words = twice as many 4-byte records) gives every region room for its mean share of this
    // batch plus six sigma of a clumpy hash distribution (a true k-mer comes ~8 times per batch). A region that gets
    // more (repeats: one k-mer a million times) raises SHK_E_SLOT_FULL and the level is redone the exact way; after two
    // such batches in a row the context stops trying.
    uint32_t cap = 0;
    if (l + 1 == c->nlevels && l >= 1 && c->lv[l].out32 && !c->slots_off && !(l == 1 && hist1_ready)) {
      uint64_t cp = 2 * c->cfg.max_batch_keys / (nb * P);
      if (cp > (1u << 20)) cp = 1u << 20;
      const double mean = (double)nmax / (double)(nb * P);
      if (cp >= 64 && (double)cp >= mean + 6.0 * sqrt(8.0 * mean + 1.0) + 16.0) cap = (uint32_t)cp;
    }
    { ProfScope ps(c, KP_RP_PREP);
      hipLaunchKernelGGL(k_rp_tile_first, dim3(nwin / 256 + 1), dim3(256), 0, c->stream, c->d_base[l], (uint32_t)nb, n_p, c->d_tfb); }
    for (;;) {
      ShkRpLevel lvl = c->lv[l];
#ifdef SHK_DIAGNOSTICS   // timing ablations give INVALID results: compiled into diagnostic builds only (make DIAG=1)
      if (const char *e = getenv("SHK_RP_ABLATE")) lvl.ablate = (uint32_t)atoi(e);
#endif
      uint64_t *cursor = c->d_cursor;
      if (cap) {
        ProfScope ps(c, KP_RP_SLOTS);
        cursor = c->d_base[l + 1];           // (ends up as the regions' end positions)
        lvl.slot_cap = cap;
        hipLaunchKernelGGL(k_rp_slot_cursors, dim3((uint32_t)((nb * P) / 256 + 1 < 4096 ? (nb * P) / 256 + 1 : 4096)), dim3(256), 0, c->stream, cursor, nb * P, cap);
      } else {
        const bool ready = (l == 0 && hist0_ready) || (l == 1 && hist1_ready);
        if (!ready && l == 0 && c->nlevels >= 2 && c->lv[0].bits + c->lv[1].bits <= 14 && c->lv[1].ng_log2 == 0 && !getenv("SHK_RP_HIST1")) {
          // the first pass over unsorted words counts the second level's digits as well (k_rp_hist2)
          HIPCHK(hipMemsetAsync(c->d_hist[0], 0, (P << c->lv[0].ng_log2) * 8, c->stream));
          HIPCHK(hipMemsetAsync(c->d_hist[1], 0, (P << c->lv[1].bits) * 8, c->stream));
          ProfScope ps(c, KP_RP_HIST);
          const uint32_t wt = nwin / 1024 + 1;   // windows per workgroup (few workgroups: each flushes up to 2^14 counters)
          for (int si = 0; si < nsrc; si++)
            hipLaunchKernelGGL((k_rp_hist2<14>), dim3(nwin / wt + 1), dim3(c->threads < 512 ? c->threads : 512), 0, c->stream, srcs[si].w, srcs[si].n_p,
                               c->lv[0], c->lv[1], c->d_hist[0], c->d_hist[1], wt);
          hist1_ready = true;
        } else if (!ready) {
          HIPCHK(hipMemsetAsync(c->d_hist[l], 0, ((nb * P) << c->lv[l].ng_log2) * 8, c->stream));
          ProfScope ps(c, KP_RP_HIST);
          const uint32_t wt = nwin / 4096 + 1;   // windows per workgroup
          for (int si = 0; si < nsrc; si++)
            hipLaunchKernelGGL(k_rp_hist, dim3(nwin / wt + 1), dim3(c->threads), 0, c->stream, srcs[si].w, srcs[si].n_p, srcs[si].base, c->d_tfb, c->lv[l],
                               c->d_hist[l], wt);
        }
        if (c->lv[l].ng_log2) {
          // (first level only: nb = 1) sub-buckets in (digit, group) order; the next level's buckets are the digits
          const uint32_t ng = c->lv[l].ng_log2;
          if (run_scan<uint64_t>(c, c->d_hist[l], P << ng, nullptr, c->d_base_sub)) return SHK_ERR_HIP;
          HIPCHK(hipMemcpyAsync(c->d_cursor, c->d_base_sub, (P << ng) * 8, hipMemcpyDeviceToDevice, c->stream));
          ProfScope ps(c, KP_RP_PREP);
          hipLaunchKernelGGL(k_rp_group_bases, dim3((uint32_t)(P / 256 + 1)), dim3(256), 0, c->stream, c->d_base_sub, (uint32_t)P, ng, c->d_base[l + 1]);
        } else {
          if (run_scan<uint64_t>(c, c->d_hist[l], nb * P, nullptr, c->d_base[l + 1])) return SHK_ERR_HIP;
          HIPCHK(hipMemcpyAsync(c->d_cursor, c->d_base[l + 1], nb * P * 8, hipMemcpyDeviceToDevice, c->stream));
        }
      }
      { ProfScope ps(c, KP_RP_SCATTER);
        for (int si = 0; si < nsrc; si++) {
          const Src &S = srcs[si];
          if (l == 0 && c->lv[0].ng_log2)      // (window groups are defined on the first level's 16384-key windows: SHK_RP_TILE0_LOG2)
            hipLaunchKernelGGL((k_rp_scatter<SHK_RP_TILE0_LOG2, 1024>), dim3((uint32_t)(S.n >> SHK_RP_TILE0_LOG2) + 1), dim3(1024), 0, c->stream, S.w,
                               c->d_words[cur ^ 1], S.n_p, S.base, c->d_tfb, lvl, cursor, c->d_err);
          else if (c->threads >= 512 && l + 1 < c->nlevels && c->lv[l].bits <= 8 && !getenv("SHK_RP_NARROW"))
            // a level in the middle: 16384-key windows as at the first level (digit runs of 1 KB instead of 256 bytes:
            // 2.85 -> 2.25 ms per 832 M keys). Not the last level: its 4-byte records in slots gain nothing (3.4 ms either way)
            hipLaunchKernelGGL((k_rp_scatter<SHK_RP_TILE0_LOG2, 1024, 256>), dim3((uint32_t)(S.n >> SHK_RP_TILE0_LOG2) + 1), dim3(1024), 0, c->stream, S.w,
                               c->d_words[cur ^ 1], S.n_p, S.base, c->d_tfb, lvl, cursor, c->d_err);
          else
            hipLaunchKernelGGL((k_rp_scatter<12, SHK_RP_THREADS>), dim3((uint32_t)(S.n / SHK_RP_TILE + 1)), dim3(SHK_RP_THREADS), 0, c->stream, S.w,
                               c->d_words[cur ^ 1], S.n_p, S.base, c->d_tfb, lvl, cursor, c->d_err);
        } }
      if (!cap) break;
      uint32_t bits = 0;
      if (fetch_err(c, &bits)) return SHK_ERR_HIP;
      if (bits & ~SHK_E_SLOT_FULL) return map_err_bits(bits & ~SHK_E_SLOT_FULL);
      if (!bits) { c->region_cap = cap; c->slot_overflows = 0; break; }
      if (++c->slot_overflows >= 2) c->slots_off = 1;
      cap = 0;                               // a region overflowed its slot: the same level again with exact bases
    }
    cur ^= 1;
    in = c->d_words[cur];
  }
  if (c->nlevels == 0) {
    // a single region: its keys are [0, n)
  }
  HIPCHK(hipGetLastError());
  *dst = cur;
  return SHK_OK;
}

// One launch per slice of at most 2^24 regions (a HIP grid holds fewer than 2^32 threads; a qb-33 filter has 2^25
// regions): ARGS must name a ShkMergeArgs variable `A`, whose r0 the loop sets.
#define SHK_REGION_SLICE (1u << 24)
#define SHK_FOR_REGION_SLICES(c, A, nblk) \
  for (uint32_t r0_ = 0, nblk = 0; r0_ < (c)->nregions && ((A).r0 = r0_, nblk = (c)->nregions - r0_ < SHK_REGION_SLICE ? (c)->nregions - r0_ : SHK_REGION_SLICE, true); r0_ += SHK_REGION_SLICE)

template <int MODE>
static void launch_merge(shk_ctx *c, const ShkMergeArgs &A0) {
  ShkMergeArgs A = A0;
  SHK_FOR_REGION_SLICES(c, A, nblk) {
    if (c->big_image)
      hipLaunchKernelGGL((k_region_merge<MODE, SHK_IMG_BLOCKS_BIG>), dim3(nblk), dim3(c->merge_group), 0, c->stream, A);
    else
      hipLaunchKernelGGL((k_region_merge<MODE, SHK_IMG_BLOCKS>), dim3(nblk), dim3(c->merge_group), 0, c->stream, A);
  }
}

struct MergeOut {
  uint64_t newd, added, removed, before;
  uint64_t hist[SHK_HIST_BINS];
  uint32_t err;
  int have_chist;               // c->h_chist[chunk] = new keys first seen in that chunk (exact)
};

static void fill_args(shk_ctx *c, ShkMergeArgs *A, const uint64_t *words, uint32_t lo, uint32_t hi, uint32_t hbase,
                      uint32_t hshift, int denoise, int want_hist = 0) {
  A->want_hist = want_hist;
  A->tabA = c->tab[c->cur]; A->tabB = c->tab[c->cur ^ 1];
  A->finA = c->fin[c->cur]; A->finB = c->fin[c->cur ^ 1];
  A->words = reinterpret_cast<const uint32_t *>(words); A->region_base = c->d_base[c->nlevels]; A->region_cap = c->region_cap;
  A->nslots = c->nslots; A->xnslots = c->xnslots; A->nblocks = c->nblocks; A->q_lo = c->q_lo; A->hb = c->cfg.hb;
  A->chunk_lo = lo; A->chunk_hi = hi; A->hist_base = hbase; A->hist_shift = hshift; A->denoise = denoise;
  A->ablate = 0;
#ifdef SHK_DIAGNOSTICS   // timing ablations give INVALID results: compiled into diagnostic builds only (make DIAG=1)
  { const char *ab = getenv("SHK_ABLATE"); A->ablate = ab ? (uint32_t)atoi(ab) : 0; }
#endif
  A->lb_agg = c->d_lb_agg; A->lb_incl = c->d_lb_incl;
  { const char *sp = getenv("SHK_STAMPS");    // diagnostics: "fused" = only the one-pass deNoise launches, "plain" = all the others, else all
    A->dbg = (sp && strcmp(sp, "fused") != 0) ? (unsigned long long *)(c->d_scalars + 16) : nullptr; }
  A->spill = c->d_spill; A->over_list = c->d_over_list; A->n_over = c->d_counters + 4 + SHK_HIST_BINS; A->list = nullptr;
  A->newchunks = nullptr; A->chist = nullptr;
  A->counted = c->counted;
  A->r0 = 0; A->rstride = 1;
  A->split = ~0u; A->isum = nullptr; A->ilens = nullptr; A->prot_list = nullptr; A->nprot = 0;
  A->summary = c->d_summary; A->counters = c->d_counters; A->hist = c->d_counters + 4; A->err = c->d_err;
}

// first request for the exact first-chunk histogram (contexts that never reach a deNoise point never pay for it)
static int ensure_chist(shk_ctx *c) {
  if (c->d_newchunks) return SHK_OK;
  if (dmalloc(&c->d_newchunks, (uint64_t)c->nregions * SHK_NC_CAP) || dmalloc(&c->d_chist, (uint64_t)SHK_MAX_CHUNKS)) return SHK_ERR_HIP;
  HIPCHK(hipHostMalloc((void **)&c->h_chist, SHK_MAX_CHUNKS * sizeof(uint64_t), hipHostMallocDefault));
  return SHK_OK;
}

// summary launch + free-pointer scan, then read the statistics back (one synchronisation)
static int merge_summary(shk_ctx *c, const uint64_t *words, uint32_t lo, uint32_t hi, uint32_t hbase, uint32_t hshift,
                         int denoise, MergeOut *o, int want_hist = 0, int spill = 0) {
  ShkMergeArgs A;
  fill_args(c, &A, words, lo, hi, hbase, hshift, denoise, want_hist);
  HIPCHK(hipMemsetAsync(c->d_counters, 0, (4 + SHK_HIST_BINS + 1) * 8, c->stream));
  spill = spill && c->use_spill;
  c->spill_valid = 0;
  if (want_hist == 2 && !getenv("SHK_COARSE_HIST")) { int rc = ensure_chist(c); if (rc) return rc; }
  const bool exact = want_hist == 2 && c->d_newchunks;
  if (exact) {   // (zeroed in front of the pass: regions whose record overflows add to the histogram themselves)
    A.newchunks = c->d_newchunks; A.chist = c->d_chist;
    HIPCHK(hipMemsetAsync(c->d_chist, 0, SHK_MAX_CHUNKS * 8, c->stream));
  }
  o->have_chist = 0;
  if (spill) { ProfScope ps(c, KP_MERGE_SPILL);
    launch_merge<3>(c, A); }
  else { ProfScope ps(c, KP_MERGE_SUM);
    launch_merge<0>(c, A); }
  { ProfScope ps(c, KP_REGION_SCAN);
    const uint32_t ntiles = (c->nregions + SHK_RSCAN_TILE - 1) / SHK_RSCAN_TILE;
    hipLaunchKernelGGL(k_region_scan_a, dim3(ntiles), dim3(c->threads), 0, c->stream, c->d_summary, c->nregions, c->d_tile_a, c->d_tile_b);
    hipLaunchKernelGGL(k_region_scan_b, dim3(1), dim3(c->threads), 0, c->stream, c->d_tile_a, c->d_tile_b, ntiles, c->d_tile_f);
    hipLaunchKernelGGL(k_region_scan_c, dim3(ntiles), dim3(c->threads), 0, c->stream, c->d_summary, c->nregions, c->d_tile_f,
                       c->xnslots, (uint32_t)(c->big_image ? SHK_IMG_BLOCKS_BIG * 64 : SHK_IMG_SLOTS), c->fin[c->cur ^ 1], c->d_counters, c->d_err); }
  if (exact) {
    ProfScope ps(c, KP_MISC);
    hipLaunchKernelGGL(k_chunk_hist, dim3((c->nregions + SHK_CHIST_REGIONS - 1) / SHK_CHIST_REGIONS), dim3(256), 0, c->stream,
                       c->d_newchunks, c->d_summary, c->nregions, c->d_chist);
    HIPCHK(hipMemcpyAsync(c->h_chist, c->d_chist, ((uint64_t)hi + 1) * 8, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_pinned, c->d_counters, (4 + SHK_HIST_BINS + 1) * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_pinned + 40, c->d_err, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  o->newd = c->h_pinned[0]; o->added = c->h_pinned[1]; o->removed = c->h_pinned[2]; o->before = c->h_pinned[3];
  for (int i = 0; i < SHK_HIST_BINS; i++) o->hist[i] = c->h_pinned[4 + i];
  o->err = *(uint32_t *)(c->h_pinned + 40);
  if (o->err) c->last_err_bits = o->err;
  if (o->err) HIPCHK(hipMemsetAsync(c->d_err, 0, 16, c->stream));
  o->have_chist = exact ? 1 : 0;
  c->chist_n = exact ? hi + 1 : 0;
  if (spill && !o->err) {
    c->spill_valid = 1; c->spill_words = words; c->spill_lo = lo; c->spill_hi = hi; c->spill_denoise = denoise;
    c->spill_big = c->big_image; c->spill_nover = c->h_pinned[4 + SHK_HIST_BINS];
  }
  return SHK_OK;
}

// write launch for the summary that was just computed; then flip the live table
static int merge_write(shk_ctx *c, const uint64_t *words, uint32_t lo, uint32_t hi, int denoise) {
  ShkMergeArgs A;
  fill_args(c, &A, words, lo, hi, 0, 0, denoise);
  HIPCHK(hipMemsetAsync(c->tab[c->cur ^ 1], 0, c->table_bytes, c->stream));
  if (c->spill_valid && c->spill_words == words && c->spill_lo == lo && c->spill_hi == hi && c->spill_denoise == denoise &&
      c->spill_big == c->big_image) {
    // the summary launch left lengths and encodings behind: placement only
    { ProfScope ps(c, KP_PLACE);
      SHK_FOR_REGION_SLICES(c, A, nblk) {
        if (c->big_image) hipLaunchKernelGGL((k_region_place<SHK_IMG_BLOCKS_BIG>), dim3(nblk), dim3(SHK_WAVE), 0, c->stream, A);
        else hipLaunchKernelGGL((k_region_place<SHK_IMG_BLOCKS>), dim3(nblk), dim3(SHK_WAVE), 0, c->stream, A);
      }
      A.r0 = 0; }
    if (c->spill_nover) {
      A.list = c->d_over_list;
      ProfScope ps(c, KP_MERGE_WRITE);
      if (c->big_image)
        hipLaunchKernelGGL((k_region_merge<1, SHK_IMG_BLOCKS_BIG>), dim3((uint32_t)c->spill_nover), dim3(SHK_MERGE_GROUP), 0, c->stream, A);
      else
        hipLaunchKernelGGL((k_region_merge<1, SHK_IMG_BLOCKS>), dim3((uint32_t)c->spill_nover), dim3(SHK_MERGE_GROUP), 0, c->stream, A);
    }
  } else {
    ProfScope ps(c, KP_MERGE_WRITE);
    launch_merge<1>(c, A);
  }
  c->spill_valid = 0;
  HIPCHK(hipGetLastError());
  c->cur ^= 1;
  return SHK_OK;
}

// Single-launch rebuild: statistics and table B in one pass (free pointers by look-back).
// The live table is NOT flipped here; the caller commits with commit_single() once it has
// looked at the statistics (a deNoise point inside the range means the pass is discarded).
static int merge_single(shk_ctx *c, const uint64_t *words, uint32_t lo, uint32_t hi, int denoise, MergeOut *o,
                        int want_hist = 0, uint32_t hbase = 0, uint32_t hshift = 0) {
  ShkMergeArgs A;
  fill_args(c, &A, words, lo, hi, hbase, hshift, denoise, want_hist);
  HIPCHK(hipMemsetAsync(c->d_counters, 0, (4 + SHK_HIST_BINS) * 8, c->stream));
  HIPCHK(hipMemsetAsync(c->d_lb_agg, 0, ((uint64_t)c->nregions + 2) * 8, c->stream));
  HIPCHK(hipMemsetAsync(c->d_lb_incl, 0, ((uint64_t)c->nregions + 2) * 8, c->stream));
  HIPCHK(hipMemsetAsync(c->tab[c->cur ^ 1], 0, c->table_bytes, c->stream));
  c->spill_valid = 0;
  o->have_chist = 0;
  c->chist_n = 0;
  { ProfScope ps(c, KP_MERGE_SINGLE);
    launch_merge<2>(c, A); }
  { ProfScope ps(c, KP_REGION_SCAN);
    const uint32_t ntiles = (c->nregions + SHK_RSCAN_TILE - 1) / SHK_RSCAN_TILE;
    hipLaunchKernelGGL(k_stats_reduce, dim3(ntiles), dim3(c->threads), 0, c->stream, c->d_summary, c->nregions, c->d_counters); }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_pinned, c->d_counters, (4 + SHK_HIST_BINS) * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_pinned + 40, c->d_err, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  o->newd = c->h_pinned[0]; o->added = c->h_pinned[1]; o->removed = c->h_pinned[2]; o->before = c->h_pinned[3];
  for (int i = 0; i < SHK_HIST_BINS; i++) o->hist[i] = c->h_pinned[4 + i];
  o->err = *(uint32_t *)(c->h_pinned + 40);
  if (o->err) c->last_err_bits = o->err;
  if (o->err) HIPCHK(hipMemsetAsync(c->d_err, 0, 16, c->stream));
  if (o->err & SHK_E_LOOKBACK) c->single_ok = 0;
  return SHK_OK;
}
static void commit_single(shk_ctx *c) { c->cur ^= 1; }

static int denoise_round_once(shk_ctx *c, uint64_t *removed) {
  uint64_t ml = c->cfg.min_denoise_len ? c->cfg.min_denoise_len : (1ULL << 20);
  { ProfScope ps(c, KP_MARKS);
    hipLaunchKernelGGL(k_denoise_marks, dim3(1), dim3(64), 0, c->stream, c->tab[c->cur], c->nslots, c->xnslots, c->nblocks,
                       ml, (unsigned long long *)(c->d_scalars + 3)); }
  MergeOut o;
  int rc;
  bool done = false;
  if (c->single_ok) {
    rc = merge_single(c, nullptr, 0, 0, 1, &o);
    if (rc) return rc;
    if (!o.err) { commit_single(c); done = true; }
    else if (!(o.err & SHK_E_LOOKBACK)) return map_err_bits(o.err);
  }
  if (!done) {
    rc = merge_summary(c, nullptr, 0, 0, 0, 0, 1, &o, 0, 1);
    if (rc) return rc;
    if (o.err) return map_err_bits(o.err);
    rc = merge_write(c, nullptr, 0, 0, 1);
    if (rc) return rc;
  }
  c->nelts -= o.removed;        // CQF_mt.h:1037-1038
  c->ndistinct -= o.removed;
  *removed = o.removed;
  return SHK_OK;
}

// deNoise round fused with the insertion of the chunks behind the deNoise point (one pass over the
// table instead of two). Not taken (*done = false) when the pass reports anything unusual or the
// trigger would be reached again inside [lo, hi]: the caller then runs the plain round.
static int denoise_with_rest(shk_ctx *c, const uint64_t *words, uint32_t lo, uint32_t hi, shk_batch_stats *st, bool *done) {
  *done = false;
  if (!c->use_spill) return SHK_OK;
  uint64_t ml = c->cfg.min_denoise_len ? c->cfg.min_denoise_len : (1ULL << 20);
  { ProfScope ps(c, KP_MARKS);
    hipLaunchKernelGGL(k_denoise_marks, dim3(1), dim3(64), 0, c->stream, c->tab[c->cur], c->nslots, c->xnslots, c->nblocks,
                       ml, (unsigned long long *)(c->d_scalars + 3)); }
  MergeOut o;
  int rc = merge_summary(c, words, lo, hi, lo, 0, 1, &o, 0, 1);
  if (rc) return rc;
  if (o.err) return SHK_OK;
  if (c->rounds_left > 0 && c->ndistinct - o.removed + o.newd >= c->cfg.ndistinct_for_denoise) return SHK_OK;
  rc = merge_write(c, words, lo, hi, 1);
  if (rc) return rc;
  c->nelts = c->nelts - o.removed + o.added;        // CQF_mt.h:1037-1038, then the inserts
  c->ndistinct = c->ndistinct - o.removed + o.newd;
  st->removed += o.removed; st->denoise_rounds++;
  st->kmers += o.added; st->new_distinct += o.newd; st->chunks += hi - lo + 1;
  c->big_image = 0;
  *done = true;
  return SHK_OK;
}

// ONE pass for a deNoise point inside a batch (the three-pass form: rebuild the chunks up to the point, mark, fused round +
// rest). The hash keeps two counts per key -- occurrences in the chunks <= cstar and behind it -- so a lane knows every
// key's count at the moment the round runs (cb) and what arrives afterwards (ca): the entry survives with cb when cb >= 2,
// is dropped when cb == 1, and ca is added on top. What the round's range walk needs of the table in between (which never
// exists in memory) leaves the same pass as 8 bytes + 256 length bytes per region; k_denoise_marks_virtual walks those.
// The few singletons the walk protects (one-slot clusters on a range end) are put back by rebuilding their regions with
// the list. Anything unusual (long runs, a cluster beyond the LDS image, a second crossing inside the rest) -> the caller
// takes the three-pass path.
// Three steps, shared by the single-table flow (denoise_fused) and the sharded one (shk_stage_point_*):
//   point_try    the FUSED pass over all regions + both free-pointer scans (+ the exact first-chunk histogram)
//   point_walk   the range walk over the intermediate layout -> protected singletons of this table / shard
//   point_finish their regions once more with the list; final statistics; the spill records are then ready for placement
#define SHK_PROT_CAP 65536u
struct PointOut {
  uint64_t newd_after, added_after, removed, added_before;   // statistics (CQF_mt.h:1037-1038 bookkeeping)
  uint32_t err;                                              // kernel flags: anything set = not this way
  uint64_t islots, ifin;      // intermediate table: slots in use, free pointer behind the last region (local, carry 0)
  int first_used;             // intermediate table: quotient 0 has a run
};

static int point_alloc(shk_ctx *c) {
  if (c->d_isum) return SHK_OK;
  if (dmalloc(&c->d_isum, 2 * (uint64_t)c->nregions + 2) || dmalloc(&c->d_ilens, (uint64_t)c->nregions * SHK_REGION) ||
      dmalloc(&c->d_fin_i, (uint64_t)c->nregions + 2) || dmalloc(&c->d_prot, (uint64_t)SHK_PROT_CAP)) return SHK_ERR_HIP;
  return SHK_OK;
}

static void point_scans(shk_ctx *c, bool final_table, bool inter_table, long long carry) {
  const uint32_t ntiles = (c->nregions + SHK_RSCAN_TILE - 1) / SHK_RSCAN_TILE;
  const uint32_t img_slots = (uint32_t)SHK_IMG_SLOTS;
  ProfScope ps(c, KP_REGION_SCAN);
  if (final_table) {
    hipLaunchKernelGGL(k_region_scan_a, dim3(ntiles), dim3(c->threads), 0, c->stream, c->d_summary, c->nregions, c->d_tile_a, c->d_tile_b, (uint32_t)SHK_SUM_STRIDE);
    hipLaunchKernelGGL(k_region_scan_b, dim3(1), dim3(c->threads), 0, c->stream, c->d_tile_a, c->d_tile_b, ntiles, c->d_tile_f, 0LL);
    hipLaunchKernelGGL(k_region_scan_c, dim3(ntiles), dim3(c->threads), 0, c->stream, c->d_summary, c->nregions, c->d_tile_f,
                       c->xnslots, img_slots, c->fin[c->cur ^ 1], c->d_counters, c->d_err, (uint32_t)SHK_SUM_STRIDE);
  }
  if (inter_table) {
    // the table in between: free pointers at the region starts (its capacity flags count like the final table's)
    hipLaunchKernelGGL(k_region_scan_a, dim3(ntiles), dim3(c->threads), 0, c->stream, c->d_isum, c->nregions, c->d_tile_a, c->d_tile_b, 2u);
    hipLaunchKernelGGL(k_region_scan_b, dim3(1), dim3(c->threads), 0, c->stream, c->d_tile_a, c->d_tile_b, ntiles, c->d_tile_f, carry);
    hipLaunchKernelGGL(k_region_scan_c, dim3(ntiles), dim3(c->threads), 0, c->stream, c->d_isum, c->nregions, c->d_tile_f,
                       c->xnslots, img_slots, c->d_fin_i, c->d_counters, c->d_err, 2u);
  }
}

static int point_read(shk_ctx *c, PointOut *po) {
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_pinned, c->d_counters, (4 + SHK_HIST_BINS + 2) * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_pinned + 40, c->d_err, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  po->newd_after = c->h_pinned[0]; po->added_after = c->h_pinned[1]; po->removed = c->h_pinned[2]; po->added_before = c->h_pinned[3];
  po->err = *(uint32_t *)(c->h_pinned + 40);
  if (po->err) HIPCHK(hipMemsetAsync(c->d_err, 0, 16, c->stream));
  return SHK_OK;
}

// with_chist: the pass also records the first chunk of every key the table has not seen; c->h_chist[lo..hi] afterwards
static int point_try(shk_ctx *c, const uint64_t *words, uint32_t lo, uint32_t split, uint32_t hi, bool with_chist, PointOut *po) {
  int rc = point_alloc(c);
  if (rc) return rc;
  if (with_chist) { rc = ensure_chist(c); if (rc) return rc; }
  ShkMergeArgs A;
  fill_args(c, &A, words, lo, hi, 0, 0, 1, 0);
  A.split = split; A.isum = c->d_isum; A.ilens = c->d_ilens;
  if (with_chist) {
    A.want_hist = 2; A.newchunks = c->d_newchunks; A.chist = c->d_chist;
    HIPCHK(hipMemsetAsync(c->d_chist, 0, SHK_MAX_CHUNKS * 8, c->stream));
  }
  { const char *sp = getenv("SHK_STAMPS");
    A.dbg = (sp && strcmp(sp, "plain") != 0) ? (unsigned long long *)(c->d_scalars + 16) : nullptr; }
  c->spill_valid = 0;
  c->chist_n = 0;
  HIPCHK(hipMemsetAsync(c->d_counters, 0, (4 + SHK_HIST_BINS + 2) * 8, c->stream));
  { ProfScope ps(c, KP_MERGE_FUSED);
    SHK_FOR_REGION_SLICES(c, A, nblk)
      hipLaunchKernelGGL((k_region_merge<3, SHK_IMG_BLOCKS, true>), dim3(nblk), dim3(c->merge_group), 0, c->stream, A); }
  point_scans(c, true, true, 0);
  if (with_chist) {
    ProfScope ps(c, KP_MISC);
    hipLaunchKernelGGL(k_chunk_hist, dim3((c->nregions + SHK_CHIST_REGIONS - 1) / SHK_CHIST_REGIONS), dim3(256), 0, c->stream,
                       c->d_newchunks, c->d_summary, c->nregions, c->d_chist, 1u);
    HIPCHK(hipMemcpyAsync(c->h_chist, c->d_chist, ((uint64_t)hi + 1) * 8, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(hipMemcpyAsync(c->h_pinned + 48, c->d_fin_i + c->nregions, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_pinned + 49, c->d_ilens, 1, hipMemcpyDeviceToHost, c->stream));
  rc = point_read(c, po);
  if (rc) return rc;
  po->islots = c->h_pinned[4 + SHK_HIST_BINS + 1];
  po->ifin = c->h_pinned[48];
  po->first_used = (c->h_pinned[49] & 0xff) != 0;
  if (with_chist && !po->err) c->chist_n = hi + 1;
  return SHK_OK;
}

// the range walk of the round over the intermediate layout (k_denoise_marks_virtual). carry: what the shards in front of
// this one spill over the border (slots, >= 0); W/state: see ShkWalkShard. *nprot singletons, their quotients in c->d_prot.
static int point_walk(shk_ctx *c, long long carry, const ShkWalkShard &W, const uint64_t state_in[2], uint64_t state_out[2],
                      uint64_t *nprot, uint32_t *err) {
  const uint64_t ml = c->cfg.min_denoise_len ? c->cfg.min_denoise_len : (1ULL << 20);
  if (carry > 0) point_scans(c, false, true, carry);      // the layout as it is in the single table
  c->h_pinned[50] = state_in[0]; c->h_pinned[51] = state_in[1];
  HIPCHK(hipMemcpyAsync(c->d_scalars + 8, c->h_pinned + 50, 16, hipMemcpyHostToDevice, c->stream));
  { ProfScope ps(c, KP_MARKS);
    hipLaunchKernelGGL(k_denoise_marks_virtual, dim3(1), dim3(64), 0, c->stream, (const uint64_t *)c->d_fin_i, (const uint8_t *)c->d_ilens,
                       (const uint32_t *)c->d_isum, c->nslots, c->xnslots, ml, c->d_prot, SHK_PROT_CAP, (unsigned long long *)(c->d_scalars + 3),
                       W, (const uint64_t *)(c->d_scalars + 8), c->d_scalars + 10); }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_pinned + 47, c->d_scalars + 3, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_pinned + 52, c->d_scalars + 10, 16, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_pinned + 40, c->d_err, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  *nprot = c->h_pinned[47];
  state_out[0] = c->h_pinned[52]; state_out[1] = c->h_pinned[53];
  *err = *(uint32_t *)(c->h_pinned + 40);
  if (*err) HIPCHK(hipMemsetAsync(c->d_err, 0, 16, c->stream));
  return SHK_OK;
}

// the regions that hold a protected singleton, once more with the list; statistics of the whole pass again
static int point_finish(shk_ctx *c, const uint64_t *words, uint32_t lo, uint32_t split, uint32_t hi, uint64_t nprot, PointOut *po) {
  if (nprot > SHK_PROT_CAP) { po->err |= SHK_E_FUSED; return SHK_OK; }
  if (nprot) {
    std::vector<uint64_t> prot(nprot);
    HIPCHK(hipMemcpyAsync(prot.data(), c->d_prot, nprot * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<uint32_t> regs;
    for (uint64_t q : prot) { const uint32_t r = (uint32_t)(q >> SHK_REGION_LOG2); if (regs.empty() || regs.back() != r) regs.push_back(r); }
    ShkMergeArgs A;
    fill_args(c, &A, words, lo, hi, 0, 0, 1, 0);
    A.split = split; A.isum = c->d_isum; A.ilens = c->d_ilens;
    HIPCHK(hipMemsetAsync(c->d_counters, 0, (4 + SHK_HIST_BINS + 2) * 8, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_over_list, regs.data(), regs.size() * 4, hipMemcpyHostToDevice, c->stream));
    A.list = c->d_over_list; A.prot_list = c->d_prot; A.nprot = (uint32_t)nprot;
    { ProfScope ps(c, KP_MISC);
      hipLaunchKernelGGL((k_region_merge<3, SHK_IMG_BLOCKS, true>), dim3((uint32_t)regs.size()), dim3(c->merge_group), 0, c->stream, A); }
    HIPCHK(hipStreamSynchronize(c->stream));   // (regs lives on this stack frame)
    point_scans(c, true, false, 0);
    int rc = point_read(c, po);
    if (rc) return rc;
  }
  if (!po->err) {
    c->spill_valid = 1; c->spill_words = words; c->spill_lo = lo; c->spill_hi = hi; c->spill_denoise = 1; c->spill_big = c->big_image;
    c->spill_nover = 0;
  }
  return SHK_OK;
}

// verify: cstar is a GUESS (sample_locate). The pass then also records the first chunk of every key the table has not seen,
// like the plain pass does; if the exact histogram puts the point at cstar the pass stands, otherwise nothing is committed
// and *exact_ch / *crossing (1: the trigger is reached at chunk *exact_ch, 0: not reached in [lo, hi]) say what is true.
static int denoise_fused(shk_ctx *c, const uint64_t *words, uint32_t lo, uint32_t cstar, uint32_t hi, uint64_t newd_before,
                         shk_batch_stats *st, bool *done, bool verify = false, uint32_t *exact_ch = nullptr, int *crossing = nullptr) {
  *done = false;
  if (crossing) *crossing = -1;
  if (!c->use_spill || c->big_image || c->single_ok || getenv("SHK_NO_FUSED_POINT")) return SHK_OK;
  PointOut po;
  int rc = point_try(c, words, lo, cstar, hi, verify, &po);
  if (rc) return rc;
  if (po.err) {                                  // not this way: nothing was committed
    if (getenv("SHK_DEBUG_FUSED")) fprintf(stderr, "SHK_DEBUG_FUSED fallback: flags 0x%x\n", po.err);
    return SHK_OK;
  }
  if (verify) {
    // where the running distinct count really reaches the trigger (the loop of merge_stage_from)
    uint64_t run = c->ndistinct;
    uint32_t ch = lo;
    for (; ch < hi; ch++) {
      run += c->h_chist[ch];
      if (run >= c->cfg.ndistinct_for_denoise) break;
    }
    if (ch == hi) run += c->h_chist[ch];
    const bool crosses = run >= c->cfg.ndistinct_for_denoise;
    if (exact_ch) *exact_ch = ch;
    if (crossing) *crossing = crosses ? 1 : 0;
    if (getenv("SHK_DEBUG_FUSED")) fprintf(stderr, "SHK_DEBUG_FUSED guess %u exact %u crossing %d\n", cstar, ch, (int)crosses);
    if (!crosses || ch != cstar) return SHK_OK;
    newd_before = run - c->ndistinct;
  }
  ShkWalkShard W;
  W.prev_fp = -1; W.cap_local = c->nslots; W.last = 1; W.next_first_used = 0;
  const uint64_t s_in[2] = {0, 0};
  uint64_t s_out[2], nprot = 0;
  uint32_t werr = 0;
  rc = point_walk(c, 0, W, s_in, s_out, &nprot, &werr);
  if (rc) return rc;
  if (werr) return SHK_OK;
  rc = point_finish(c, words, lo, cstar, hi, nprot, &po);
  if (rc) return rc;
  if (po.err) {
    c->spill_valid = 0;
    if (getenv("SHK_DEBUG_FUSED")) fprintf(stderr, "SHK_DEBUG_FUSED fallback: flags 0x%x (second go)\n", po.err);
    return SHK_OK;
  }
  // would the trigger be reached again inside the rest? then the rounds have to be taken one by one
  if (c->rounds_left > 1 && c->ndistinct + newd_before - po.removed + po.newd_after >= c->cfg.ndistinct_for_denoise) { c->spill_valid = 0; return SHK_OK; }
  rc = merge_write(c, words, lo, hi, 1);
  if (rc) return rc;
  c->nelts = c->nelts + po.added_before - po.removed + po.added_after;      // inserts, CQF_mt.h:1037-1038, inserts
  c->ndistinct = c->ndistinct + newd_before - po.removed + po.newd_after;
  c->rounds_left--; c->rounds_done++;
  st->removed += po.removed; st->denoise_rounds++;
  st->kmers += po.added_before + po.added_after; st->new_distinct += newd_before + po.newd_after; st->chunks += hi - lo + 1;
  *done = true;
  if (getenv("SHK_DEBUG_FUSED")) fprintf(stderr, "SHK_DEBUG_FUSED one-pass point at chunk %u of [%u, %u]: removed %llu protected %llu\n", cstar, lo, hi,
                                         (unsigned long long)po.removed, (unsigned long long)nprot);
  return SHK_OK;
}

static int denoise_round(shk_ctx *c, uint64_t *removed) {
  int rc = denoise_round_once(c, removed);
  if (rc == SHK_ERR_REGION && !c->big_image && (c->last_err_bits & (SHK_E_OLD_EXTENT | SHK_E_NEW_EXTENT))) {
    c->big_image = 1;
    c->last_err_bits = 0;
    rc = denoise_round_once(c, removed);
  }
  if (!rc) c->big_image = 0;   // the round thinned the table out: back to the small image
  return rc;
}

// Where will the deNoise point of this batch fall? A statistics pass over every sample_stride-th region with the exact
// first-chunk record, scaled up: regions are hash buckets, so the sample's per-chunk counts of new keys are the whole
// table's divided by the stride, up to Poisson noise (variance of the scaled sum = stride x sum). The answer is only a
// guess -- the one-pass point that is run with it checks it against the full histogram it produces itself.
// sample_pass: c->h_chist[lo..hi] = the sample's histogram; *ns regions of c->nregions were looked at
static int sample_pass(shk_ctx *c, const uint64_t *words, uint32_t lo, uint32_t hi, uint32_t *ns_out, uint32_t *err_out) {
  int rc = ensure_chist(c);
  if (rc) return rc;
  const uint32_t stride = c->sample_stride > 1 ? c->sample_stride : 1;
  const uint32_t ns = (c->nregions + stride - 1) / stride;
  ShkMergeArgs A;
  fill_args(c, &A, words, lo, hi, lo, 0, 0, 2);
  A.newchunks = c->d_newchunks; A.chist = c->d_chist; A.rstride = stride;
  c->spill_valid = 0;
  c->chist_n = 0;
  HIPCHK(hipMemsetAsync(c->d_chist, 0, SHK_MAX_CHUNKS * 8, c->stream));
  { ProfScope ps(c, KP_MERGE_SAMPLE);
    for (uint32_t r0 = 0; r0 < ns; r0 += SHK_REGION_SLICE) {
      A.r0 = r0;
      const uint32_t nblk = ns - r0 < SHK_REGION_SLICE ? ns - r0 : SHK_REGION_SLICE;
      hipLaunchKernelGGL((k_region_merge<0, SHK_IMG_BLOCKS>), dim3(nblk), dim3(c->merge_group), 0, c->stream, A);
    }
    hipLaunchKernelGGL(k_chunk_hist, dim3((ns + SHK_CHIST_REGIONS - 1) / SHK_CHIST_REGIONS), dim3(256), 0, c->stream,
                       c->d_newchunks, c->d_summary, c->nregions, c->d_chist, stride); }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_chist, c->d_chist, ((uint64_t)hi + 1) * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_pinned + 40, c->d_err, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  *err_out = *(uint32_t *)(c->h_pinned + 40);
  if (*err_out) HIPCHK(hipMemsetAsync(c->d_err, 0, 16, c->stream));   // whatever it is, the full pass will meet it again and deal with it
  *ns_out = ns;
  return SHK_OK;
}

// verdict 0: no point expected in [lo, hi]; 1: expected at chunk *guess; 2: cannot tell
static int sample_locate(shk_ctx *c, const uint64_t *words, uint32_t lo, uint32_t hi, int *verdict, uint32_t *guess) {
  *verdict = 2;
  if (c->ndistinct >= c->cfg.ndistinct_for_denoise) return SHK_OK;
  uint32_t ns = 0, err = 0;
  int rc = sample_pass(c, words, lo, hi, &ns, &err);
  if (rc) return rc;
  if (err) return SHK_OK;
  const double F = (double)c->nregions / (double)ns;
  const double need = (double)(c->cfg.ndistinct_for_denoise - c->ndistinct);
  double cum = 0;
  uint32_t at = hi + 1;
  for (uint32_t ch = lo; ch <= hi; ch++) {
    cum += F * (double)c->h_chist[ch];
    if (at > hi && cum >= need) at = ch;
  }
  const double margin = 6.0 * sqrt(F * cum + 1.0);
  if (cum + margin < need) *verdict = 0;
  else if (cum - margin >= need && at <= hi) { *verdict = 1; *guess = at; }
  if (getenv("SHK_DEBUG_FUSED")) fprintf(stderr, "SHK_DEBUG_FUSED sample: need %.0f predicted %.0f +- %.0f -> verdict %d at %u of [%u, %u]\n", need, cum, margin / 6.0, *verdict, at, lo, hi);
  return SHK_OK;
}

// Insert the words of chunks [0, nchunks) (already partitioned in `words`), firing deNoise
// rounds where the t = 1 reference would: after the first chunk at which
// ndistinct >= trigger while rounds are left (CQF_mt.h:837, 860-869).
static int merge_stage_from(shk_ctx *c, const uint64_t *words, uint32_t nchunks, uint64_t nwords, shk_batch_stats *st,
                            uint32_t *lo_io) {
  // A summary over chunks that will turn out to lie behind a deNoise point is speculative:
  // "table full"/"extent" raised by its free-pointer scan mean nothing then.
  const uint32_t soft = SHK_E_TABLE_FULL | SHK_E_NEW_EXTENT;
  uint32_t &lo = *lo_io;
  while (lo < nchunks) {
    uint32_t hi = nchunks - 1;
    const bool watch = c->rounds_left > 0;
    MergeOut o;
    uint32_t shift;
    int rc;
    bool have_hist = false;
    {
      uint32_t span = hi - lo + 1;
      shift = 0;
      while ((span + (1u << shift) - 1) >> shift > SHK_HIST_BINS) shift++;
    }
    // a deNoise point inside this range is likely when the last range's rate of new keys would
    // carry the distinct count past the trigger: then the statistics-only launch runs first
    const bool likely = watch && c->new_frac > 0 &&
                        (double)c->ndistinct + 0.8 * c->new_frac * (double)nwords * (double)(hi - lo + 1) / (double)nchunks >=
                            (double)c->cfg.ndistinct_for_denoise;
    // When the trigger is within reach of this batch the first launch also fills the first-chunk
    // histogram, so that a pass which turns out to contain the deNoise point already yields its coarse position.
    // (with a known rate of new keys per k-mer, "within reach" means within twice the predicted gain; a
    // point that is missed this way only costs one more statistics pass)
    const double reach = c->new_frac > 0 ? 2.0 * c->new_frac * (double)nwords * (double)(hi - lo + 1) / (double)nchunks : (double)nwords;
    const bool possible = watch && (double)c->ndistinct + reach >= (double)c->cfg.ndistinct_for_denoise;
    int sv = 2;
    if (possible && c->sample_stride > 1 && c->use_spill && !c->single_ok && !c->big_image && !c->counted) {
      // guess the chunk of the deNoise point from a sample of the regions and run the one-pass point with it; the pass
      // verifies the guess against the exact histogram it collects itself and, when it was wrong, is run once more
      uint32_t guess = 0;
      rc = sample_locate(c, words, lo, hi, &sv, &guess);
      if (rc) return rc;
      if (sv == 1 && guess + 1 < nchunks) {
        bool fused = false;
        uint32_t ech = 0;
        int crossing = -1;
        rc = denoise_fused(c, words, lo, guess, nchunks - 1, 0, st, &fused, true, &ech, &crossing);
        if (rc) return rc;
        if (fused) { lo = nchunks; continue; }
        if (crossing == 1 && ech != guess && ech + 1 < nchunks) {
          uint64_t run = 0;
          for (uint32_t ch = lo; ch <= ech; ch++) run += c->h_chist[ch];
          rc = denoise_fused(c, words, lo, ech, nchunks - 1, run, st, &fused);
          if (rc) return rc;
          if (fused) { lo = nchunks; continue; }
        }
        // (anything else: the general path below)
      }
    }
    if (c->single_ok && !likely) {
      // single-launch scheme: one launch does statistics and table
      rc = merge_single(c, words, lo, hi, 0, &o, possible ? 1 : 0, lo, shift);
      have_hist = possible && !(o.err & ~soft);
      if (rc) return rc;
      if (o.err & ~(soft | SHK_E_HASH_FULL | SHK_E_LOOKBACK)) return map_err_bits(o.err & ~(soft | SHK_E_HASH_FULL | SHK_E_LOOKBACK));
      const bool crosses = watch && c->ndistinct + o.newd >= c->cfg.ndistinct_for_denoise;
      if (!o.err && !crosses) {
        commit_single(c);
        if (o.added) c->new_frac = (double)o.newd / (double)o.added;
        c->ndistinct += o.newd; c->nelts += o.added;
        st->kmers += o.added; st->new_distinct += o.newd; st->chunks += hi - lo + 1;
        lo = hi + 1;
        continue;
      }
      if (!crosses && (o.err & soft) && !(o.err & (SHK_E_HASH_FULL | SHK_E_LOOKBACK))) return map_err_bits(o.err);
      // otherwise (deNoise point inside, hash overflow, or look-back gave up): the two-launch path below
    }
    for (;;) {
      if (have_hist) break;   // statistics of [lo, hi] are already known from the discarded single launch
      uint32_t span = hi - lo + 1;
      shift = 0;
      while ((span + (1u << shift) - 1) >> shift > SHK_HIST_BINS) shift++;
      const int wh = sv == 0 ? 0 : ((likely || (possible && c->use_spill)) ? 2 : 0);   // (sv == 0: the sample rules a point out)
      rc = merge_summary(c, words, lo, hi, lo, shift, 0, &o, wh, 1);
      if (rc) return rc;
      if (wh && !(o.err & ~soft)) have_hist = true;
      if (o.err & ~(soft | SHK_E_HASH_FULL)) return map_err_bits(o.err & ~(soft | SHK_E_HASH_FULL));
      if (o.err & SHK_E_HASH_FULL) {
        // more distinct new keys in one region than its LDS hash holds: take fewer chunks at once
        if (hi == lo) return SHK_ERR_REGION;
        hi = lo + (hi - lo) / 2;
        continue;
      }
      break;
    }
    bool fire = false;
    if (watch && c->ndistinct + o.newd >= c->cfg.ndistinct_for_denoise) {
      // locate the first chunk at which the running distinct count reaches the trigger:
      // only now is the per-chunk histogram of first occurrences needed
      uint32_t base = lo;
      if (!have_hist) {
        rc = merge_summary(c, words, lo, hi, lo, shift, 0, &o, 2);
        if (rc) return rc;
        if (o.err & ~soft) return map_err_bits(o.err & ~soft);
      }
      if (o.have_chist) {
        // exact: first chunk at which the running distinct count reaches the trigger
        uint64_t run = c->ndistinct;
        uint32_t ch = lo;
        for (; ch < hi; ch++) {
          run += c->h_chist[ch];
          if (run >= c->cfg.ndistinct_for_denoise) break;
        }
        if (ch == hi) run += c->h_chist[ch];      // (the loop leaves the last chunk's keys out)
        if (ch + 1 < nchunks) {
          // the point lies inside the batch: everything -- the chunks up to it, the round, the chunks behind it -- in one pass
          bool fused = false;
          rc = denoise_fused(c, words, lo, ch, nchunks - 1, run - c->ndistinct, st, &fused);
          if (rc) return rc;
          if (fused) { lo = nchunks; continue; }
        }
        hi = ch;
      } else
      for (;;) {
        uint32_t bin = 0;
        uint64_t run = c->ndistinct + o.before;
        for (bin = 0; bin < SHK_HIST_BINS; bin++) {
          if (run + o.hist[bin] >= c->cfg.ndistinct_for_denoise) break;
          run += o.hist[bin];
        }
        if (bin == SHK_HIST_BINS) bin = SHK_HIST_BINS - 1;  // cannot happen: the total crosses
        uint32_t b_lo = base + (bin << shift);
        uint32_t b_hi = b_lo + (1u << shift) - 1;
        if (b_hi > hi) b_hi = hi;
        if (shift == 0) { hi = b_lo; break; }
        // refine inside [b_lo, b_hi]: keys first seen before b_lo are counted in `before`
        uint32_t span2 = b_hi - b_lo + 1;
        shift = 0;
        while ((span2 + (1u << shift) - 1) >> shift > SHK_HIST_BINS) shift++;
        base = b_lo;
        rc = merge_summary(c, words, lo, b_hi, base, shift, 0, &o, 1);
        if (rc) return rc;
        if (o.err & ~soft) return map_err_bits(o.err & ~soft);
      }
      fire = true;
      // rebuild for exactly the chunks [lo, hi]
      bool written = false;
      if (c->single_ok) {
        rc = merge_single(c, words, lo, hi, 0, &o);
        if (rc) return rc;
        if (!o.err) { commit_single(c); written = true; }
        else if (!(o.err & SHK_E_LOOKBACK)) return map_err_bits(o.err);
      }
      if (!written) {
        rc = merge_summary(c, words, lo, hi, lo, 0, 0, &o, 0, 1);
        if (rc) return rc;
        if (o.err) return map_err_bits(o.err);
        rc = merge_write(c, words, lo, hi, 0);
        if (rc) return rc;
      }
    } else {
      if (o.err) return map_err_bits(o.err);
      if (have_hist) {
        // (only reached when the single launch was clean but is not committed: cannot happen without a crossing)
      }
      rc = merge_write(c, words, lo, hi, 0);
      if (rc) return rc;
    }
    if (o.added && !fire) c->new_frac = (double)o.newd / (double)o.added;
    c->ndistinct += o.newd;
    c->nelts += o.added;
    st->kmers += o.added;
    st->new_distinct += o.newd;
    st->chunks += hi - lo + 1;
    if (fire) {
      uint64_t removed = 0;
      c->rounds_left--;
      c->rounds_done++;
      if (hi + 1 < nchunks) {
        bool done = false;
        rc = denoise_with_rest(c, words, hi + 1, nchunks - 1, st, &done);
        if (rc) return rc;
        if (done) { lo = nchunks; continue; }
      }
      rc = denoise_round(c, &removed);
      if (rc) return rc;
      st->removed += removed;
      st->denoise_rounds++;
    }
    lo = hi + 1;
  }
  return SHK_OK;
}

// A cluster longer than the small LDS image makes a pass fail with an extent flag before anything is
// committed: the remaining chunks are then rebuilt with the big image (until the next deNoise round
// thins the table out again).
static int merge_stage(shk_ctx *c, const uint64_t *words, uint32_t nchunks, uint64_t nwords, shk_batch_stats *st) {
  uint32_t lo = 0;
  int rc = merge_stage_from(c, words, nchunks, nwords, st, &lo);
  if (rc == SHK_ERR_REGION && !c->big_image && (c->last_err_bits & (SHK_E_OLD_EXTENT | SHK_E_NEW_EXTENT))) {
    c->big_image = 1;
    c->last_err_bits = 0;
    rc = merge_stage_from(c, words, nchunks, nwords, st, &lo);
  }
  return rc;
}

static int finish(shk_ctx *c, int rc) {
  uint32_t bits = 0;
  int rc2 = fetch_err(c, &bits);
  prof_collect(c);
  if (rc) return rc;
  if (rc2) return rc2;
  return map_err_bits(bits);
}

extern "C" int shk_count_chunks(shk_ctx *c, const void *text, int text_on_device, uint64_t text_bytes,
                                const uint64_t *chunk_off, const uint64_t *chunk_len, uint32_t nchunks,
                                shk_batch_stats *stats) {
  if (!c || !text || !chunk_off || !chunk_len) return SHK_ERR_ARG;
  shk_batch_stats st;
  memset(&st, 0, sizeof(st));
  HIPCHK(hipSetDevice(c->dev));
  const bool roll = roll_path(c);
  bool h1 = false;
  int rc = roll ? roll_stage(c, text, text_on_device, text_bytes, chunk_off, chunk_len, nchunks, 0, 1, &h1)
                : hash_stage(c, text, text_on_device, text_bytes, chunk_off, chunk_len, nchunks, 0, 1, true);
  if (rc) return finish(c, rc);
  uint32_t bits = 0;
  HIPCHK(hipMemcpyAsync(c->h_pinned + 42, c->d_scalars + 1, 8, hipMemcpyDeviceToHost, c->stream));
  if (fetch_err(c, &bits)) return SHK_ERR_HIP;
  if (bits) { prof_collect(c); return map_err_bits(bits); }
  const uint64_t nwords = c->h_pinned[42];
  if (nwords > c->cfg.max_batch_keys) { prof_collect(c); return SHK_ERR_BATCH; }
  int dst = 0;
  // (the roll kernels leave the words partitioned by the first digit; the hash kernel has counted the first level's digits)
  rc = roll ? partition_stage(c, 0, nwords, &dst, nullptr, false, 1, h1) : partition_stage(c, 0, nwords, &dst, nullptr, true);
  if (rc) return finish(c, rc);
  rc = merge_stage(c, c->d_words[dst], nchunks, nwords, &st);
  if (stats) *stats = st;
  return finish(c, rc);
}

// ------------------------------------------------------------------ overlapped front end
// shk_count_chunks = front end (parse, hash, partition: bound by HBM and instruction issue in turn) + rebuild (bound by
// the CUs' LDS pipelines and instruction issue), one after the other on one stream. The two halves of DIFFERENT batches
// have nothing to do with each other until the rebuild reads the partitioned words, so the front end of batch s+1 can
// run on a second stream while batch s is rebuilt: shk_prepare_chunks starts it and returns, shk_count_prepared takes
// the oldest prepared batch through the rebuild. Two batches may be prepared ahead. The front end works in a shadow of the
// context: its own stream, scalars, error word, scan scratch and partition buffers; two slots of (partitioned words,
// region bases) alternate between "being prepared" and "being rebuilt". The shadow runs the unchanged stage functions
// (host synchronisations included) in a helper thread, so the caller's thread is free to drive the rebuild.
struct ShkFrontSlot {
  uint64_t *words = nullptr;    // the buffer the last partition level writes into (and the roll kernels, two levels earlier)
  uint64_t *base = nullptr;     // region bases of that batch (region ENDS when cap != 0)
  uint32_t cap = 0;             // shk_ctx::region_cap of that batch
  std::thread th;
  bool busy = false;
  int rc = 0, dst = 0;
  uint64_t nwords = 0;
  uint32_t nchunks = 0;
};
struct ShkFront {
  shk_ctx *f = nullptr;
  ShkFrontSlot slot[2];
  int head = 0, count = 0;      // oldest prepared slot, prepared slots
  int par = 0;                  // index of d_words[] the last level writes into
  uint64_t *scratch = nullptr;  // the other d_words[] of the shadow
};

static int front_init(shk_ctx *c) {
  ShkFront *F = new ShkFront();
  shk_ctx *f = new shk_ctx(*c);
  F->f = f;
  f->front = nullptr;
  f->pending.clear(); f->evpool.clear();
  f->copy_stream = hipStream_t(); f->d_up[0] = f->d_up[1] = nullptr; f->up_pending[0] = f->up_pending[1] = 0;
  for (int i = 0; i < KP_N; i++) { f->prof_ms[i] = 0; f->prof_n[i] = 0; }
  c->front = F;
  const uint32_t maxch = SHK_MAX_CHUNKS;
  const uint64_t capk = c->cfg.max_batch_keys;
  // everything the front end writes is the shadow's own (null first: a failed allocation leaves nothing dangling)
  f->d_text = nullptr; f->d_chunk_off = f->d_chunk_len = f->d_nlines = f->d_reads_base = f->d_rd_start = f->d_rd_end = nullptr;
  f->d_rd_chunk = nullptr; f->d_nkeys = nullptr; f->d_key_base = nullptr; f->d_scalars = nullptr; f->d_block_sums = nullptr;
  f->d_base_sub = nullptr; f->d_cursor = nullptr; f->d_tfb = nullptr; f->d_err = nullptr; f->h_pinned = nullptr;
  for (int l = 0; l < 4; l++) f->d_hist[l] = nullptr;
  for (int l = 0; l < 5; l++) f->d_base[l] = nullptr;
  f->d_words[0] = f->d_words[1] = nullptr;
  // (a higher stream priority changes nothing measurable: the rebuild's small workgroups refill every CU as fast as they
  // leave it, whatever the priority of the queue whose big workgroups are waiting)
  HIPCHK(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
  if (dmalloc(&f->d_text, c->cfg.max_batch_bytes + 64)) return SHK_ERR_HIP;
  if (dmalloc(&f->d_chunk_off, maxch) || dmalloc(&f->d_chunk_len, maxch) || dmalloc(&f->d_nlines, (uint64_t)maxch * SHK_PARSE_SEGS) ||
      dmalloc(&f->d_reads_base, maxch + 1)) return SHK_ERR_HIP;
  if (dmalloc(&f->d_rd_start, c->max_reads + 1) || dmalloc(&f->d_rd_end, c->max_reads + 1) || dmalloc(&f->d_nkeys, c->max_reads + 1) ||
      dmalloc(&f->d_rd_chunk, c->max_reads + 1) || dmalloc(&f->d_key_base, c->max_reads + 2)) return SHK_ERR_HIP;
  if (dmalloc(&f->d_scalars, 64)) return SHK_ERR_HIP;
  HIPCHK(hipMemsetAsync(f->d_scalars, 0, 64 * 8, f->stream));
  { uint64_t mx = capk > c->max_reads ? capk : c->max_reads;
    uint64_t pw = 1ULL << c->rbits;
    if (pw > mx) mx = pw;
    if (dmalloc(&f->d_block_sums, mx / SHK_SCAN_TILE + 4 + 8192)) return SHK_ERR_HIP; }
  { uint64_t nb = 1;
    if (dmalloc(&f->d_base[0], 2)) return SHK_ERR_HIP;
    for (uint32_t l = 0; l < c->nlevels; l++) {
      const uint64_t n = nb << c->lv[l].bits;
      if (dmalloc(&f->d_hist[l], (n << c->lv[l].ng_log2) + 1)) return SHK_ERR_HIP;
      if (l + 1 < c->nlevels && dmalloc(&f->d_base[l + 1], n + 2)) return SHK_ERR_HIP;
      if (l + 1 == c->nlevels) for (int k2 = 0; k2 < 2; k2++) if (dmalloc(&F->slot[k2].base, n + 2)) return SHK_ERR_HIP;
      if (l == 0 && dmalloc(&f->d_base_sub, (n << c->lv[0].ng_log2) + 2)) return SHK_ERR_HIP;
      nb = n;
    }
    { const uint64_t first = (1ULL << (c->lv[0].bits + c->lv[0].ng_log2));
      if (dmalloc(&f->d_cursor, (nb > first ? nb : first) + 2)) return SHK_ERR_HIP; } }
  if (dmalloc(&f->d_tfb, capk / SHK_RP_TILE + 2)) return SHK_ERR_HIP;
  if (dmalloc(&f->d_err, 4)) return SHK_ERR_HIP;
  HIPCHK(hipMemsetAsync(f->d_err, 0, 16, f->stream));
  HIPCHK(hipHostMalloc((void **)&f->h_pinned, 64 * sizeof(uint64_t), hipHostMallocDefault));
  // the roll kernels write d_words[0]; every further level flips: the last one lands in d_words[(nlevels - 1) & 1]
  F->par = roll_path(c) ? (int)((c->nlevels - 1) & 1) : (int)(c->nlevels & 1);
  if (dmalloc(&F->scratch, capk + 1) || dmalloc(&F->slot[0].words, capk + 1) || dmalloc(&F->slot[1].words, capk + 1)) return SHK_ERR_HIP;
  HIPCHK(hipStreamSynchronize(f->stream));
  return SHK_OK;
}
static void front_destroy(shk_ctx *c) {
  ShkFront *F = c->front;
  if (!F) return;
  for (int k2 = 0; k2 < 2; k2++) if (F->slot[k2].th.joinable()) F->slot[k2].th.join();
  shk_ctx *f = F->f;
  if (f) {
    if (f->stream) hipStreamSynchronize(f->stream);
    prof_collect(f);
    for (size_t i = 0; i < f->evpool.size(); i++) hipEventDestroy(f->evpool[i]);
    hipFree(f->d_text); hipFree(f->d_chunk_off); hipFree(f->d_chunk_len); hipFree(f->d_nlines); hipFree(f->d_reads_base);
    hipFree(f->d_rd_start); hipFree(f->d_rd_end); hipFree(f->d_rd_chunk); hipFree(f->d_nkeys); hipFree(f->d_key_base); hipFree(f->d_scalars);
    hipFree(f->d_block_sums); hipFree(f->d_base[0]); hipFree(f->d_base_sub);
    for (uint32_t l = 0; l < c->nlevels; l++) { hipFree(f->d_hist[l]); if (l + 1 < c->nlevels) hipFree(f->d_base[l + 1]); }
    hipFree(f->d_cursor); hipFree(f->d_tfb); hipFree(f->d_err);
    if (f->h_pinned) hipHostFree(f->h_pinned);
    if (f->stream) hipStreamDestroy(f->stream);
    delete f;
  }
  hipFree(F->scratch);
  for (int k2 = 0; k2 < 2; k2++) { hipFree(F->slot[k2].words); hipFree(F->slot[k2].base); }
  delete F;
  c->front = nullptr;
}

// the front end of one batch in the shadow context; runs in the helper thread
static void front_run(shk_ctx *c, ShkFrontSlot *S, const void *text, int on_device, uint64_t text_bytes, std::vector<uint64_t> off,
                      std::vector<uint64_t> len) {
  ShkFront *F = c->front;
  shk_ctx *f = F->f;
  hipSetDevice(c->dev);
  f->prof_on = c->prof_on;
  f->d_words[F->par] = S->words; f->d_words[F->par ^ 1] = F->scratch;
  f->d_base[c->nlevels] = S->base;
  const uint32_t nchunks = (uint32_t)off.size();
  const bool roll = roll_path(c);
  S->nwords = 0; S->nchunks = nchunks;
  bool h1 = false;
  int rc = roll ? roll_stage(f, text, on_device, text_bytes, off.data(), len.data(), nchunks, 0, 1, &h1)
                : hash_stage(f, text, on_device, text_bytes, off.data(), len.data(), nchunks, 0, 1, true);
  uint32_t bits = 0;
  if (!rc) {
    if (hipMemcpyAsync(f->h_pinned + 42, f->d_scalars + 1, 8, hipMemcpyDeviceToHost, f->stream) != hipSuccess || fetch_err(f, &bits)) rc = SHK_ERR_HIP;
    else if (bits) rc = map_err_bits(bits);
    else if (f->h_pinned[42] > c->cfg.max_batch_keys) rc = SHK_ERR_BATCH;
  }
  if (!rc) {
    S->nwords = f->h_pinned[42];
    int dst = 0;
    rc = roll ? partition_stage(f, 0, S->nwords, &dst, nullptr, false, 1, h1) : partition_stage(f, 0, S->nwords, &dst, nullptr, true);
    S->dst = dst; S->cap = f->region_cap;
    if (!rc) {
      if (fetch_err(f, &bits)) rc = SHK_ERR_HIP;          // (synchronises the shadow's stream: the batch is ready)
      else if (bits) rc = map_err_bits(bits);
      else if (dst != F->par) rc = SHK_ERR_CORRUPT;       // (the slot's buffer must be the one the last level wrote)
    }
  }
  if (rc) hipStreamSynchronize(f->stream);
  S->rc = rc;
}

extern "C" int shk_prepare_chunks(shk_ctx *c, const void *text, int text_on_device, uint64_t text_bytes, const uint64_t *chunk_off,
                                  const uint64_t *chunk_len, uint32_t nchunks) {
  if (!c || !text || !chunk_off || !chunk_len || nchunks == 0 || nchunks > SHK_MAX_CHUNKS) return SHK_ERR_ARG;
  if (c->cfg.num_shards > 1) return SHK_ERR_ARG;          // (a shard's words go through the exchange: shk_hash_chunks)
  HIPCHK(hipSetDevice(c->dev));
  if (!c->front) { int rc = front_init(c); if (rc) { front_destroy(c); return rc; } }
  ShkFront *F = c->front;
  if (F->count == 2) return SHK_ERR_BATCH;                // two batches are prepared already: count one first
  ShkFrontSlot *S = &F->slot[(F->head + F->count) & 1];
  // one front end at a time: the previous one (the other slot's) must have left the shadow's buffers
  ShkFrontSlot *O = &F->slot[(F->head + F->count + 1) & 1];
  if (O->th.joinable()) O->th.join();
  if (text_on_device)
    for (int b = 0; b < 2; b++)   // a buffer of shk_upload_text whose copy may still be running
      if (c->d_up[b] && text == (const void *)c->d_up[b] && c->up_pending[b]) { HIPCHK(hipStreamWaitEvent(F->f->stream, c->up_done[b], 0)); c->up_pending[b] = 0; }
  std::vector<uint64_t> off(chunk_off, chunk_off + nchunks), len(chunk_len, chunk_len + nchunks);
  S->busy = true;
  F->count++;
#if defined(__HIPCC__)
  S->th = std::thread(front_run, c, S, text, text_on_device, text_bytes, std::move(off), std::move(len));
#else     // (the CPU emulator build of the tests keeps its kernels on the calling thread)
  front_run(c, S, text, text_on_device, text_bytes, std::move(off), std::move(len));
#endif
  return SHK_OK;
}

extern "C" int shk_prepare_reserve(shk_ctx *c) {
  if (!c || c->cfg.num_shards > 1) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  if (!c->front) { int rc = front_init(c); if (rc) { front_destroy(c); return rc; } }
  if (c->cfg.num_denoise) {      // (the records of a deNoise point, otherwise allocated by the first pass that needs them)
    int rc = ensure_chist(c);
    if (!rc) rc = point_alloc(c);
    if (rc) return rc;
  }
  return SHK_OK;
}

extern "C" int shk_count_prepared(shk_ctx *c, shk_batch_stats *stats) {
  if (!c || !c->front || c->front->count == 0) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  ShkFront *F = c->front;
  ShkFrontSlot *S = &F->slot[F->head];
  if (S->th.joinable()) S->th.join();
  F->head ^= 1; F->count--;
  S->busy = false;
  shk_ctx *f = F->f;
  shk_batch_stats st;
  memset(&st, 0, sizeof(st));
  if (stats) *stats = st;
  // the front end's kernel times join the context's (its events were recorded on the shadow's stream)
  { // (the other slot's front end may be running: it only appends to f->pending from its own thread, so collect
    // what THIS batch left only when nobody else is inside the shadow)
    ShkFrontSlot *O = &F->slot[F->head];
    if (!(O->busy && O->th.joinable())) {
      prof_collect(f);
      for (int i = 0; i < KP_N; i++) { c->prof_ms[i] += f->prof_ms[i]; c->prof_n[i] += f->prof_n[i]; f->prof_ms[i] = 0; f->prof_n[i] = 0; }
    } }
  if (S->rc) return S->rc;
  uint64_t *saved = c->d_base[c->nlevels];
  const uint32_t saved_cap = c->region_cap;
  c->d_base[c->nlevels] = S->base; c->region_cap = S->cap;
  int rc = merge_stage(c, S->words, S->nchunks, S->nwords, &st);
  c->d_base[c->nlevels] = saved; c->region_cap = saved_cap;
  if (stats) *stats = st;
  return finish(c, rc);
}

// Start copying host text for a later call into one of two context-owned device buffers (they alternate). The copy
// runs on its own stream, next to whatever the context is computing; the call that is handed the returned pointer
// (text_on_device = 1) waits for it. A buffer is reused by the second-next upload, i.e. after the call that read it.
extern "C" int shk_upload_text(shk_ctx *c, const void *host_text, uint64_t nbytes, void **d_text) {
  if (!c || !host_text || !d_text || nbytes > c->cfg.max_batch_bytes) return c && host_text && d_text ? SHK_ERR_BATCH : SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  if (!c->copy_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; b++) HIPCHK(hipEventCreateWithFlags(&c->up_done[b], hipEventDisableTiming));
  }
  const int b = c->up_next;
  if (!c->d_up[b] && dmalloc(&c->d_up[b], c->cfg.max_batch_bytes + 64)) return SHK_ERR_HIP;
  HIPCHK(hipMemcpyAsync(c->d_up[b], host_text, nbytes, hipMemcpyHostToDevice, c->copy_stream));
  HIPCHK(hipEventRecord(c->up_done[b], c->copy_stream));
  c->up_pending[b] = 1;
  c->up_next ^= 1;
  *d_text = c->d_up[b];
  return SHK_OK;
}

// page-locked host memory for shk_upload_text sources (callers that do not link the HIP runtime themselves)
extern "C" int shk_host_alloc(uint64_t nbytes, void **p) {
  if (!p) return SHK_ERR_ARG;
  HIPCHK(hipHostMalloc(p, nbytes, hipHostMallocDefault));
  return SHK_OK;
}
extern "C" void shk_host_free(void *p) { if (p) hipHostFree(p); }

extern "C" int shk_hash_chunks(shk_ctx *c, const void *text, int text_on_device, uint64_t text_bytes,
                               const uint64_t *chunk_off, const uint64_t *chunk_len, uint32_t nchunks,
                               uint64_t **d_words, uint64_t *nwords) {
  if (!c || !text || !d_words || !nwords) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  int rc = hash_stage(c, text, text_on_device, text_bytes, chunk_off, chunk_len, nchunks, c->cfg.shard_index, c->cfg.num_shards ? c->cfg.num_shards : 1);
  if (rc) return finish(c, rc);
  HIPCHK(hipMemcpyAsync(c->h_pinned + 42, c->d_scalars + 1, 8, hipMemcpyDeviceToHost, c->stream));
  rc = finish(c, 0);
  *d_words = c->d_words[0];
  *nwords = c->h_pinned[42];
  return rc;
}

extern "C" int shk_count_words(shk_ctx *c, const uint64_t *d_words, uint64_t nwords, uint32_t nchunks,
                               shk_batch_stats *stats) {
  if (!c || (!d_words && nwords) || nchunks == 0 || nchunks > SHK_MAX_CHUNKS) return SHK_ERR_ARG;
  if (nwords > c->cfg.max_batch_keys) return SHK_ERR_BATCH;
  shk_batch_stats st;
  memset(&st, 0, sizeof(st));
  HIPCHK(hipSetDevice(c->dev));
  c->h_pinned[43] = nwords;
  HIPCHK(hipMemcpyAsync(c->d_scalars + 1, c->h_pinned + 43, 8, hipMemcpyHostToDevice, c->stream));
  int dst = 0;
  // the first partition level reads the caller's buffer in place (no staging copy)
  int rc = d_words == c->d_words[0] ? partition_stage(c, 0, nwords, &dst)
         : d_words == c->d_words[1] ? partition_stage(c, 1, nwords, &dst)
                                    : partition_stage(c, 0, nwords, &dst, d_words);
  if (rc) return finish(c, rc);
  rc = merge_stage(c, c->d_words[dst], nchunks, nwords, &st);
  if (stats) *stats = st;
  return finish(c, rc);
}

extern "C" int shk_route_words(shk_ctx *c, uint64_t nwords, uint32_t nshards, uint64_t **d_out, uint64_t *counts) {
  if (!c || !d_out || !counts || nshards == 0 || (nshards & (nshards - 1)) || nshards > SHK_RP_MAXP) return SHK_ERR_ARG;
  if (nwords > c->cfg.max_batch_keys) return SHK_ERR_BATCH;
  HIPCHK(hipSetDevice(c->dev));
  uint32_t lg = 0;
  while ((1u << lg) < nshards) lg++;
  uint64_t *send = c->d_words[1];
  if (!getenv("SHK_ROUTE_SINGLE_BUFFER")) {
    const int b = c->send_next;
    if (!c->d_send[b] && dmalloc(&c->d_send[b], c->cfg.max_batch_keys + 1)) return SHK_ERR_HIP;
    send = c->d_send[b];
    c->send_next ^= 1;
  }
  if (lg == 0) {
    HIPCHK(hipMemcpyAsync(send, c->d_words[0], nwords * 8, hipMemcpyDeviceToDevice, c->stream));
    *d_out = send; counts[0] = nwords;
    return finish(c, 0);
  }
  if (c->cfg.qb < SHK_REGION_LOG2 + lg) return SHK_ERR_ARG;
  // one partition level over the WHOLE filter's regions: digit = owner
  ShkRpLevel lv;
  lv.shift = (c->cfg.qb - SHK_REGION_LOG2) - lg; lv.bits = lg; lv.nbuckets = 1; lv.hb = c->cfg.hb; lv.q_lo = 0;
  lv.nslots = ~0ULL; lv.out32 = 0; lv.ablate = 0; lv.ng_log2 = 0; lv.slot_cap = 0;
  c->h_pinned[43] = nwords;
  HIPCHK(hipMemcpyAsync(c->d_scalars + 1, c->h_pinned + 43, 8, hipMemcpyHostToDevice, c->stream));
  const uint64_t *n_p = c->d_scalars + 1;
  const uint32_t nwin = (uint32_t)(nwords / SHK_RP_TILE + 1);
  uint64_t *hist = c->d_block_sums;            // scratch: nshards <= 1024 words each
  uint64_t *base = c->d_block_sums + 2048;
  uint64_t *cursor = c->d_block_sums + 4096;
  { ProfScope ps(c, KP_RP_PREP);
    hipLaunchKernelGGL(k_rp_base1, dim3(1), dim3(64), 0, c->stream, n_p, c->d_base[0]);
    hipLaunchKernelGGL(k_rp_tile_first, dim3(nwin / 256 + 1), dim3(256), 0, c->stream, c->d_base[0], 1u, n_p, c->d_tfb);
    HIPCHK(hipMemsetAsync(hist, 0, nshards * 8, c->stream)); }
  { ProfScope ps(c, KP_RP_HIST);
    const uint32_t wt = nwin / 4096 + 1;
    hipLaunchKernelGGL(k_rp_hist, dim3(nwin / wt + 1), dim3(c->threads), 0, c->stream, c->d_words[0], n_p, c->d_base[0], c->d_tfb, lv, hist, wt); }
  HIPCHK(hipMemcpyAsync(c->h_pinned + 16, hist, nshards * 8 > 16 * 8 ? 16 * 8 : nshards * 8, hipMemcpyDeviceToHost, c->stream));
  std::vector<uint64_t> hh(nshards);
  HIPCHK(hipMemcpyAsync(hh.data(), hist, nshards * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  std::vector<uint64_t> bb(nshards + 1, 0);
  for (uint32_t i = 0; i < nshards; i++) { counts[i] = hh[i]; bb[i + 1] = bb[i] + hh[i]; }
  HIPCHK(hipMemcpyAsync(cursor, bb.data(), nshards * 8, hipMemcpyHostToDevice, c->stream));
  (void)base;
  { ProfScope ps(c, KP_RP_SCATTER);
    hipLaunchKernelGGL((k_rp_scatter<12, SHK_RP_THREADS>), dim3(nwin), dim3(SHK_RP_THREADS), 0, c->stream, c->d_words[0], send, n_p,
                       c->d_base[0], c->d_tfb, lv, cursor, c->d_err); }
  HIPCHK(hipGetLastError());
  *d_out = send;
  return finish(c, 0);
}

// shk_hash_chunks + shk_route_words in one: the roll kernels (roll_kernels.hip) hash every k-mer and send it straight
// to its OWNER's bin of the send buffer (digit = the top log2(nshards) bits of the whole filter's regions) -- no key word
// is written to HBM and read back before the exchange. Chunk i is labelled i * num_shards + shard_index as in
// shk_hash_chunks.
extern "C" int shk_hash_route_chunks(shk_ctx *c, const void *text, int text_on_device, uint64_t text_bytes, const uint64_t *chunk_off,
                                     const uint64_t *chunk_len, uint32_t nchunks, uint32_t nshards, uint64_t **d_out, uint64_t *counts,
                                     uint64_t *nwords) {
  if (!c || !text || !chunk_off || !chunk_len || !d_out || !counts || !nwords) return SHK_ERR_ARG;
  if (nshards == 0 || (nshards & (nshards - 1)) || nshards > SHK_RP_MAXP) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  uint32_t lg = 0;
  while ((1u << lg) < nshards) lg++;
  if (c->cfg.qb < SHK_REGION_LOG2 + lg) return SHK_ERR_ARG;
  const uint32_t chunk_first = c->cfg.shard_index, chunk_mul = c->cfg.num_shards ? c->cfg.num_shards : 1;
  if (nchunks == 0 || nchunks > SHK_MAX_CHUNKS || chunk_first + (uint64_t)(nchunks - 1) * chunk_mul >= SHK_MAX_CHUNKS) return SHK_ERR_BATCH;
  uint64_t *send = c->d_words[1];
  if (!getenv("SHK_ROUTE_SINGLE_BUFFER")) {
    const int b = c->send_next;
    if (!c->d_send[b] && dmalloc(&c->d_send[b], c->cfg.max_batch_keys + 1)) return SHK_ERR_HIP;
    send = c->d_send[b];
    c->send_next ^= 1;
  }
  const uint8_t *dtext;
  uint64_t nreads;
  int rc = parse_stage(c, text, text_on_device, text_bytes, chunk_off, chunk_len, nchunks, &dtext, &nreads);
  if (rc) return finish(c, rc);
  // The words are binned by MORE bits than the owner's (7, when the filter has them): with a handful of bins every
  // lane's LDS atomic lands on the same few counters (one rank: 64-way serialised, the kernels took 3.3 and 8.3 ms
  // instead of 1.3 and 4.7). An owner's bin is then 2^(bits - lg) consecutive sub-bins, contiguous in the send buffer.
  const uint32_t rb = c->cfg.qb - SHK_REGION_LOG2;
  const uint32_t db = lg > 7 ? lg : (rb < 7 ? (rb > lg ? rb : lg) : 7);
  const uint32_t nbins = 1u << db, per_owner = nbins / nshards;
  uint64_t *hist = c->d_block_sums;            // scratch: nbins <= 1024 words each
  uint64_t *cursor = c->d_block_sums + 4096;
  ShkRollArgs A;
  A.text = dtext; A.safe_end = (text_bytes + 15) & ~15ULL;
  A.rd_start = c->d_rd_start; A.rd_end = c->d_rd_end; A.nreads_p = c->d_scalars + 0; A.rd_chunk = c->d_rd_chunk;
  A.chunk_first = chunk_first; A.chunk_mul = chunk_mul; A.k = c->cfg.k; A.hb = c->cfg.hb;
  A.q_lo = 0;                                   // (owners are ranges of the WHOLE filter's quotients)
  A.dig_shift = rb - db; A.dig_bits = db;
  A.hist_shift = A.dig_shift; A.hist_bits = db;
  A.hist = hist; A.cursor = cursor; A.out = send; A.cap = c->cfg.max_batch_keys; A.err = c->d_err;
  rc = pack_stage(c, A, nreads, text_bytes, c->d_words[0]);      // (d_words[0]: filled by shk_stage_words, after this call)
  if (rc) return finish(c, rc);
  HIPCHK(hipMemsetAsync(hist, 0, nbins * 8, c->stream));     // (behind pack_stage, whose scan uses the head of d_block_sums too)
  { ProfScope ps(c, KP_ROLL_HIST);
    if (c->threads >= 512) {
      const uint64_t blocks = nreads / 256 + 1;
      hipLaunchKernelGGL((k_roll_hist<10, 256>), dim3((uint32_t)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, c->stream, A);
    } else {
      const uint64_t blocks = nreads / 64 + 1;
      hipLaunchKernelGGL((k_roll_hist<10, 64>), dim3((uint32_t)(blocks < 64 ? blocks : 64)), dim3(64), 0, c->stream, A);
    } }
  std::vector<uint64_t> hh(nbins);
  HIPCHK(hipMemcpyAsync(hh.data(), hist, nbins * 8, hipMemcpyDeviceToHost, c->stream));
  uint32_t bits = 0;
  if (fetch_err(c, &bits)) return SHK_ERR_HIP;
  if (bits) { prof_collect(c); return map_err_bits(bits); }
  std::vector<uint64_t> bb(nbins + 1, 0);
  for (uint32_t i = 0; i < nshards; i++) counts[i] = 0;
  for (uint32_t i = 0; i < nbins; i++) { counts[i / per_owner] += hh[i]; bb[i + 1] = bb[i] + hh[i]; }
  *nwords = bb[nbins];
  if (bb[nbins] > c->cfg.max_batch_keys) { prof_collect(c); return SHK_ERR_BATCH; }
  HIPCHK(hipMemcpyAsync(cursor, bb.data(), nbins * 8, hipMemcpyHostToDevice, c->stream));
  { ProfScope ps(c, KP_ROLL_SCATTER);
    if (c->threads >= 512) {
      const uint64_t blocks = nreads / 1024 + 1;
      hipLaunchKernelGGL((k_roll_scatter<1024, 1>), dim3((uint32_t)(blocks < 512 ? blocks : 512)), dim3(1024), 0, c->stream, A);
    } else {
      const uint64_t blocks = nreads / 64 + 1;
      hipLaunchKernelGGL((k_roll_scatter<64, 4>), dim3((uint32_t)(blocks < 64 ? blocks : 64)), dim3(64), 0, c->stream, A);
    } }
  HIPCHK(hipGetLastError());
  *d_out = send;
  return finish(c, 0);
}

extern "C" int shk_route_reserve(shk_ctx *c) {
  if (!c) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  for (int b = 0; b < 2; b++)
    if (!c->d_send[b] && dmalloc(&c->d_send[b], c->cfg.max_batch_keys + 1)) return SHK_ERR_HIP;
  // (and the records of a deNoise point: a shard's rounds are decided outside the context)
  { int rc = ensure_chist(c); if (!rc) rc = point_alloc(c); if (rc) return rc; }
  return SHK_OK;
}

extern "C" int shk_stage_words_pair(shk_ctx *c, const uint64_t *d_a, uint64_t na, const uint64_t *d_b, uint64_t nb) {
  if (!c || (!d_a && na) || (!d_b && nb)) return SHK_ERR_ARG;
  if (na == 0) return shk_stage_words(c, d_b, nb);
  if (nb == 0) return shk_stage_words(c, d_a, na);
  if (na + nb > c->cfg.max_batch_keys) return SHK_ERR_BATCH;
  // (the first level writes d_words[0] while it reads both sources)
  { const uint64_t *o0 = c->d_words[0], *o1 = c->d_words[0] + c->cfg.max_batch_keys + 1;
    if ((d_a + na > o0 && d_a < o1) || (d_b + nb > o0 && d_b < o1)) return SHK_ERR_ARG; }
  if (c->nlevels == 0) return SHK_ERR_ARG;       // (a single region has no partition to read two sources: concatenate)
  HIPCHK(hipSetDevice(c->dev));
  c->h_pinned[43] = na + nb;
  c->h_pinned[56] = na; c->h_pinned[57] = nb; c->h_pinned[58] = 0; c->h_pinned[59] = na; c->h_pinned[60] = 0; c->h_pinned[61] = nb;
  HIPCHK(hipMemcpyAsync(c->d_scalars + 1, c->h_pinned + 43, 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_scalars + 40, c->h_pinned + 56, 48, hipMemcpyHostToDevice, c->stream));
  c->stage2_b = d_b; c->stage2_na = na; c->stage2_nb = nb;
  int dst = 0;
  int rc = partition_stage(c, 0, na + nb, &dst, d_a);
  c->stage2_b = nullptr;
  c->staged = dst;
  return finish(c, rc);
}

extern "C" int shk_stage_words(shk_ctx *c, const uint64_t *d_words, uint64_t nwords) {
  if (!c || (!d_words && nwords)) return SHK_ERR_ARG;
  if (nwords > c->cfg.max_batch_keys) return SHK_ERR_BATCH;
  HIPCHK(hipSetDevice(c->dev));
  c->h_pinned[43] = nwords;
  HIPCHK(hipMemcpyAsync(c->d_scalars + 1, c->h_pinned + 43, 8, hipMemcpyHostToDevice, c->stream));
  int dst = 0;
  // the first partition level reads the caller's buffer in place (no staging copy)
  int rc = d_words == c->d_words[0] ? partition_stage(c, 0, nwords, &dst)
         : d_words == c->d_words[1] ? partition_stage(c, 1, nwords, &dst)
                                    : partition_stage(c, 0, nwords, &dst, d_words);
  c->staged = dst;
  return finish(c, rc);
}

extern "C" int shk_stage_summary(shk_ctx *c, uint32_t lo, uint32_t hi, uint32_t hist_base, uint32_t hist_shift,
                                 int want_hist, shk_summary *out) {
  if (!c || !out || hi < lo || hi >= SHK_MAX_CHUNKS) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  MergeOut o;
  int rc = merge_summary(c, c->d_words[c->staged], lo, hi, hist_base, hist_shift, 0, &o, want_hist, 1);
  if (!rc && !c->big_image && (o.err & (SHK_E_OLD_EXTENT | SHK_E_NEW_EXTENT)) && !(o.err & SHK_E_TABLE_FULL)) {
    c->big_image = 1;   // a cluster outgrew the small LDS image: same range with the big one
    rc = merge_summary(c, c->d_words[c->staged], lo, hi, hist_base, hist_shift, 0, &o, want_hist, 1);
  }
  prof_collect(c);
  if (rc) return rc;
  out->new_distinct = o.newd; out->added = o.added; out->removed = o.removed; out->before = o.before;
  for (int i = 0; i < SHK_HIST_BINS; i++) out->hist[i] = o.hist[i];
  out->err_bits = o.err; out->reserved = 0;
  return SHK_OK;
}

extern "C" int shk_stage_commit(shk_ctx *c, uint32_t lo, uint32_t hi, const shk_summary *s) {
  if (!c || !s || hi < lo) return SHK_ERR_ARG;
  if (s->err_bits) return map_err_bits(s->err_bits);
  HIPCHK(hipSetDevice(c->dev));
  int rc = merge_write(c, c->d_words[c->staged], lo, hi, 0);
  if (!rc) { c->ndistinct += s->new_distinct; c->nelts += s->added; }
  return finish(c, rc);
}

extern "C" int shk_stage_try(shk_ctx *c, uint32_t lo, uint32_t hi, uint32_t hist_base, uint32_t hist_shift, int want_hist,
                             shk_summary *out) {
  if (!c || !out || hi < lo || hi >= SHK_MAX_CHUNKS) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  MergeOut o;
  int rc;
  for (int attempt = 0; attempt < 2; attempt++) {
    if (c->single_ok) rc = merge_single(c, c->d_words[c->staged], lo, hi, 0, &o, want_hist, hist_base, hist_shift);
    else {
      rc = merge_summary(c, c->d_words[c->staged], lo, hi, hist_base, hist_shift, 0, &o, want_hist, 1);
      if (!c->use_spill) o.err |= SHK_E_LOOKBACK;   // plain two-launch scheme: the caller commits through shk_stage_commit
    }
    if (rc || c->big_image || !(o.err & (SHK_E_OLD_EXTENT | SHK_E_NEW_EXTENT)) || (o.err & SHK_E_TABLE_FULL)) break;
    c->big_image = 1;
  }
  prof_collect(c);
  if (rc) return rc;
  out->new_distinct = o.newd; out->added = o.added; out->removed = o.removed; out->before = o.before;
  for (int i = 0; i < SHK_HIST_BINS; i++) out->hist[i] = o.hist[i];
  out->err_bits = o.err; out->reserved = 0;
  return SHK_OK;
}

extern "C" int shk_stage_accept(shk_ctx *c, const shk_summary *s) {
  if (!c || !s) return SHK_ERR_ARG;
  if (s->err_bits) return map_err_bits(s->err_bits);
  HIPCHK(hipSetDevice(c->dev));
  if (c->single_ok) commit_single(c);
  else {
    if (!c->spill_valid) return SHK_ERR_ARG;   // nothing was tried
    const int dn = c->spill_denoise;
    int rc = merge_write(c, c->spill_words, c->spill_lo, c->spill_hi, dn);
    if (rc) return finish(c, rc);
    if (dn) { c->big_image = 0; c->rounds_done++; }
  }
  c->ndistinct += s->new_distinct; c->nelts += s->added;
  c->ndistinct -= s->removed; c->nelts -= s->removed;     // (only a deNoise try removes anything)
  return finish(c, SHK_OK);
}

// deNoise round on this shard fused with the insertion of the staged chunks [lo, hi] (the chunks behind the deNoise
// point): marks + statistics pass; nothing is written until shk_stage_accept. s->removed = singletons dropped,
// s->new_distinct counts dropped keys that reappear in [lo, hi] as new. Needs the spill scheme (default).
extern "C" int shk_stage_try_denoise(shk_ctx *c, uint32_t lo, uint32_t hi, shk_summary *out) {
  if (!c || !out || hi < lo || hi >= SHK_MAX_CHUNKS) return SHK_ERR_ARG;
  if (!c->use_spill) {        // (another rebuild scheme was asked for: the caller runs shk_denoise and goes on)
    memset(out, 0, sizeof(*out));
    out->err_bits = SHK_E_FUSED;
    return SHK_OK;
  }
  HIPCHK(hipSetDevice(c->dev));
  uint64_t ml = c->cfg.min_denoise_len ? c->cfg.min_denoise_len : (1ULL << 20);
  { ProfScope ps(c, KP_MARKS);
    hipLaunchKernelGGL(k_denoise_marks, dim3(1), dim3(64), 0, c->stream, c->tab[c->cur], c->nslots, c->xnslots, c->nblocks,
                       ml, (unsigned long long *)(c->d_scalars + 3)); }
  MergeOut o;
  int rc = merge_summary(c, c->d_words[c->staged], lo, hi, lo, 0, 1, &o, 0, 1);
  prof_collect(c);
  if (rc) return rc;
  out->new_distinct = o.newd; out->added = o.added; out->removed = o.removed; out->before = o.before;
  for (int i = 0; i < SHK_HIST_BINS; i++) out->hist[i] = o.hist[i];
  out->err_bits = o.err; out->reserved = 0;
  return SHK_OK;
}

extern "C" int shk_stage_chunk_hist(shk_ctx *c, uint64_t *out, uint32_t n) {
  if (!c || !out || !c->chist_n || n > c->chist_n) return SHK_ERR_ARG;
  memcpy(out, c->h_chist, (size_t)n * sizeof(uint64_t));
  return SHK_OK;
}

// ---- one-pass deNoise point on a shard (shk/dist.py: the ranks take the steps together)
extern "C" int shk_stage_sample(shk_ctx *c, uint32_t lo, uint32_t hi, uint64_t *hist, uint32_t *regions, uint32_t *sampled,
                                uint32_t *err_bits) {
  if (!c || !hist || !regions || !sampled || !err_bits || hi < lo || hi >= SHK_MAX_CHUNKS) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  uint32_t ns = 0, err = 0;
  int rc = sample_pass(c, c->d_words[c->staged], lo, hi, &ns, &err);
  prof_collect(c);
  if (rc) return rc;
  memcpy(hist, c->h_chist, ((size_t)hi + 1) * sizeof(uint64_t));
  *regions = c->nregions; *sampled = ns; *err_bits = err;
  return SHK_OK;
}

extern "C" int shk_stage_point_try(shk_ctx *c, uint32_t lo, uint32_t split, uint32_t hi, shk_point *out) {
  if (!c || !out || hi < lo || split + 1 < lo || split > hi || hi >= SHK_MAX_CHUNKS) return SHK_ERR_ARG;   // (split = lo - 1: the round comes first)
  HIPCHK(hipSetDevice(c->dev));
  c->pt_valid = 0;
  memset(out, 0, sizeof(*out));
  // (needs the spill scheme; the retry image is not instantiated for this pass: the caller takes another path)
  if (c->big_image || !c->use_spill || c->single_ok) { out->err_bits = SHK_E_FUSED; return SHK_OK; }
  PointOut po;
  int rc = point_try(c, c->d_words[c->staged], lo, split, hi, true, &po);
  prof_collect(c);
  if (rc) return rc;
  out->new_after = po.newd_after; out->added_after = po.added_after; out->removed = po.removed; out->added_before = po.added_before;
  out->err_bits = po.err; out->first_used = (uint32_t)po.first_used; out->islots = po.islots; out->ifin = po.ifin;
  if (!po.err) { c->pt_lo = lo; c->pt_split = split; c->pt_hi = hi; c->pt_valid = 1; c->pt_nprot = 0; c->pt_words = c->d_words[c->staged]; }
  return SHK_OK;
}

// a deNoise round on its own (no words), taken the same way: the range walk then runs over the single table's layout
extern "C" int shk_stage_round_try(shk_ctx *c, shk_point *out) {
  if (!c || !out) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  c->pt_valid = 0;
  memset(out, 0, sizeof(*out));
  // (needs the spill scheme; the retry image is not instantiated for this pass: the caller takes another path)
  if (c->big_image || !c->use_spill || c->single_ok) { out->err_bits = SHK_E_FUSED; return SHK_OK; }
  PointOut po;
  int rc = point_try(c, nullptr, 0, 0, 0, false, &po);
  prof_collect(c);
  if (rc) return rc;
  out->new_after = po.newd_after; out->added_after = po.added_after; out->removed = po.removed; out->added_before = po.added_before;
  out->err_bits = po.err; out->first_used = (uint32_t)po.first_used; out->islots = po.islots; out->ifin = po.ifin;
  if (!po.err) { c->pt_lo = 1; c->pt_split = 0; c->pt_hi = 0; c->pt_valid = 1; c->pt_nprot = 0; c->pt_words = nullptr; }
  return SHK_OK;
}

extern "C" int shk_stage_point_walk(shk_ctx *c, int64_t carry, int64_t prev_fp, int last, int next_first_used,
                                    const uint64_t state_in[2], uint64_t state_out[2], uint64_t *nprot, uint32_t *err_bits) {
  if (!c || !state_in || !state_out || !nprot || !err_bits || carry < 0 || !c->pt_valid) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  ShkWalkShard W;
  W.prev_fp = prev_fp; W.cap_local = c->g_nslots - c->q_lo; W.last = last; W.next_first_used = next_first_used;
  int rc = point_walk(c, carry, W, state_in, state_out, nprot, err_bits);
  prof_collect(c);
  if (rc) return rc;
  c->pt_nprot = *nprot;
  return SHK_OK;
}

extern "C" int shk_stage_point_finish(shk_ctx *c, shk_point *out, shk_summary *accept) {
  if (!c || !out || !accept || !c->pt_valid || (c->pt_words && !c->chist_n)) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  PointOut po;
  po.newd_after = out->new_after; po.added_after = out->added_after; po.removed = out->removed; po.added_before = out->added_before;
  po.err = 0;
  const bool round_only = c->pt_words == nullptr;
  int rc = point_finish(c, c->pt_words, round_only ? 0 : c->pt_lo, c->pt_split, c->pt_hi, c->pt_nprot, &po);
  prof_collect(c);
  c->pt_valid = 0;
  if (rc) return rc;
  out->new_after = po.newd_after; out->added_after = po.added_after; out->removed = po.removed; out->added_before = po.added_before;
  out->err_bits = po.err;
  uint64_t newd_before = 0;
  if (!round_only) for (uint32_t ch = c->pt_lo; ch <= c->pt_split; ch++) newd_before += c->h_chist[ch];
  memset(accept, 0, sizeof(*accept));
  accept->new_distinct = newd_before + po.newd_after; accept->added = po.added_before + po.added_after; accept->removed = po.removed;
  accept->err_bits = po.err;
  return SHK_OK;
}

extern "C" int shk_denoise(shk_ctx *c, uint64_t *removed) {
  if (!c) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  uint64_t r = 0;
  int rc = denoise_round(c, &r);
  if (!rc) c->rounds_done++;
  if (removed) *removed = r;
  return finish(c, rc);
}

extern "C" int shk_stats(shk_ctx *c, shk_totals *o) {
  if (!c || !o) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  o->nelts = c->nelts; o->ndistinct = c->ndistinct; o->rounds_left = c->rounds_left; o->rounds_done = c->rounds_done;
  o->nslots = c->nslots; o->xnslots = c->xnslots; o->nblocks = c->nblocks; o->table_bytes = c->table_bytes;
  HIPCHK(hipMemcpyAsync(c->h_pinned + 44, c->fin[c->cur] + c->nregions, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  o->free_pointer = c->h_pinned[44];
  return SHK_OK;
}

// quotient_filter_metadata, gqf.h:62-77 (field offsets checked against the compiled
// reference: see oracle/cqf_oracle.c orc_qf_header)
extern "C" int shk_header(shk_ctx *c, uint8_t out[128]) {
  if (!c || !out) return SHK_ERR_ARG;
  memset(out, 0, 128);
  uint64_t v;
#define PUT(off, val) do { v = (val); memcpy(out + (off), &v, 8); } while (0)
  PUT(0, c->table_bytes);
  memcpy(out + 8, &c->cfg.seed, 4);
  PUT(16, c->nslots); PUT(24, c->xnslots); PUT(32, (uint64_t)c->cfg.hb); PUT(40, 0);
  PUT(48, 8); PUT(56, 8);
  { unsigned __int128 range = (unsigned __int128)c->nslots << 8; memcpy(out + 64, &range, 16); }
  PUT(80, c->nblocks); PUT(88, c->nelts); PUT(96, c->ndistinct); PUT(104, 0);
  PUT(112, c->xnslots / (1ULL << 16) + 2);
#undef PUT
  return SHK_OK;
}

extern "C" int shk_export_blocks(shk_ctx *c, void *dst, uint64_t cap) {
  if (!c || !dst || cap < c->table_bytes) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  HIPCHK(hipMemcpyAsync(dst, c->tab[c->cur], c->table_bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return SHK_OK;
}

// device pointer and size of the live table (valid until the next call that rebuilds it): lets a caller move a
// shard's table to the GPU that stitches them without a trip through host memory
extern "C" int shk_table_ptr(shk_ctx *c, void **d_table, uint64_t *nbytes) {
  if (!c || !d_table || !nbytes) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  HIPCHK(hipStreamSynchronize(c->stream));
  *d_table = c->tab[c->cur]; *nbytes = c->table_bytes;
  return SHK_OK;
}

extern "C" int shk_export_cqf(shk_ctx *c, const char *path) {
  if (!c || !path) return SHK_ERR_ARG;
  std::vector<uint8_t> buf(c->table_bytes);
  int rc = shk_export_blocks(c, buf.data(), buf.size());
  if (rc) return rc;
  uint8_t hdr[128];
  shk_header(c, hdr);
  FILE *f = fopen(path, "wb+");
  if (!f) return SHK_ERR_IO;
  size_t ok = fwrite(hdr, 128, 1, f);
  ok += fwrite(buf.data(), buf.size(), 1, f);
  fclose(f);
  return ok == 2 ? SHK_OK : SHK_ERR_IO;
}

extern "C" int shk_import_blocks(shk_ctx *c, const void *src, uint64_t nbytes, uint64_t nelts, uint64_t ndistinct) {
  if (!c || !src || nbytes != c->table_bytes) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  HIPCHK(hipMemcpyAsync(c->tab[c->cur], src, nbytes, hipMemcpyHostToDevice, c->stream));
  { ProfScope ps(c, KP_MISC);
    hipLaunchKernelGGL(k_build_fin, dim3(c->nregions / 256 + 1), dim3(256), 0, c->stream, c->tab[c->cur], c->nslots,
                       c->nregions, c->fin[c->cur]); }
  HIPCHK(hipGetLastError());
  c->nelts = nelts; c->ndistinct = ndistinct;
  return finish(c, 0);
}

extern "C" int shk_import_cqf(shk_ctx *c, const char *path) {
  if (!c || !path) return SHK_ERR_ARG;
  FILE *f = fopen(path, "rb");
  if (!f) return SHK_ERR_IO;
  uint8_t hdr[128];
  if (fread(hdr, 128, 1, f) != 1) { fclose(f); return SHK_ERR_IO; }
  uint64_t size, nslots, key_bits, bps, nelts, nd;
  memcpy(&size, hdr + 0, 8); memcpy(&nslots, hdr + 16, 8); memcpy(&key_bits, hdr + 32, 8); memcpy(&bps, hdr + 56, 8);
  memcpy(&nelts, hdr + 88, 8); memcpy(&nd, hdr + 96, 8);
  if (bps != 8 || nslots != c->nslots || key_bits != c->cfg.hb || size != c->table_bytes || c->q_lo != 0) { fclose(f); return SHK_ERR_ARG; }
  std::vector<uint8_t> buf(size);
  if (fread(buf.data(), size, 1, f) != 1) { fclose(f); return SHK_ERR_IO; }
  fclose(f);
  return shk_import_blocks(c, buf.data(), size, nelts, nd);
}

extern "C" int shk_lookup(shk_ctx *c, const uint64_t *keys, uint64_t n, int on_device, int mode, uint64_t *counts,
                          uint8_t *was_traveled) {
  if (!c || (n && (!keys || !counts)) || mode < 0 || mode > 2) return SHK_ERR_ARG;
  if (n == 0) return SHK_OK;
  HIPCHK(hipSetDevice(c->dev));
  uint64_t *dk = nullptr, *dc = nullptr; uint8_t *dt = nullptr;
  if (on_device) { dk = (uint64_t *)keys; dc = counts; dt = was_traveled; }
  else {
    HIPCHK(hipMalloc((void **)&dk, n * 8)); HIPCHK(hipMalloc((void **)&dc, n * 8)); HIPCHK(hipMalloc((void **)&dt, n));
    HIPCHK(hipMemcpyAsync(dk, keys, n * 8, hipMemcpyHostToDevice, c->stream));
  }
  { ProfScope ps(c, KP_LOOKUP);
    hipLaunchKernelGGL(k_lookup, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, c->tab[c->cur], dk, n, c->q_lo,
                       c->nslots, mode, dc, dt); }
  HIPCHK(hipGetLastError());
  if (!on_device) {
    HIPCHK(hipMemcpyAsync(counts, dc, n * 8, hipMemcpyDeviceToHost, c->stream));
    if (was_traveled) HIPCHK(hipMemcpyAsync(was_traveled, dt, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    hipFree(dk); hipFree(dc); hipFree(dt);
  }
  return finish(c, 0);
}

// ------------------------------------------------------------------ counted inserts, iterator dump, merge, stitch
// (insert_advance with count > 1 gqf.c:2024-2136; qf_iterator/qfi_* :2474-2601; qf_merge/qf_multi_merge :2614-2704)

// rebuild with the words of a counted insert; nothing is committed unless the pass is clean
static int merge_plain(shk_ctx *c, const uint64_t *words, MergeOut *o) {
  for (int attempt = 0; attempt < 2; attempt++) {
    int rc = merge_summary(c, words, 0, SHK_MAX_CHUNKS - 1, 0, 0, 0, o, 0, 1);
    if (rc) return rc;
    if (!c->big_image && (o->err & (SHK_E_OLD_EXTENT | SHK_E_NEW_EXTENT)) && !(o->err & SHK_E_TABLE_FULL)) { c->big_image = 1; continue; }
    break;
  }
  if (o->err) return SHK_OK;   // the caller looks at o->err
  return merge_write(c, words, 0, SHK_MAX_CHUNKS - 1, 0);
}

extern "C" int shk_insert_counted(shk_ctx *c, const uint64_t *keys, const uint64_t *counts, uint64_t n, int on_device,
                                  shk_batch_stats *stats) {
  if (!c || (n && (!keys || !counts))) return SHK_ERR_ARG;
  shk_batch_stats st;
  memset(&st, 0, sizeof(st));
  if (stats) *stats = st;
  if (n == 0) return SHK_OK;
  HIPCHK(hipSetDevice(c->dev));
  uint64_t *dk = nullptr, *dc = nullptr, *doff = nullptr;
  uint32_t *dnw = nullptr;
  int rc = SHK_OK;
  struct Free { uint64_t *&a, *&b, *&o; uint32_t *&w; bool own; ~Free() { if (own) { hipFree(a); hipFree(b); } hipFree(o); hipFree(w); } } fr{dk, dc, doff, dnw, !on_device};
  if (on_device) { dk = (uint64_t *)keys; dc = (uint64_t *)counts; }
  else {
    if (dmalloc(&dk, n) || dmalloc(&dc, n)) return SHK_ERR_HIP;
    HIPCHK(hipMemcpyAsync(dk, keys, n * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dc, counts, n * 8, hipMemcpyHostToDevice, c->stream));
  }
  // pairs are inserted slice by slice (bounded by the key-word capacity; halved when one region would receive more
  // distinct new keys than its LDS hash holds); occurrences beyond `take` of one key go into further passes
  const uint64_t take = 1ULL << 22;
  const uint64_t key_lo = c->q_lo << 8, key_hi = (c->q_lo + c->nslots) << 8;
  uint64_t slice = n;
  { const uint64_t cap = c->cfg.max_batch_keys / 2 > 0 ? c->cfg.max_batch_keys / 2 : 1; if (slice > cap) slice = cap; }
  if (dmalloc(&doff, slice + 2) || dmalloc(&dnw, slice + 2)) return SHK_ERR_HIP;
  c->counted = 1;
  struct CountedScope { shk_ctx *c; ~CountedScope() { c->counted = 0; } } counted_scope{c};   // every exit, incl. HIPCHK's
  uint64_t done = 0;
  while (done < n && !rc) {
    const uint64_t m = n - done < slice ? n - done : slice;
    uint64_t skip = 0, maxc = 0;
    bool halve = false;
    do {
      HIPCHK(hipMemsetAsync(c->d_scalars + 4, 0, 8, c->stream));
      const uint32_t nb = (uint32_t)((m + 255) / 256);
      { ProfScope ps(c, KP_MISC);
        hipLaunchKernelGGL(k_expand_counted<0>, dim3(nb), dim3(256), 0, c->stream, dk + done, dc + done, m, skip, take, c->cfg.hb, dnw,
                           (const uint64_t *)nullptr, (uint64_t *)nullptr, c->d_err, key_lo, key_hi, (unsigned long long *)(c->d_scalars + 4)); }
      if (run_scan<uint32_t>(c, dnw, m, nullptr, doff)) { rc = SHK_ERR_HIP; break; }
      HIPCHK(hipMemcpyAsync(c->h_pinned + 45, doff + m, 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipMemcpyAsync(c->h_pinned + 46, c->d_scalars + 4, 8, hipMemcpyDeviceToHost, c->stream));
      uint32_t bits = 0;
      if (fetch_err(c, &bits)) { rc = SHK_ERR_HIP; break; }
      if (bits) { rc = map_err_bits(bits); break; }
      const uint64_t nwords = c->h_pinned[45];
      maxc = c->h_pinned[46];
      if (nwords > c->cfg.max_batch_keys) { halve = true; break; }
      if (nwords) {
        { ProfScope ps(c, KP_MISC);
          hipLaunchKernelGGL(k_expand_counted<1>, dim3(nb), dim3(256), 0, c->stream, dk + done, dc + done, m, skip, take, c->cfg.hb, dnw,
                             doff, c->d_words[0], c->d_err, key_lo, key_hi, (unsigned long long *)nullptr); }
        c->h_pinned[43] = nwords;
        HIPCHK(hipMemcpyAsync(c->d_scalars + 1, c->h_pinned + 43, 8, hipMemcpyHostToDevice, c->stream));
        int dst = 0;
        rc = partition_stage(c, 0, nwords, &dst);
        if (rc) break;
        MergeOut o;
        rc = merge_plain(c, c->d_words[dst], &o);
        if (rc) break;
        if (o.err & SHK_E_HASH_FULL) { if (skip) { rc = SHK_ERR_REGION; break; } halve = true; break; }
        if (o.err) { rc = map_err_bits(o.err); break; }
        c->nelts += o.added; c->ndistinct += o.newd;
        st.kmers += o.added; st.new_distinct += o.newd;
      }
      skip += take;
    } while (maxc > skip);
    if (rc) break;
    if (halve) {
      if (slice == 1) { rc = SHK_ERR_REGION; break; }
      slice = (slice + 1) / 2;
      continue;
    }
    done += m;
  }
  c->counted = 0;
  if (stats) *stats = st;
  return finish(c, rc);
}

// (key, count) of every entry in the order of the reference's iterator. keys == NULL: only the number of entries.
// ref_iterator_end != 0: *n_out is where the reference's own iteration would END (see k_region_dump).
extern "C" int shk_dump(shk_ctx *c, uint64_t *keys, uint64_t *counts, uint64_t cap, int on_device, int ref_iterator_end,
                        uint64_t *n_out) {
  if (!c || !n_out || (keys && !counts)) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  ShkMergeArgs A;
  fill_args(c, &A, nullptr, 0, 0, 0, 0, 0);
  uint32_t *nper = c->d_over_list;                       // scratch of the spill scheme: [nregions + 1]
  uint64_t *offs = (uint64_t *)c->d_lb_agg;              // [nregions + 2]
  unsigned long long *stop = (unsigned long long *)(c->d_scalars + 5);
  const uint64_t *no_offs = nullptr;
  uint64_t *no_out = nullptr;
  c->spill_valid = 0;
  for (int attempt = 0; attempt < 2; attempt++) {
    { ProfScope ps(c, KP_MISC);
      SHK_FOR_REGION_SLICES(c, A, nblk) {
        if (c->big_image) hipLaunchKernelGGL((k_region_dump<0, SHK_IMG_BLOCKS_BIG>), dim3(nblk), dim3(SHK_WAVE), 0, c->stream, A, nper, no_offs, no_out, no_out, 0ULL, (unsigned long long *)nullptr);
        else hipLaunchKernelGGL((k_region_dump<0, SHK_IMG_BLOCKS>), dim3(nblk), dim3(SHK_WAVE), 0, c->stream, A, nper, no_offs, no_out, no_out, 0ULL, (unsigned long long *)nullptr);
      } }
    uint32_t bits = 0;
    if (fetch_err(c, &bits)) return SHK_ERR_HIP;
    if ((bits & SHK_E_OLD_EXTENT) && !c->big_image) { c->big_image = 1; c->last_err_bits = 0; continue; }
    if (bits) { prof_collect(c); return map_err_bits(bits); }
    break;
  }
  if (run_scan<uint32_t>(c, nper, c->nregions, nullptr, offs)) return SHK_ERR_HIP;
  HIPCHK(hipMemcpyAsync(c->h_pinned + 45, offs + c->nregions, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  const uint64_t total = c->h_pinned[45];
  *n_out = total;
  if (!keys && !ref_iterator_end) return finish(c, 0);
  const uint64_t m = keys ? (total < cap ? total : cap) : 0;
  uint64_t *dk = keys, *dc = counts;
  if (!on_device && m) { if (dmalloc(&dk, m) || dmalloc(&dc, m)) return SHK_ERR_HIP; }
  c->h_pinned[46] = ~0ULL;
  HIPCHK(hipMemcpyAsync(stop, c->h_pinned + 46, 8, hipMemcpyHostToDevice, c->stream));
  if (total) {
    ProfScope ps(c, KP_MISC);
    unsigned long long *sp = ref_iterator_end ? stop : nullptr;
    SHK_FOR_REGION_SLICES(c, A, nblk) {
      if (c->big_image) hipLaunchKernelGGL((k_region_dump<1, SHK_IMG_BLOCKS_BIG>), dim3(nblk), dim3(SHK_WAVE), 0, c->stream, A, nper, (const uint64_t *)offs, m ? dk : no_out, m ? dc : no_out, m, sp);
      else hipLaunchKernelGGL((k_region_dump<1, SHK_IMG_BLOCKS>), dim3(nblk), dim3(SHK_WAVE), 0, c->stream, A, nper, (const uint64_t *)offs, m ? dk : no_out, m ? dc : no_out, m, sp);
    }
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_pinned + 46, stop, 8, hipMemcpyDeviceToHost, c->stream));
  if (!on_device && m) {
    HIPCHK(hipMemcpyAsync(keys, dk, m * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(counts, dc, m * 8, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (!on_device && m) { hipFree(dk); hipFree(dc); }
  if (ref_iterator_end && c->h_pinned[46] < total) *n_out = c->h_pinned[46];
  return finish(c, 0);
}

// dst := canonical table of (dst's entries + the second source's entries), counts of equal keys added
static int merge2_run(shk_ctx *c, const ShkSrc2 &S, uint64_t *newd_out, uint64_t *added_out) {
  ShkMergeArgs A;
  uint64_t newd = 0, added = 0;
  for (int attempt = 0; attempt < 2; attempt++) {
    fill_args(c, &A, nullptr, 0, 0, 0, 0, 0);
    HIPCHK(hipMemsetAsync(c->d_counters, 0, (4 + SHK_HIST_BINS + 1) * 8, c->stream));
    c->spill_valid = 0;
    { ProfScope ps(c, KP_MERGE_SUM);
      SHK_FOR_REGION_SLICES(c, A, nblk) {
        if (c->big_image) hipLaunchKernelGGL((k_region_merge2<false, SHK_IMG_BLOCKS_BIG>), dim3(nblk), dim3(SHK_WAVE), 0, c->stream, A, S);
        else hipLaunchKernelGGL((k_region_merge2<false, SHK_IMG_BLOCKS>), dim3(nblk), dim3(SHK_WAVE), 0, c->stream, A, S);
      } }
    { ProfScope ps(c, KP_REGION_SCAN);
      const uint32_t ntiles = (c->nregions + SHK_RSCAN_TILE - 1) / SHK_RSCAN_TILE;
      hipLaunchKernelGGL(k_region_scan_a, dim3(ntiles), dim3(c->threads), 0, c->stream, c->d_summary, c->nregions, c->d_tile_a, c->d_tile_b);
      hipLaunchKernelGGL(k_region_scan_b, dim3(1), dim3(c->threads), 0, c->stream, c->d_tile_a, c->d_tile_b, ntiles, c->d_tile_f);
      hipLaunchKernelGGL(k_region_scan_c, dim3(ntiles), dim3(c->threads), 0, c->stream, c->d_summary, c->nregions, c->d_tile_f,
                         c->xnslots, (uint32_t)(c->big_image ? SHK_IMG_BLOCKS_BIG * 64 : SHK_IMG_SLOTS), c->fin[c->cur ^ 1], c->d_counters, c->d_err); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_pinned, c->d_counters, 4 * 8, hipMemcpyDeviceToHost, c->stream));
    uint32_t bits = 0;
    if (fetch_err(c, &bits)) return SHK_ERR_HIP;
    if (!c->big_image && (bits & (SHK_E_OLD_EXTENT | SHK_E_NEW_EXTENT)) && !(bits & SHK_E_TABLE_FULL)) { c->big_image = 1; c->last_err_bits = 0; continue; }
    if (bits) return map_err_bits(bits);
    newd = c->h_pinned[0]; added = c->h_pinned[1];
    break;
  }
  HIPCHK(hipMemsetAsync(c->tab[c->cur ^ 1], 0, c->table_bytes, c->stream));
  { ProfScope ps(c, KP_MERGE_WRITE);
    SHK_FOR_REGION_SLICES(c, A, nblk) {
      if (c->big_image) hipLaunchKernelGGL((k_region_merge2<true, SHK_IMG_BLOCKS_BIG>), dim3(nblk), dim3(SHK_WAVE), 0, c->stream, A, S);
      else hipLaunchKernelGGL((k_region_merge2<true, SHK_IMG_BLOCKS>), dim3(nblk), dim3(SHK_WAVE), 0, c->stream, A, S);
    } }
  HIPCHK(hipGetLastError());
  c->cur ^= 1;
  *newd_out = newd; *added_out = added;
  return SHK_OK;
}

extern "C" int shk_merge(shk_ctx *dst, shk_ctx *src, shk_batch_stats *stats) {
  if (!dst || !src || dst == src) return SHK_ERR_ARG;
  if (dst->dev != src->dev || dst->cfg.qb != src->cfg.qb || dst->cfg.hb != src->cfg.hb || dst->q_lo != src->q_lo || dst->nslots != src->nslots)
    return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(dst->dev));
  HIPCHK(hipStreamSynchronize(src->stream));   // the source's table must be at rest
  ShkSrc2 S;
  memset(&S, 0, sizeof(S));
  S.tab[0] = src->tab[src->cur]; S.fin[0] = src->fin[src->cur]; S.nblocks = src->nblocks; S.regions_per_src = dst->nregions; S.nsrc = 1;
  uint64_t newd = 0, added = 0;
  int rc = merge2_run(dst, S, &newd, &added);
  if (!rc) {
    dst->nelts += added; dst->ndistinct += newd;
    if (stats) { memset(stats, 0, sizeof(*stats)); stats->kmers = added; stats->new_distinct = newd; }
  }
  return finish(dst, rc);
}

extern "C" int shk_multi_merge(shk_ctx *dst, shk_ctx *const *srcs, uint32_t n, shk_batch_stats *stats) {
  if (!dst || (n && !srcs)) return SHK_ERR_ARG;
  shk_batch_stats tot;
  memset(&tot, 0, sizeof(tot));
  for (uint32_t i = 0; i < n; i++) {
    shk_batch_stats st;
    int rc = shk_merge(dst, srcs[i], &st);
    if (rc) return rc;
    tot.kmers += st.kmers; tot.new_distinct += st.new_distinct;
  }
  if (stats) *stats = tot;
  return SHK_OK;
}

// The whole filter from its quotient-range shards: shard s (a table in the layout shk_export_blocks gives for a context
// with num_shards = nshards, shard_index = s: its nslots / nshards quotients plus its own overflow tail) supplies the
// runs of its quotients; they are laid out again in the single table, where a cluster may now run across a shard border.
extern "C" int shk_import_shards(shk_ctx *c, const void *const *shard_blocks, const uint64_t *shard_bytes, uint32_t nshards,
                                 int on_device, uint64_t nelts, uint64_t ndistinct) {
  if (!c || !shard_blocks || !shard_bytes || nshards == 0 || nshards > SHK_MAX_SRC || (nshards & (nshards - 1))) return SHK_ERR_ARG;
  if (c->cfg.num_shards > 1 || c->q_lo != 0) return SHK_ERR_ARG;
  const uint64_t s_nslots = c->nslots / nshards;
  if (s_nslots < SHK_REGION || s_nslots % SHK_REGION) return SHK_ERR_ARG;
  const uint64_t s_xnslots = s_nslots + (uint64_t)(10 * sqrt((double)c->g_nslots));
  const uint64_t s_nblocks = (s_xnslots + 63) / 64, s_bytes = s_nblocks * SHK_BLOCK_BYTES;
  const uint32_t s_nregions = (uint32_t)(s_nslots / SHK_REGION);
  for (uint32_t s = 0; s < nshards; s++) if (shard_bytes[s] != s_bytes || !shard_blocks[s]) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  // start from an empty table
  HIPCHK(hipMemsetAsync(c->tab[c->cur], 0, c->table_bytes + SHK_SLACK, c->stream));
  HIPCHK(hipMemsetAsync(c->fin[c->cur], 0, ((uint64_t)c->nregions + 2) * 8, c->stream));
  c->nelts = 0; c->ndistinct = 0;
  ShkSrc2 S;
  memset(&S, 0, sizeof(S));
  S.nblocks = s_nblocks; S.regions_per_src = s_nregions; S.nsrc = nshards;
  std::vector<uint8_t *> own;
  std::vector<uint64_t *> fins;
  int rc = SHK_OK;
  for (uint32_t s = 0; s < nshards && !rc; s++) {
    uint8_t *dt = (uint8_t *)shard_blocks[s];
    if (!on_device) {
      if (dmalloc(&dt, s_bytes)) { rc = SHK_ERR_HIP; break; }
      own.push_back(dt);
      if (hipMemcpyAsync(dt, shard_blocks[s], s_bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = SHK_ERR_HIP; break; }
      if (hipMemsetAsync(dt + s_bytes, 0, SHK_SLACK, c->stream) != hipSuccess) { rc = SHK_ERR_HIP; break; }
    }
    uint64_t *df = nullptr;
    if (dmalloc(&df, (uint64_t)s_nregions + 2)) { rc = SHK_ERR_HIP; break; }
    fins.push_back(df);
    hipLaunchKernelGGL(k_build_fin, dim3(s_nregions / 256 + 1), dim3(256), 0, c->stream, dt, s_nslots, s_nregions, df);
    S.tab[s] = dt; S.fin[s] = df;
  }
  uint64_t newd = 0, added = 0;
  if (!rc) rc = merge2_run(c, S, &newd, &added);
  hipStreamSynchronize(c->stream);
  for (auto p : own) hipFree(p);
  for (auto p : fins) hipFree(p);
  if (!rc) { c->nelts = nelts ? nelts : added; c->ndistinct = ndistinct ? ndistinct : newd; }
  return finish(c, rc);
}

// ------------------------------------------------------------------ Contiger: seeds (processDataChunk, contig_assembly.cpp:1839-1884)
extern "C" int shk_select_seeds(shk_ctx *c, const void *text, int text_on_device, uint64_t text_bytes, const uint64_t *chunk_off,
                                const uint64_t *chunk_len, uint32_t nchunks, uint32_t k, uint64_t count_min, uint64_t count_max,
                                int use_traveled, char *out_seeds, uint32_t *out_counts, uint32_t cap, uint32_t *n_out) {
  if (!c || !text || !out_seeds || !out_counts || !n_out || nchunks == 0 || nchunks > SHK_MAX_CHUNKS) return SHK_ERR_ARG;
  if (k < 2 || k > SHK_WALK_MAX_K) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  const uint8_t *dtext;
  uint64_t nreads;
  int rc = parse_stage(c, text, text_on_device, text_bytes, chunk_off, chunk_len, nchunks, &dtext, &nreads);
  if (rc) return finish(c, rc);
  *n_out = 0;
  if (nreads == 0) return finish(c, 0);
  struct Scratch { char *s = nullptr; uint32_t *c = nullptr; ~Scratch() { hipFree(s); hipFree(c); } } w;
  if (dmalloc(&w.s, nreads * k) || dmalloc(&w.c, nreads)) return SHK_ERR_HIP;
  char *ds = w.s; uint32_t *dc = w.c;
  { ProfScope ps(c, KP_WALK);
    hipLaunchKernelGGL(k_select_seeds, dim3((uint32_t)((nreads + 255) / 256)), dim3(256), 0, c->stream, c->tab[c->cur], c->q_lo, c->nslots,
                       c->cfg.hb, dtext, c->d_rd_start, c->d_rd_end, (uint64_t)0, nreads, k, count_min, count_max, use_traveled ? 1 : 2, ds, dc); }
  HIPCHK(hipGetLastError());
  std::vector<char> hs(nreads * k);
  std::vector<uint32_t> hc(nreads);
  HIPCHK(hipMemcpyAsync(hs.data(), ds, nreads * k, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(hc.data(), dc, nreads * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint32_t n = 0;
  for (uint64_t r = 0; r < nreads; r++)
    if (hc[r]) {                       // 0 = no seed from this read
      if (n >= cap) return finish(c, SHK_ERR_BATCH);
      memcpy(out_seeds + (size_t)n * k, &hs[r * k], k);
      out_counts[n++] = hc[r];
    }
  *n_out = n;
  return finish(c, 0);
}

// ------------------------------------------------------------------ Contiger: unitig extension (first slice)
extern "C" int shk_extend_forward(shk_ctx *c, const char *cur_kmers, const char *first_kmers, uint32_t n, uint32_t k,
                                  uint64_t abundance_min, int mark_traveled, uint32_t max_ext, char *out_bases,
                                  uint32_t *out_counts, uint32_t *out_n, uint8_t *out_stop, uint8_t *out_branch,
                                  uint32_t *out_ncount) {
  if (!c || (n && (!cur_kmers || !first_kmers || !out_bases || !out_counts || !out_n || !out_stop))) return SHK_ERR_ARG;
  if (k < 2 || k > SHK_WALK_MAX_K || max_ext == 0) return SHK_ERR_ARG;
  if (n == 0) return SHK_OK;
  HIPCHK(hipSetDevice(c->dev));
  // scratch of this call, released on every way out
  struct Scratch {
    char *dk = nullptr, *df = nullptr, *db = nullptr;
    uint32_t *dc = nullptr, *dn = nullptr, *dnc = nullptr;
    uint8_t *ds = nullptr, *dbr = nullptr;
    ~Scratch() { hipFree(dk); hipFree(df); hipFree(db); hipFree(dc); hipFree(dn); hipFree(dnc); hipFree(ds); hipFree(dbr); }
  } w;
  const size_t nk = (size_t)n * k, ne = (size_t)n * max_ext;
  if (dmalloc(&w.dk, nk) || dmalloc(&w.df, nk) || dmalloc(&w.db, ne) || dmalloc(&w.dc, ne) || dmalloc(&w.dn, (size_t)n) || dmalloc(&w.ds, (size_t)n) ||
      dmalloc(&w.dbr, (size_t)n) || dmalloc(&w.dnc, (size_t)n * 8))
    return SHK_ERR_HIP;
  HIPCHK(hipMemcpyAsync(w.dk, cur_kmers, nk, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(w.df, first_kmers, nk, hipMemcpyHostToDevice, c->stream));
  { ProfScope ps(c, KP_WALK);
    char *dk = w.dk, *df = w.df, *db = w.db;          // (plain pointers: the launch must not capture the owning struct)
    uint32_t *dc = w.dc, *dn = w.dn, *dnc = w.dnc;
    uint8_t *ds = w.ds, *dbr = w.dbr;
    hipLaunchKernelGGL(k_extend_forward, dim3((n + 63) / 64), dim3(64), 0, c->stream, c->tab[c->cur], c->q_lo, c->nslots,
                       c->cfg.hb, dk, df, n, k, abundance_min, mark_traveled ? 1 : 2, max_ext, db, dc, dn, ds, dbr, dnc); }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out_bases, w.db, ne, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(out_counts, w.dc, ne * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(out_n, w.dn, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(out_stop, w.ds, n, hipMemcpyDeviceToHost, c->stream));
  if (out_branch) HIPCHK(hipMemcpyAsync(out_branch, w.dbr, n, hipMemcpyDeviceToHost, c->stream));
  if (out_ncount) HIPCHK(hipMemcpyAsync(out_ncount, w.dnc, (size_t)n * 32, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return finish(c, 0);
}

// median() of base/Utility.cpp:27-40 stored into Contig::median_abundance, an int (truncation)
static int walk_median(std::vector<int> &v) {
  if (v.empty()) return 0;
  if (v.size() == 1) return v[0];
  std::sort(v.begin(), v.end());
  const size_t t = v.size() / 2;
  return v.size() % 2 == 0 ? (int)((v[t - 1] + v[t]) / 2.0) : v[t];
}

// One maximal unitig per seed k-mer: extend forward, reverse-complement, extend forward again -- the two
// get_unitig_forward calls of processDataChunk (contig_assembly.cpp:1886-1904) in the case where no
// other unitig is met. seeds: n * k upper-case bases; seed_counts: their filter counts (Contig(kmer, count)).
extern "C" int shk_unitigs_from_seeds(shk_ctx *c, const char *seeds, const uint32_t *seed_counts, uint32_t n, uint32_t k,
                                      uint64_t abundance_min, uint32_t max_len, char *out_seq, uint32_t *out_len,
                                      int32_t *out_median, uint8_t *out_stop) {
  if (!c || (n && (!seeds || !seed_counts || !out_seq || !out_len || !out_median || !out_stop))) return SHK_ERR_ARG;
  if (k < 2 || k > SHK_WALK_MAX_K || max_len < k + 1) return SHK_ERR_ARG;
  const uint32_t max_ext = max_len - k;
  std::vector<std::string> seq(n);
  std::vector<int> med(n);
  for (uint32_t i = 0; i < n; i++) { seq[i].assign(seeds + (size_t)i * k, k); med[i] = (int)seed_counts[i]; }
  std::vector<char> cur((size_t)n * k), first((size_t)n * k), ext((size_t)n * max_ext);
  std::vector<uint32_t> cnt((size_t)n * max_ext), en(n);
  std::vector<uint8_t> st(n);
  for (int pass = 0; pass < 2; pass++) {
    for (uint32_t i = 0; i < n; i++) {
      memcpy(&first[(size_t)i * k], seq[i].data(), k);
      memcpy(&cur[(size_t)i * k], seq[i].data() + seq[i].size() - k, k);
    }
    int rc = shk_extend_forward(c, cur.data(), first.data(), n, k, abundance_min, 0, max_ext, ext.data(), cnt.data(), en.data(), st.data(), nullptr, nullptr);
    if (rc) return rc;
    for (uint32_t i = 0; i < n; i++) {
      // abundances start as (length - K + 1) copies of the contig's current median (contig_assembly.cpp:3049)
      std::vector<int> ab(seq[i].size() - k + 1, med[i]);
      uint32_t take = en[i];
      if (seq[i].size() + take > max_len) { take = max_len - (uint32_t)seq[i].size(); st[i] = SHK_STOP_BUFFER; }
      for (uint32_t j = 0; j < take; j++) ab.push_back((int)cnt[(size_t)i * max_ext + j]);
      seq[i].append(&ext[(size_t)i * max_ext], take);
      med[i] = walk_median(ab);
      out_stop[(size_t)i * 2 + pass] = st[i];
      if (pass == 0) {   // DNAString::RC
        std::string r(seq[i].rbegin(), seq[i].rend());
        for (auto &ch : r) ch = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : ch;
        seq[i].swap(r);
      }
    }
  }
  for (uint32_t i = 0; i < n; i++) {
    memcpy(out_seq + (size_t)i * max_len, seq[i].data(), seq[i].size());
    out_len[i] = (uint32_t)seq[i].size();
    out_median[i] = med[i];
  }
  return SHK_OK;
}

// ------------------------------------------------------------------ Contiger: the unitig set on the device
// (unitig_kernels.hip; the reference's find_unitigs_mt_master/worker, check_unitig, track_kmer_worker,
// build_graph_worker and the writer, src/contig_assembly.cpp:2034-2269, 935-1084, 600-629)
template <typename T> static int ug_grow(T **p, uint64_t old_n, uint64_t new_n, hipStream_t st, bool zero_new) {
  T *q = nullptr;
  if (dmalloc(&q, new_n)) return SHK_ERR_HIP;
  if (zero_new) HIPCHK(hipMemsetAsync(q, 0, new_n * sizeof(T), st));
  if (*p && old_n) HIPCHK(hipMemcpyAsync(q, *p, old_n * sizeof(T), hipMemcpyDeviceToDevice, st));
  if (*p) { HIPCHK(hipStreamSynchronize(st)); hipFree(*p); }
  *p = q;
  return SHK_OK;
}

struct shk_unitig_set {
  shk_ctx *c = nullptr;
  ShkUG G;
  uint32_t cap = 0, mcap = 0, ccap = 0, lcap = 0;     // contigs, map slots, circle slots, list entries
  uint32_t *d_list[2] = {nullptr, nullptr};
  uint32_t *d_scal = nullptr;                          // [0] ncontigs [1] next_n [2] flags [3] nactive (seeds from reads)
  unsigned long long *d_stats = nullptr;
  uint32_t *h_scal = nullptr;                          // pinned mirror (4 u32 + 4 u64)
  char *d_seeds = nullptr; uint32_t *d_counts = nullptr; uint64_t seeds_cap = 0;
  uint32_t ncontigs = 1;                               // next free id (the reference starts with contigs.resize(1))
  uint32_t k = 0, max_len = 0;
  uint64_t amin = 0;
  shk_unitig_stats st;
  shk_unitig_set() { memset(&st, 0, sizeof(st)); memset(&G, 0, sizeof(G)); }
};
extern "C" shk_unitig_set *shk_unitig_set_new(void) { return new shk_unitig_set(); }
extern "C" void shk_unitig_set_free(shk_unitig_set *u) {
  if (!u) return;
  if (u->c) {
    hipSetDevice(u->c->dev);
    hipStreamSynchronize(u->c->stream);
    ShkUG &G = u->G;
    hipFree(G.first_lo); hipFree(G.first_hi); hipFree(G.cur_lo); hipFree(G.cur_hi); hipFree(G.rc_lo); hipFree(G.rc_hi);
    hipFree(G.fh); hipFree(G.rh); hipFree(G.hmin); hipFree(G.len); hipFree(G.l1); hipFree(G.cnt0); hipFree(G.state); hipFree(G.kind);
    hipFree(G.stop); hipFree(G.mk_lo); hipFree(G.mk_hi); hipFree(G.mv); hipFree(G.ck); hipFree(G.cv);
    hipFree(u->d_list[0]); hipFree(u->d_list[1]); hipFree(u->d_scal); hipFree(u->d_stats); hipFree(u->d_seeds); hipFree(u->d_counts);
    if (u->h_scal) hipHostFree(u->h_scal);
  }
  delete u;
}

// capacities for `ncontigs_after` contig ids and `nlist` list entries
static int ug_reserve(shk_unitig_set *u, uint64_t ncontigs_after, uint64_t nlist) {
  shk_ctx *c = u->c;
  ShkUG &G = u->G;
  if (ncontigs_after + 1 > u->cap) {
    uint64_t nc = u->cap ? u->cap : 1024;
    while (nc < ncontigs_after + 1) nc *= 2;
    if (nc > 0x7FFFFFF0ull) return SHK_ERR_BATCH;
    const uint64_t o = u->cap;
    if (ug_grow(&G.first_lo, o, nc, c->stream, false) || ug_grow(&G.first_hi, o, nc, c->stream, false) || ug_grow(&G.cur_lo, o, nc, c->stream, false) ||
        ug_grow(&G.cur_hi, o, nc, c->stream, false) || ug_grow(&G.rc_lo, o, nc, c->stream, false) || ug_grow(&G.rc_hi, o, nc, c->stream, false) ||
        ug_grow(&G.fh, o, nc, c->stream, false) || ug_grow(&G.rh, o, nc, c->stream, false) || ug_grow(&G.hmin, o, nc, c->stream, false) ||
        ug_grow(&G.len, o, nc, c->stream, true) || ug_grow(&G.l1, o, nc, c->stream, true) || ug_grow(&G.cnt0, o, nc, c->stream, true) ||
        ug_grow(&G.state, o, nc, c->stream, true) || ug_grow(&G.kind, o, nc, c->stream, true) || ug_grow(&G.stop, o, nc, c->stream, true))
      return SHK_ERR_HIP;
    u->cap = (uint32_t)nc; G.cap = (uint32_t)nc;
  }
  if (nlist > u->lcap) {
    uint64_t nl = u->lcap ? u->lcap : 1024;
    while (nl < nlist) nl *= 2;
    if (ug_grow(&u->d_list[0], u->lcap, nl, c->stream, false) || ug_grow(&u->d_list[1], u->lcap, nl, c->stream, false)) return SHK_ERR_HIP;
    u->lcap = (uint32_t)nl;
  }
  // every contig owns at most two keys; the table stays at most a quarter full
  if (ncontigs_after * 8 > u->mcap) {
    uint64_t nm = u->mcap ? u->mcap : 4096;
    while (nm < ncontigs_after * 8) nm *= 2;
    if (nm > 0x80000000ull) return SHK_ERR_BATCH;
    uint64_t *ok_lo = G.mk_lo, *ok_hi = G.mk_hi; uint32_t *ov = G.mv;
    const uint32_t ocap = u->mcap;
    G.mk_lo = G.mk_hi = nullptr; G.mv = nullptr;
    if (dmalloc(&G.mk_lo, nm) || dmalloc(&G.mk_hi, nm) || dmalloc(&G.mv, nm)) return SHK_ERR_HIP;
    HIPCHK(hipMemsetAsync(G.mv, 0, nm * 4, c->stream));
    G.mmask = (uint32_t)(nm - 1); u->mcap = (uint32_t)nm;
    if (ocap) {
      hipLaunchKernelGGL(k_ug_rehash, dim3((ocap + 255) / 256), dim3(256), 0, c->stream, G, (const uint64_t *)ok_lo, (const uint64_t *)ok_hi, (const uint32_t *)ov, ocap);
      HIPCHK(hipStreamSynchronize(c->stream));
      hipFree(ok_lo); hipFree(ok_hi); hipFree(ov);
    }
  }
  if (!u->ccap) {
    const uint64_t ncs = 1 << 16;
    if (dmalloc(&G.ck, ncs) || dmalloc(&G.cv, ncs)) return SHK_ERR_HIP;
    HIPCHK(hipMemsetAsync(G.cv, 0, ncs * 4, c->stream));
    G.cmask = (uint32_t)(ncs - 1); u->ccap = (uint32_t)ncs;
  }
  return SHK_OK;
}

static int ug_bind(shk_unitig_set *u, shk_ctx *c, uint32_t k, uint64_t amin, uint32_t max_len) {
  if (u->c) {
    if (u->c != c || u->k != k || u->amin != amin || u->max_len != max_len) return SHK_ERR_ARG;   // one filter, one set of rules
    return SHK_OK;
  }
  // bound only once everything is allocated: a set whose first bind failed stays unbound (and can be bound again)
  uint32_t *d_scal = nullptr, *h_scal = nullptr;
  unsigned long long *d_stats = nullptr;
  if (dmalloc(&d_scal, 16) || dmalloc(&d_stats, 8) || hipHostMalloc((void **)&h_scal, 64, hipHostMallocDefault) != hipSuccess ||
      hipMemsetAsync(d_scal, 0, 16 * 4, c->stream) != hipSuccess || hipMemsetAsync(d_stats, 0, 8 * 8, c->stream) != hipSuccess) {
    hipFree(d_scal); hipFree(d_stats); if (h_scal) hipHostFree(h_scal);
    return SHK_ERR_HIP;
  }
  u->c = c; u->k = k; u->amin = amin; u->max_len = max_len;
  u->d_scal = d_scal; u->d_stats = d_stats; u->h_scal = h_scal;
  u->G.ncontigs = u->d_scal; u->G.next_n = u->d_scal + 1; u->G.flags = u->d_scal + 2; u->G.stats = u->d_stats;
  int rc = ug_reserve(u, 1024, 1024);
  if (rc) {           // release what the half-made set holds and unbind it
    hipStreamSynchronize(c->stream);
    ShkUG &G = u->G;
    hipFree(G.first_lo); hipFree(G.first_hi); hipFree(G.cur_lo); hipFree(G.cur_hi); hipFree(G.rc_lo); hipFree(G.rc_hi);
    hipFree(G.fh); hipFree(G.rh); hipFree(G.hmin); hipFree(G.len); hipFree(G.l1); hipFree(G.cnt0); hipFree(G.state); hipFree(G.kind);
    hipFree(G.stop); hipFree(G.mk_lo); hipFree(G.mk_hi); hipFree(G.mv); hipFree(G.ck); hipFree(G.cv);
    hipFree(u->d_list[0]); hipFree(u->d_list[1]); hipFree(u->d_scal); hipFree(u->d_stats); hipHostFree(u->h_scal);
    memset(&u->G, 0, sizeof(u->G));
    u->d_list[0] = u->d_list[1] = nullptr; u->d_scal = nullptr; u->d_stats = nullptr; u->h_scal = nullptr;
    u->cap = u->mcap = u->ccap = u->lcap = 0;
    u->c = nullptr;
  }
  return rc;
}

// rounds of k_ug_walk until no contig is open; d_list[0] holds `nactive` ids
static int ug_run(shk_unitig_set *u, uint32_t nactive, int mark) {
  shk_ctx *c = u->c;
  // Extensions per contig and launch. A launch lasts as long as its longest walk while the neighbours that the short
  // ones queued wait for the next one: with many contigs open, short launches keep the frontier moving (2 M unitigs of a
  // 100x C. elegans graph, 126 M extensions: 1.25 s at 2048 steps per launch, 0.92 at 512, 0.56 at 128, 0.49 at 64, 0.47 at
  // 32, 0.48 at 16 with 2675 launches; longer launches while fewer than 4096 / 64 contigs are open: 0.73 / 0.55). The price
  // is paid by a graph that is one long unitig: a launch and its synchronisation (~70 us) per 64 extensions (~100 us)
  uint32_t fixed = 0;
  if (const char *e = getenv("SHK_WALK_STEP")) { int v = atoi(e); if (v > 0) fixed = (uint32_t)v; }   // tests: force continuations
  int cur = 0;
  while (nactive) {
    const uint32_t step = fixed ? fixed : 64u;
    // a contig may queue up to 7 neighbours; every open contig may come back once
    int rc = ug_reserve(u, (uint64_t)u->ncontigs + 8ull * nactive + 16, 8ull * nactive + 16);
    if (rc) return rc;
    u->h_scal[0] = u->ncontigs; u->h_scal[1] = 0; u->h_scal[2] = 0;
    HIPCHK(hipMemcpyAsync(u->d_scal, u->h_scal, 12, hipMemcpyHostToDevice, c->stream));
    u->G.next = u->d_list[cur ^ 1];
    { ProfScope ps(c, KP_UG_WALK);
      hipLaunchKernelGGL(k_ug_walk, dim3((nactive + 7) / 8), dim3(64), 0, c->stream, u->G, (const uint32_t *)u->d_list[cur], nactive, c->tab[c->cur], c->q_lo,
                         c->nslots, c->cfg.hb, u->k, u->amin, mark ? 1 : 2, step, u->max_len); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(u->h_scal, u->d_scal, 12, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (u->h_scal[2]) return u->h_scal[2] & SHK_UG_E_MAP ? SHK_ERR_CORRUPT : SHK_ERR_BATCH;
    u->ncontigs = u->h_scal[0];
    nactive = u->h_scal[1];
    cur ^= 1;
    u->st.rounds++;
  }
  // (the lists may have been swapped an odd number of times: nothing depends on which one is list 0 between calls)
  return SHK_OK;
}

extern "C" int shk_unitigs_add_seeds(shk_ctx *c, shk_unitig_set *u, const char *seeds, const uint32_t *seed_counts, uint32_t n,
                                     uint32_t k, uint64_t abundance_min, uint32_t max_len, int mark_traveled) {
  if (!c || !u || (n && (!seeds || !seed_counts))) return SHK_ERR_ARG;
  if (k < 2 || k > SHK_WALK_MAX_K || max_len < k + 1) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  int rc = ug_bind(u, c, k, abundance_min, max_len);
  if (rc || n == 0) return rc;
  if ((uint64_t)n > u->seeds_cap) {
    hipFree(u->d_seeds); hipFree(u->d_counts); u->d_seeds = nullptr; u->d_counts = nullptr;
    if (dmalloc(&u->d_seeds, (uint64_t)n * k) || dmalloc(&u->d_counts, (uint64_t)n)) return SHK_ERR_HIP;
    u->seeds_cap = n;
  }
  rc = ug_reserve(u, (uint64_t)u->ncontigs + n + 16, (uint64_t)n + 16);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(u->d_seeds, seeds, (size_t)n * k, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(u->d_counts, seed_counts, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_ug_add_seeds, dim3((n + 255) / 256), dim3(256), 0, c->stream, u->G, (const char *)u->d_seeds, (const uint32_t *)u->d_counts, n, k,
                     u->ncontigs, u->d_list[0]);
  HIPCHK(hipGetLastError());
  u->ncontigs += n;
  rc = ug_run(u, n, mark_traveled);
  return finish(c, rc);
}

// Seeds straight from FASTQ chunks (processDataChunk's rule, contig_assembly.cpp:1856-1876) and their walks, without
// the seeds leaving the device: parse -> k_select_seeds (lookup that marks) -> new contigs -> rounds.
extern "C" int shk_unitigs_add_reads(shk_ctx *c, shk_unitig_set *u, const void *text, int text_on_device, uint64_t text_bytes,
                                     const uint64_t *chunk_off, const uint64_t *chunk_len, uint32_t nchunks, uint32_t k,
                                     uint64_t abundance_min, uint64_t count_min, uint64_t count_max, uint32_t max_len,
                                     uint64_t *nseeds) {
  if (!c || !u || !text || !chunk_off || !chunk_len || nchunks == 0 || nchunks > SHK_MAX_CHUNKS) return SHK_ERR_ARG;
  if (k < 2 || k > SHK_WALK_MAX_K || max_len < k + 1) return SHK_ERR_ARG;
  HIPCHK(hipSetDevice(c->dev));
  int rc = ug_bind(u, c, k, abundance_min, max_len);
  if (rc) return rc;
  const uint8_t *dtext;
  uint64_t nreads;
  rc = parse_stage(c, text, text_on_device, text_bytes, chunk_off, chunk_len, nchunks, &dtext, &nreads);
  if (rc) return finish(c, rc);
  if (nseeds) *nseeds = 0;
  if (nreads == 0) return finish(c, 0);
  if (nreads > 0x7FFFFFF0ull) return SHK_ERR_BATCH;
  if (nreads > u->seeds_cap) {
    hipFree(u->d_seeds); hipFree(u->d_counts); u->d_seeds = nullptr; u->d_counts = nullptr;
    if (dmalloc(&u->d_seeds, nreads * k) || dmalloc(&u->d_counts, nreads)) return SHK_ERR_HIP;
    u->seeds_cap = nreads;
  }
  // The reference takes its seeds read by read: a read whose middle k-mer an earlier walk has already marked gives none
  // (:1871-1873). Selecting all seeds of a batch at once would walk every unitig once per read that covers it, so the
  // reads are taken in slices that grow geometrically: the walks of one slice mark their unitigs before the next, four
  // times larger, slice looks at its reads -- the duplicates stay a small multiple of the number of unitigs.
  uint64_t total_seeds = 0, lo = 0, slice = 16384;
  if (const char *e = getenv("SHK_SEED_SLICE")) { long long v = atoll(e); if (v > 0) slice = (uint64_t)v; }
  while (lo < nreads) {
    const uint64_t hi = nreads - lo < slice ? nreads : lo + slice;
    const uint64_t m = hi - lo;
    rc = ug_reserve(u, (uint64_t)u->ncontigs + m + 16, m + 16);
    if (rc) return finish(c, rc);
    { ProfScope ps(c, KP_WALK);
      hipLaunchKernelGGL(k_select_seeds, dim3((uint32_t)((m + 255) / 256)), dim3(256), 0, c->stream, c->tab[c->cur], c->q_lo, c->nslots,
                         c->cfg.hb, dtext, c->d_rd_start, c->d_rd_end, lo, hi, k, count_min, count_max, 1, u->d_seeds, u->d_counts); }
    u->h_scal[0] = u->ncontigs; u->h_scal[1] = 0; u->h_scal[2] = 0; u->h_scal[3] = 0;
    HIPCHK(hipMemcpyAsync(u->d_scal, u->h_scal, 16, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_ug_seeds_from_reads, dim3((uint32_t)((m + 255) / 256)), dim3(256), 0, c->stream, u->G, (const char *)u->d_seeds,
                       (const uint32_t *)u->d_counts, lo, hi, k, u->d_list[0], u->d_scal + 3);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(u->h_scal, u->d_scal, 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (u->h_scal[2]) return finish(c, SHK_ERR_BATCH);
    u->ncontigs = u->h_scal[0];
    const uint32_t n = u->h_scal[3];
    total_seeds += n;
    rc = ug_run(u, n, 1);
    if (rc) return finish(c, rc);
    lo = hi;
    slice *= 4;
  }
  if (nseeds) *nseeds = total_seeds;
  return finish(c, SHK_OK);
}

extern "C" int shk_unitig_set_write(shk_unitig_set *u, uint32_t k, const char *out_path, shk_unitig_stats *stats) {
  if (!u || !out_path) return SHK_ERR_ARG;
  FILE *fo = fopen(out_path, "w");
  if (!fo) return SHK_ERR_IO;
  if (!u->c) {          // nothing was ever added
    fclose(fo);
    if (stats) *stats = u->st;
    return SHK_OK;
  }
  if (k != u->k) { fclose(fo); return SHK_ERR_ARG; }
  shk_ctx *c = u->c;
  HIPCHK(hipSetDevice(c->dev));
  const uint32_t n = u->ncontigs;
  uint32_t *d_keep = nullptr, *d_lens = nullptr, *d_ulen = nullptr, *d_ul1 = nullptr, *d_cnt = nullptr;
  uint64_t *d_newid = nullptr, *d_off = nullptr, *d_uoff = nullptr, *d_sums = nullptr;
  char *d_bases = nullptr;
  int32_t *d_med = nullptr, *d_links = nullptr;
  uint64_t *m_lo = nullptr, *m_hi = nullptr; uint32_t *m_v = nullptr;
  int rc = SHK_OK;
  std::vector<char> bases;
  std::vector<uint64_t> uoff;
  std::vector<uint32_t> ulen;
  std::vector<int32_t> med, links;
  uint64_t nunits = 0, total = 0;
  do {
    // the scans run over all contig ids (seeds, queued neighbours, duplicates): their block sums get scratch of their own,
    // sized from n (the context's is sized for its key batches -- a graph of more than 16.7 M ids used to be refused here,
    // after all the walks)
    if (dmalloc(&d_sums, (uint64_t)n / SHK_SCAN_TILE + 8)) { rc = SHK_ERR_HIP; break; }
    if (dmalloc(&d_keep, (uint64_t)n + 1) || dmalloc(&d_lens, (uint64_t)n + 1) || dmalloc(&d_newid, (uint64_t)n + 2) || dmalloc(&d_off, (uint64_t)n + 2)) { rc = SHK_ERR_HIP; break; }
    hipLaunchKernelGGL(k_ug_check, dim3((n + 255) / 256), dim3(256), 0, c->stream, u->G, n, d_keep, d_lens);
    if (getenv("SHK_UG_DEBUG")) {   // diagnostics: contigs by kind, state and last stop reason
      std::vector<uint8_t> hs(n), hk(n), hp(n);
      std::vector<uint32_t> hkeep(n);
      hipMemcpy(hs.data(), u->G.state, n, hipMemcpyDeviceToHost); hipMemcpy(hk.data(), u->G.kind, n, hipMemcpyDeviceToHost);
      hipMemcpy(hp.data(), u->G.stop, n, hipMemcpyDeviceToHost); hipMemcpy(hkeep.data(), d_keep, (size_t)n * 4, hipMemcpyDeviceToHost);
      unsigned long long h[2][4][8] = {{{0}}}, kept[2] = {0, 0};
      for (uint32_t i = 1; i < n; i++) { h[hk[i] & 1][hs[i] & 3][hp[i] & 7]++; kept[hk[i] & 1] += hkeep[i]; }
      for (int kd = 0; kd < 2; kd++) for (int stt = 0; stt < 4; stt++) for (int sp = 0; sp < 8; sp++)
        if (h[kd][stt][sp]) fprintf(stderr, "SHK_UG_DEBUG %s state %d stop %d: %llu\n", kd ? "seed" : "cand", stt, sp, h[kd][stt][sp]);
      fprintf(stderr, "SHK_UG_DEBUG kept seeds %llu candidates %llu\n", kept[1], kept[0]);
      std::vector<uint64_t> hh(n);
      std::vector<uint32_t> hl(n);
      hipMemcpy(hh.data(), u->G.hmin, (size_t)n * 8, hipMemcpyDeviceToHost); hipMemcpy(hl.data(), u->G.len, (size_t)n * 4, hipMemcpyDeviceToHost);
      for (uint32_t i = 1; i < n; i++)
        if ((hp[i] & 15) == SHK_STOP_CIRCLE) fprintf(stderr, "SHK_UG_DEBUG circle id %u state %d keep %u len %u hmin %016llx\n", i, hs[i], hkeep[i], hl[i], (unsigned long long)hh[i]);
    }
    if (run_scan<uint32_t>(c, d_keep, n, nullptr, d_newid, d_sums) || run_scan<uint32_t>(c, d_lens, n, nullptr, d_off, d_sums)) { rc = SHK_ERR_HIP; break; }
    if (hipMemcpyAsync(c->h_pinned + 45, d_newid + n, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipMemcpyAsync(c->h_pinned + 46, d_off + n, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) { rc = SHK_ERR_HIP; break; }
    nunits = c->h_pinned[45]; total = c->h_pinned[46];
    if (nunits == 0) break;
    if (dmalloc(&d_bases, total + 16) || dmalloc(&d_cnt, total + 16) || dmalloc(&d_uoff, nunits + 1) || dmalloc(&d_ulen, nunits + 1) ||
        dmalloc(&d_ul1, nunits + 1) || dmalloc(&d_med, nunits + 1) || dmalloc(&d_links, nunits * 8 + 8)) { rc = SHK_ERR_HIP; break; }
    { ProfScope ps(c, KP_UG_FINISH);
      hipLaunchKernelGGL(k_ug_emit, dim3((n + 63) / 64), dim3(64), 0, c->stream, u->G, n, (const uint32_t *)d_keep, (const uint64_t *)d_newid, (const uint64_t *)d_off,
                         c->tab[c->cur], c->q_lo, c->nslots, c->cfg.hb, k, u->amin, d_bases, d_cnt, d_uoff, d_ulen, d_ul1); }
    hipLaunchKernelGGL(k_ug_median, dim3((uint32_t)nunits), dim3(SHK_WAVE), 0, c->stream, (uint32_t)nunits, (const uint64_t *)d_uoff, (const uint32_t *)d_ulen,
                       (const uint32_t *)d_ul1, (const uint32_t *)d_cnt, k, d_med);
    // the graph pass's own map: first k-mer -> +number, RC(last k-mer) -> -number
    uint64_t nm = 4096;
    while (nm < nunits * 8) nm *= 2;
    if (dmalloc(&m_lo, nm) || dmalloc(&m_hi, nm) || dmalloc(&m_v, nm)) { rc = SHK_ERR_HIP; break; }
    if (hipMemsetAsync(m_v, 0, nm * 4, c->stream) != hipSuccess) { rc = SHK_ERR_HIP; break; }
    ShkUG G2 = u->G;
    G2.mk_lo = m_lo; G2.mk_hi = m_hi; G2.mv = m_v; G2.mmask = (uint32_t)(nm - 1);
    hipLaunchKernelGGL(k_ug_map2, dim3((n + 255) / 256), dim3(256), 0, c->stream, G2, n, (const uint32_t *)d_keep, (const uint64_t *)d_newid);
    hipLaunchKernelGGL(k_ug_links, dim3((n + 255) / 256), dim3(256), 0, c->stream, G2, n, (const uint32_t *)d_keep, (const uint64_t *)d_newid, k, d_links);
    if (hipGetLastError() != hipSuccess) { rc = SHK_ERR_HIP; break; }
    bases.resize(total); uoff.resize(nunits); ulen.resize(nunits); med.resize(nunits); links.resize(nunits * 8);
    if (hipMemcpyAsync(bases.data(), d_bases, total, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipMemcpyAsync(uoff.data(), d_uoff, nunits * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipMemcpyAsync(ulen.data(), d_ulen, nunits * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipMemcpyAsync(med.data(), d_med, nunits * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipMemcpyAsync(links.data(), d_links, nunits * 32, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipMemcpyAsync(u->h_scal + 4, u->d_stats, 32, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) { rc = SHK_ERR_HIP; break; }
  } while (0);
  hipStreamSynchronize(c->stream);
  hipFree(d_keep); hipFree(d_lens); hipFree(d_newid); hipFree(d_off); hipFree(d_bases); hipFree(d_cnt); hipFree(d_uoff); hipFree(d_ulen);
  hipFree(d_ul1); hipFree(d_med); hipFree(d_links); hipFree(m_lo); hipFree(m_hi); hipFree(m_v); hipFree(d_sums);
  if (rc) { fclose(fo); return finish(c, rc); }
  // the records as the reference writes them (:606-626): ids 0-based in final numbering, successors then predecessors.
  // Formatted in slices by a few host threads (two million records are ~6 M numbers to print), written in order
  {
    const auto put_num = [](std::string &o, long long v) {
      char t[24]; int n = 0;
      unsigned long long a = v < 0 ? 0ULL - (unsigned long long)v : (unsigned long long)v;
      do { t[n++] = (char)('0' + a % 10); a /= 10; } while (a);
      if (v < 0) o.push_back('-');
      while (n) o.push_back(t[--n]);
    };
    const auto format = [&](uint64_t a, uint64_t b, std::string &o) {
      o.clear();
      for (uint64_t i = a; i < b; i++) {
        const long long len = ulen[i];
        o.push_back('>'); put_num(o, (long long)i);
        o.append(" LN:i:"); put_num(o, len);
        o.append(" KC:i:"); put_num(o, (long long)med[i] * (len - (long long)k + 1));
        o.append(" km:f:"); put_num(o, med[i]);
        for (int x = 0; x < 8; x++) {
          const int32_t v = links[i * 8 + x];
          if (!v) continue;
          o.append(x < 4 ? " L:+:" : " L:-:"); put_num(o, (v > 0 ? v : -v) - 1);
          o.append(v > 0 ? ":+" : ":-");
        }
        o.push_back('\n');
        o.append(bases.data() + uoff[i], (size_t)len);
        o.push_back('\n');
      }
    };
    const uint64_t slice = 1u << 15;
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : nt > 8 ? 8 : nt;
    std::vector<std::string> outs(nt);
    for (uint64_t a0 = 0; a0 < nunits; a0 += slice * nt) {
      std::vector<std::thread> th;
      unsigned used = 0;
      for (unsigned t = 0; t < nt && a0 + t * slice < nunits; t++, used++) {
        const uint64_t a = a0 + t * slice, b = a + slice < nunits ? a + slice : nunits;
        if (nt == 1) format(a, b, outs[0]);
        else th.emplace_back(format, a, b, std::ref(outs[t]));
      }
      for (auto &x : th) x.join();
      for (unsigned t = 0; t < used; t++)
        if (fwrite(outs[t].data(), 1, outs[t].size(), fo) != outs[t].size()) rc = SHK_ERR_IO;
    }
  }
  if (fclose(fo) != 0) rc = SHK_ERR_IO;
  if (rc) return finish(c, rc);
  const unsigned long long *ds = reinterpret_cast<const unsigned long long *>(u->h_scal + 4);
  u->st.unitigs = nunits; u->st.total_len = total;
  if (nunits) { u->st.extensions = ds[0]; u->st.duplicates = ds[1]; u->st.truncated = ds[2]; }
  if (stats) *stats = u->st;
  return finish(c, SHK_OK);
}

extern "C" int shk_find_unitigs(shk_ctx *c, const char *seeds, const uint32_t *seed_counts, uint32_t n, uint32_t k,
                                uint64_t abundance_min, uint32_t max_len, const char *out_path, shk_unitig_stats *stats) {
  if (!out_path) return SHK_ERR_ARG;
  shk_unitig_set *u = shk_unitig_set_new();
  int rc = shk_unitigs_add_seeds(c, u, seeds, seed_counts, n, k, abundance_min, max_len, 0);
  if (!rc) rc = shk_unitig_set_write(u, k, out_path, stats);
  shk_unitig_set_free(u);
  return rc;
}

extern "C" int shk_profile_enable(shk_ctx *c, int on) { if (!c) return SHK_ERR_ARG; c->prof_on = on; return SHK_OK; }
extern "C" int shk_profile_reset(shk_ctx *c) {
  if (!c) return SHK_ERR_ARG;
  for (int i = 0; i < KP_N; i++) { c->prof_ms[i] = 0; c->prof_n[i] = 0; }
  return SHK_OK;
}
extern "C" int shk_profile_get(shk_ctx *c, shk_kernel_time *out, int cap) {
  if (!c || !out) return SHK_ERR_ARG;
  int n = 0;
  for (int i = 0; i < KP_N && n < cap; i++) {
    out[n].name = kp_names[i]; out[n].launches = c->prof_n[i]; out[n].ms = c->prof_ms[i]; n++;
  }
  return n;
}
