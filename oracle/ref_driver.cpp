// TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// oracle/_ref/libshk_ref.so = the REAL reference counting-quotient-filter
// (/root/reference/cqf/gqf.c, compiled from where it lies, unmodified) and the
// REAL reference ntHash (/root/reference/base/nthash.hpp), plus this driver.
//
// The reference's own driver layer (cqf/CQF_mt.h, src/CQF-deNoise.cpp) cannot be
// compiled in this image (it needs boost headers, which are absent, and writing
// stand-ins for them is not allowed), so the few control-flow functions of that
// layer are RESTATED here on top of the real qf_* entry points:
//   ref_reads_to_kmers        <- cqf/CQF_mt.h:610-731  (reads_to_kmers)
//   rd_skip_next_eol          <- cqf/CQF_mt.h:573-585  (skip_next_eol)
//   rd_fastq_read_parts       <- cqf/CQF_mt.h:735-816  (fastq_read_parts)
//   ref_clean_with_lock       <- cqf/CQF_mt.h:999-1039 (qf_clean_singleton_with_lock)
//   ref_denoise_round_t1      <- cqf/CQF_mt.h:884-901  (DeNoise mode, one thread)
//   ref_build_t1              <- cqf/CQF_mt.h:821-931, 959-995 (t = 1 schedule)
// Everything below those (insert, counters, sweep, rank/select, serialize,
// lookups, traveled bits) is the reference's compiled code.
//
// Built only by oracle/Makefile (target `ref`), only when /root/reference exists.

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <deque>
#include <atomic>
#include <chrono>
#include <thread>

#include "cqf/gqf.h"        // -I/root/reference
#include "base/nthash.hpp"  // -I/root/reference

extern "C" {

// ---------------------------------------------------------------- ntHash
void ref_nthash(const char *seq, unsigned k, uint64_t *fh, uint64_t *rh) {
  NTPC64(seq, k, *fh, *rh);
}
void ref_nthash_roll(unsigned char out, unsigned char in, unsigned k,
                     uint64_t *fh, uint64_t *rh) {
  NTPC64(out, in, k, *fh, *rh);
}

// ---------------------------------------------------------------- QF handle
struct RefQF {
  QF qf;
  uint64_t nelts;      // runtime->nelts      (CQF_mt.h:278)
  uint64_t ndistinct;  // runtime->ndistinct  (CQF_mt.h:277)
};

// The reference's slot accessors load and store 8 bytes at a slot's address whatever the slot width (gqf.c:542-574), so
// with the last slots of the overflow tail in use they touch up to 7 bytes behind the calloc(size) of qf_init /
// qf_deserialize (gqf.c:2236, 2416) -- a heap overrun in the reference itself (tests/test_oracle.py shows it under
// AddressSanitizer). The checker must survive any scenario, so the driver re-allocates the table with two zeroed guard
// blocks behind it; metadata->size and everything the reference reads or writes inside the table are unchanged.
// REF_NO_GUARD=1 keeps the reference's own allocation (used by that test only).
static void guard_table(QF *qf) {
  if (getenv("REF_NO_GUARD")) return;
  uint64_t size = qf->metadata->size;
  void *p = realloc((void *)qf->blocks, size + 2 * 89 + 16);
  if (!p) abort();
  memset((char *)p + size, 0, 2 * 89 + 16);
  qf->blocks = (qfblock *)p;
}

RefQF *ref_qf_new(uint64_t qb, uint64_t hb, uint32_t seed) {
  RefQF *h = new RefQF();
  // CQF_mt ctor, cqf/CQF_mt.h:427-445
  uint64_t nslots = 1ULL << qb;
  if (qb == hb) qf_init(&h->qf, nslots, hb + 8, 0, true, "", seed);
  else          qf_init(&h->qf, nslots, hb, 0, true, "", seed);
  guard_table(&h->qf);
  h->nelts = h->ndistinct = 0;
  return h;
}
void ref_qf_free(RefQF *h) { qf_destroy(&h->qf, true); delete h; }

// returns 1 when the key was new (isNew), 0 otherwise
int ref_qf_insert(RefQF *h, uint64_t key, uint64_t count) {
  bool isNew = false;
  qf_insert_advance(&h->qf, key, 0, count, true, true, isNew);
  if (isNew) h->ndistinct++;
  h->nelts += count;
  return isNew ? 1 : 0;
}
uint64_t ref_qf_count(RefQF *h, uint64_t key) { return qf_count_key_value(&h->qf, key, 0); }
int ref_qf_count_set_traveled(RefQF *h, uint64_t key, uint64_t *count) {
  return qf_count_key_value_set_traveled(&h->qf, key, 0, count) ? 1 : 0;
}
int ref_qf_count_is_traveled(RefQF *h, uint64_t key, uint64_t *count) {
  return qf_count_key_value_is_traveled(&h->qf, key, 0, count) ? 1 : 0;
}
const void *ref_qf_blocks(RefQF *h) { return (const void *)h->qf.blocks; }
uint64_t ref_qf_size(RefQF *h) { return h->qf.metadata->size; }
const void *ref_qf_metadata(RefQF *h) { return (const void *)h->qf.metadata; }
uint64_t ref_sizeof_metadata() { return sizeof(qfmetadata); }
uint64_t ref_qf_nelts(RefQF *h) { return h->nelts; }
uint64_t ref_qf_ndistinct(RefQF *h) { return h->ndistinct; }
int ref_qf_check_offset(RefQF *h) { return check_offset(&h->qf) ? 1 : 0; }
uint64_t ref_qf_popcnt_occupieds(RefQF *h) {
  // popcnt_occupieds() in the reference reads one block past the end
  // (gqf.c:3051 `x <= nblocks`), so count here over the real blocks.
  uint64_t n = 0;
  const uint8_t *b = (const uint8_t *)h->qf.blocks;
  uint64_t bsz = h->qf.metadata->size / h->qf.metadata->nblocks;
  for (uint64_t i = 0; i < h->qf.metadata->nblocks; i++) {
    uint64_t w; memcpy(&w, b + i * bsz + 1, 8); n += __builtin_popcountll(w);
  }
  return n;
}
uint64_t ref_qf_popcnt_runends(RefQF *h) { return popcnt_runends(&h->qf); }
uint64_t ref_find_first_empty_slot(RefQF *h, uint64_t from) { return find_first_empty_slot(&h->qf, from); }
uint64_t ref_find_first_nonempty_slot(RefQF *h, uint64_t from) { return find_first_nonempty_slot(&h->qf, from); }

// (key,count) dump through the reference iterator (gqf.c:2474-2601)
uint64_t ref_qf_dump(RefQF *h, uint64_t *keys, uint64_t *counts, uint64_t cap) {
  QFi it; uint64_t n = 0;
  if (!qf_iterator(&h->qf, &it, 0)) return 0;
  do {
    uint64_t k = 0, v = 0, c = 0;
    if (qfi_get(&it, &k, &v, &c)) break;
    if (n < cap) { keys[n] = k; counts[n] = c; }
    n++;
  } while (!qfi_next(&it));
  return n;
}

// the reference's own merges (gqf.c:2614-2655, 2660-2704): c := a + b, r := sum of arr[] (inputs must be non-empty:
// qf_merge reads keya/keyb of an empty iterator uninitialised). The runtime counters of the result are what a dump
// of it gives (gqf.c maintains none: every modify_metadata call is commented out).
static void ref_recount(RefQF *h) {
  QFi it; uint64_t n = 0, tot = 0;
  if (qf_iterator(&h->qf, &it, 0)) {
    do {
      uint64_t k = 0, v = 0, c = 0;
      if (qfi_get(&it, &k, &v, &c)) break;
      n++; tot += c;
    } while (!qfi_next(&it));
  }
  h->ndistinct = n; h->nelts = tot;
}
void ref_qf_merge(RefQF *a, RefQF *b, RefQF *c) { qf_merge(&a->qf, &b->qf, &c->qf); ref_recount(c); }
void ref_qf_multi_merge(RefQF **arr, int n, RefQF *r) {
  std::vector<QF *> q(n);
  for (int i = 0; i < n; i++) q[i] = &arr[i]->qf;
  qf_multi_merge(q.data(), n, &r->qf);
  ref_recount(r);
}

// encode_counter (gqf.c:1225-1255): returns the number of slots, written to out[]
int ref_encode_counter(RefQF *h, uint64_t remainder, uint64_t counter, uint64_t *out) {
  uint64_t buf[67];
  uint64_t *p = encode_counter(&h->qf, remainder, counter, &buf[67]);
  int n = (int)(&buf[67] - p);
  for (int i = 0; i < n; i++) out[i] = p[i];
  return n;
}

void ref_qf_serialize(RefQF *h, const char *path) {
  // CQF_mt.h:986-987 patches the counters in before save
  h->qf.metadata->nelts = h->nelts;
  h->qf.metadata->ndistinct_elts = h->ndistinct;
  qf_serialize(&h->qf, path);
}
RefQF *ref_qf_load(const char *path) {
  RefQF *h = new RefQF();
  qf_deserialize(&h->qf, path);
  guard_table(&h->qf);
  h->nelts = h->qf.metadata->nelts;
  h->ndistinct = h->qf.metadata->ndistinct_elts;
  return h;
}

// ---------------------------------------------------------------- deNoise (t = 1)
// restates qf_clean_singleton_with_lock, cqf/CQF_mt.h:999-1039 (locks are no-ops
// with one thread; the control flow, including which clusters get visited, is kept)
static void ref_clean_with_lock(RefQF *h, uint64_t start_bucket_id, uint64_t end_bucket_id) {
  const QF *qf = &h->qf;
  uint64_t start, end, start_block_idx, end_block_idx, removed_elts;
  start_block_idx = start_bucket_id / SLOTS_PER_BLOCK;
  end_block_idx = end_bucket_id / SLOTS_PER_BLOCK;
  removed_elts = 0;
  if (start_block_idx == end_block_idx) {
    qf_clean_singleton_discrete(qf, start_bucket_id, end_bucket_id, &removed_elts);
  } else {
    start = start_bucket_id;
    end = find_first_empty_slot(qf, start + 1) - 1;
    while ((end / SLOTS_PER_BLOCK) == start_block_idx) {
      qf_clean_singleton(qf, start, end, &removed_elts);
      start = find_first_nonempty_slot(qf, end + 1);
      end = find_first_empty_slot(qf, start + 1) - 1;
    }
    if (start / SLOTS_PER_BLOCK == start_block_idx) {
      qf_clean_singleton_with_lock_atStart(qf, start, end, &removed_elts);
      start = find_first_nonempty_slot(qf, end + 1);
      end = find_first_empty_slot(qf, start + 1) - 1;
    }
    while (start < end_bucket_id) {
      if (end / SLOTS_PER_BLOCK == end_block_idx) {
        qf_clean_singleton_discrete(qf, start, end_bucket_id, &removed_elts);
        break;
      } else {
        qf_clean_singleton(qf, start, end, &removed_elts);
      }
      start = find_first_nonempty_slot(qf, end + 1);
      end = find_first_empty_slot(qf, start + 1) - 1;
    }
  }
  h->nelts -= removed_elts;
  h->ndistinct -= removed_elts;
}

// One full DeNoise phase as a single thread runs it, cqf/CQF_mt.h:866, 884-901.
// min_len = runtime->min_DeNoise_len (NUM_SLOTS_TO_LOCK<<4 = 2^20 in the reference).
uint64_t ref_denoise_round_t1(RefQF *h, uint64_t min_len) {
  QF *qf = &h->qf;
  uint64_t before = h->ndistinct;
  uint64_t max_index = qf->metadata->nslots;                    // CQF_mt.h:965
  uint64_t current_index = find_first_nonempty_slot(qf, 0);     // CQF_mt.h:866
  while (current_index < max_index) {
    uint64_t start = current_index, end;
    end = (start + min_len) > (qf->metadata->nslots) ? (qf->metadata->nslots) : (start + min_len);
    end = find_first_empty_slot(qf, end) - 1;
    current_index = find_first_nonempty_slot(qf, end + 1);
    ref_clean_with_lock(h, start, end);
  }
  return before - h->ndistinct;
}

// ---------------------------------------------------------------- reads_to_kmers
// restates cqf/CQF_mt.h:610-731 for one thread (the try-lock never fails with one
// thread, so the local-QF fallback at :639-645 is unreachable).
// spin = false is the reference's own call (one thread never finds a region lock busy); the timing entry below
// runs several threads and lets them wait for the lock instead of diverting to a thread-local filter
static void rd_reads_to_kmers(RefQF *h, const char *chunk, uint64_t size, unsigned ksize, bool spin, uint64_t *new_out,
                              uint64_t *total_out) {
  uint64_t new_elts = 0, total_elts = 0;
  bool isNew;
  QF *main_qf = &h->qf;
  const char *fs = chunk;
  const char *fe = chunk;
  const char *end = fs + size;
  while (fs && fs != end) {
    fs = static_cast<const char *>(memchr(fs, '\n', end - fs));
    fs++;
    fe = static_cast<const char *>(memchr(fs, '\n', end - fs));
    std::string read(fs, fe - fs);
    uint64_t hash, hash_RC;
  start_read:
    if (read.length() < ksize) goto next_read;
    {
      NTPC64(read.c_str(), ksize, hash, hash_RC);
      {
        uint64_t hv = hash < hash_RC ? hash : hash_RC;
        qf_insert_advance(main_qf, (uint64_t)(hv % main_qf->metadata->range), 0, 1, true, spin, isNew);
        if (isNew) new_elts++; else total_elts++;
      }
      for (uint32_t i = ksize; i < read.length(); i++) {
        if (read[i] == 'N') {
          read = read.substr(i + 1, read.length());
          goto start_read;
        }
        NTPC64(read[i - ksize], read[i], ksize, hash, hash_RC);
        uint64_t hv = hash < hash_RC ? hash : hash_RC;
        qf_insert_advance(main_qf, (uint64_t)(hv % main_qf->metadata->range), 0, 1, true, spin, isNew);
        if (isNew) new_elts++; else total_elts++;
      }
    }
  next_read:
    fs = ++fe;
    fs = static_cast<const char *>(memchr(fs, '\n', end - fs));
    fs++;
    fs = static_cast<const char *>(memchr(fs, '\n', end - fs));
    fs++;
  }
  *new_out = new_elts;
  *total_out = new_elts + total_elts;
}
void ref_reads_to_kmers(RefQF *h, const char *chunk, uint64_t size, unsigned ksize) {
  uint64_t nw = 0, tot = 0;
  rd_reads_to_kmers(h, chunk, size, ksize, false, &nw, &tot);
  h->ndistinct += nw;
  h->nelts += tot;
}

// CPU baseline for bench.py: `nthreads` threads take chunks from a shared cursor and insert into the
// one filter under the reference's region locks (what CQF_mt does with -t N, CQF_mt.h:821-831, minus
// the thread-local overflow filter). Stops taking chunks after `budget_s` seconds. Returns seconds.
double ref_time_chunks_mt(RefQF *h, const char *text, const uint64_t *offs, const uint64_t *lens, uint32_t nchunks,
                          unsigned ksize, unsigned nthreads, double budget_s, uint64_t *kmers_out, uint32_t *chunks_out) {
  std::atomic<uint32_t> next(0);
  std::atomic<uint64_t> kmers(0), newk(0);
  auto t0 = std::chrono::steady_clock::now();
  auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nthreads; t++)
    th.emplace_back([&]() {
      for (;;) {
        if (elapsed() > budget_s) break;
        uint32_t c = next.fetch_add(1);
        if (c >= nchunks) break;
        uint64_t nw = 0, tot = 0;
        rd_reads_to_kmers(h, text + offs[c], lens[c], ksize, true, &nw, &tot);
        kmers += tot; newk += nw;
      }
    });
  for (auto &x : th) x.join();
  double dt = elapsed();
  h->nelts += kmers; h->ndistinct += newk;
  *kmers_out = kmers;
  uint32_t used = next.load();
  *chunks_out = used < nchunks ? used : nchunks;
  return dt;
}

// ---------------------------------------------------------------- chunker
struct RdFile {
  FILE *in;
  std::vector<char> part_buffer;  // carry-over, file_pointer::part_buffer
  uint64_t part_filled;
};

// cqf/CQF_mt.h:573-585
static bool rd_skip_next_eol(char *part, int64_t &pos, int64_t max_pos) {
  int64_t i;
  for (i = pos; i < max_pos - 2; ++i)
    if ((part[i] == '\n' || part[i] == '\r') && !(part[i + 1] == '\n' || part[i + 1] == '\r'))
      break;
  if (i >= max_pos - 2) return false;
  pos = i + 1;
  return true;
}

// cqf/CQF_mt.h:735-816, plain-text mode. part_size/overhead are the reference's
// 1<<23 and 65535 unless a test shrinks them. Returns false at end of file.
static bool rd_fastq_read_parts(RdFile *fp, uint64_t part_size, uint32_t OVERHEAD_SIZE,
                                std::vector<char> &out) {
  std::vector<char> buf(part_size + OVERHEAD_SIZE);
  char *part = buf.data();
  memcpy(part, fp->part_buffer.data(), fp->part_filled);
  if (feof(fp->in)) return false;
  uint64_t readed = fread(part + fp->part_filled, 1, part_size, fp->in);
  int64_t total_filled = fp->part_filled + readed;
  int64_t i;
  uint64_t size;
  if (fp->part_filled >= OVERHEAD_SIZE) {
    fprintf(stderr, "Error: Wrong input file!\n");
    exit(EXIT_FAILURE);
  }
  if (feof(fp->in)) {
    out.assign(part, part + total_filled);
    return true;
  }
  {
    int64_t line_start[9];
    int32_t j;
    i = total_filled - OVERHEAD_SIZE / 2;
    for (j = 0; j < 9; ++j) {
      if (!rd_skip_next_eol(part, i, total_filled)) break;
      line_start[j] = i;
    }
    if (j < 9) size = 0;
    else {
      int k;
      for (k = 0; k < 4; ++k) {
        if (part[line_start[k] + 0] == '@' && part[line_start[k + 2] + 0] == '+') {
          if (part[line_start[k + 2] + 1] == '\n' || part[line_start[k + 2] + 1] == '\r') break;
          if (line_start[k + 1] - line_start[k] == line_start[k + 3] - line_start[k + 2] &&
              memcmp(part + line_start[k] + 1, part + line_start[k + 2] + 1,
                     line_start[k + 3] - line_start[k + 2] - 1) == 0)
            break;
        }
      }
      if (k == 4) size = 0;
      else size = line_start[k];
    }
  }
  memcpy(fp->part_buffer.data(), part + size, total_filled - size);
  fp->part_filled = total_filled - size;
  out.assign(part, part + size);
  return true;
}

// Chunk sizes only (fixture for the chunker): returns the number of chunks.
uint64_t ref_chunk_sizes(const char *path, uint64_t part_size, uint32_t overhead,
                         uint64_t *sizes, uint64_t cap) {
  RdFile f; f.in = fopen(path, "rb"); if (!f.in) return 0;
  f.part_buffer.resize(overhead + part_size); f.part_filled = 0;
  uint64_t n = 0; std::vector<char> c;
  while (rd_fastq_read_parts(&f, part_size, overhead, c)) { if (n < cap) sizes[n] = c.size(); n++; }
  fclose(f.in);
  return n;
}

// ---------------------------------------------------------------- t = 1 build
// restates the ExtractKmer/DeNoise/Idle machine of cqf/CQF_mt.h:821-931 for
// workerNum == 1, and build_KmerSpectrum :959-995.
// stats[0]=rounds fired, stats[1]=total removed, stats[2]=chunks processed
void ref_build_t1(RefQF *h, const char **files, int nfiles, unsigned ksize,
                  uint64_t ndistinct_for_denoise, uint32_t num_deNoise, int end_deNoise,
                  uint64_t part_size, uint32_t overhead, uint64_t min_denoise_len,
                  uint64_t *stats) {
  std::deque<RdFile *> ip_files;  // boost::lockfree::queue, FIFO (CQF_mt.h:336)
  int num_files = 0;
  for (int i = 0; i < nfiles; i++) {
    FILE *in = fopen(files[i], "rb");
    if (!in) continue;  // getFileReader failure: file skipped (CQF_mt.h:351-360)
    RdFile *f = new RdFile(); f->in = in; f->part_buffer.resize(overhead + part_size); f->part_filled = 0;
    ip_files.push_back(f); num_files++;
  }
  stats[0] = stats[1] = stats[2] = 0;
  enum { ExtractKmer, DeNoise, Idle } mode = ExtractKmer;
  std::vector<char> chunk;
  while (true) {
    if (mode == ExtractKmer) {
      while (num_files) {
        RdFile *fp = ip_files.front(); ip_files.pop_front();
        if (rd_fastq_read_parts(fp, part_size, overhead, chunk)) {
          ip_files.push_back(fp);
          ref_reads_to_kmers(h, chunk.data(), chunk.size(), ksize);
          stats[2]++;
          if (num_deNoise && h->ndistinct >= ndistinct_for_denoise) break;  // :837
        } else {
          fclose(fp->in); delete fp; num_files--;
        }
      }
      if ((num_deNoise && h->ndistinct >= ndistinct_for_denoise) || (end_deNoise && !num_files)) {  // :860
        if (num_deNoise) num_deNoise--;
        mode = DeNoise;
      } else if (!num_files) {
        mode = Idle;
      }
    } else if (mode == DeNoise) {
      stats[1] += ref_denoise_round_t1(h, min_denoise_len);
      stats[0]++;
      mode = num_files ? ExtractKmer : Idle;  // :904-909
    } else {
      break;
    }
  }
}

}  // extern "C"
