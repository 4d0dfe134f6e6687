// C entry points over the host-only pieces (chunker, sizing) so that the CPU test-suite can
// check them against the oracle without a GPU (libshkhost.so, built with g++).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "fastq_chunker.hpp"
#include "sizing.hpp"
#include "stitch.hpp"

extern "C" {
// sizes of the parts getDataChunk hands out for a list of files (round robin), 0 = plain, 1 = gzip
uint64_t shkh_chunk_sizes(const char **paths, int nfiles, int mode, uint64_t part_size, uint32_t overhead, uint64_t *sizes,
                          uint64_t cap) {
  std::vector<std::string> f(paths, paths + nfiles);
  shk::seqFile_batch b(f, shk::FASTQ, mode == 1 ? shk::GZIP : mode == 2 ? shk::BZIP2 : shk::TEXT, part_size, overhead);
  shk::chunk c;
  uint64_t n = 0;
  while (b.getDataChunk(c)) { if (n < cap) sizes[n] = c.get_size(); n++; free(c.get_reads()); }
  return n;
}
// the same walk, reporting how it ended: 0 = every file read to its end, 1 = stopped on an error (not FASTQ, a corrupt or
// truncated compressed stream, a part size below the overhead); *nparts = parts handed out before that
int shkh_chunk_status(const char **paths, int nfiles, int mode, uint64_t part_size, uint32_t overhead, uint64_t *nparts) {
  std::vector<std::string> f(paths, paths + nfiles);
  shk::seqFile_batch b(f, shk::FASTQ, mode == 1 ? shk::GZIP : mode == 2 ? shk::BZIP2 : shk::TEXT, part_size, overhead);
  shk::chunk c;
  uint64_t n = 0;
  while (b.getDataChunk(c)) { n++; free(c.get_reads()); }
  if (nparts) *nparts = n;
  return b.bad() ? 1 : 0;
}
// shard tables -> the single table (see stitch.hpp)
int shkh_stitch(const uint8_t *const *shards, const uint64_t *shard_blocks, uint32_t nshards, uint32_t qb, uint8_t *out,
                uint64_t out_bytes) {
  return shk::stitch_shards(shards, shard_blocks, nshards, qb, out, out_bytes);
}
// the stitch, one rank at a time (stitch.hpp). shkh_shard_layout returns the number of spilled slots (or < 0); the spill
// itself is fetched with shkh_shard_spill afterwards (kept in a thread-local buffer in between)
int shkh_shard_summary(const uint8_t *shard, uint64_t shard_blocks, uint32_t g, uint32_t nshards, uint32_t qb, uint64_t *ab) {
  return shk::shard_summary(shard, shard_blocks, g, nshards, qb, ab, ab + 1);
}
static thread_local std::vector<uint8_t> t_spill_slots, t_spill_runends;
long long shkh_shard_layout(const uint8_t *shard, uint64_t shard_blocks, uint32_t g, uint32_t nshards, uint32_t qb, uint64_t free_in,
                            uint8_t *own_blocks, uint64_t own_bytes, uint64_t *spill_start, uint64_t *free_out) {
  int rc = shk::shard_layout(shard, shard_blocks, g, nshards, qb, free_in, own_blocks, own_bytes, &t_spill_slots, &t_spill_runends,
                             spill_start, free_out);
  return rc ? rc : (long long)t_spill_slots.size();
}
void shkh_shard_spill(uint8_t *slots, uint8_t *runends) {
  if (!t_spill_slots.empty()) { memcpy(slots, t_spill_slots.data(), t_spill_slots.size()); memcpy(runends, t_spill_runends.data(), t_spill_runends.size()); }
}
int shkh_shard_apply_spill(uint8_t *own_blocks, uint32_t g, uint32_t nshards, uint32_t qb, uint64_t spill_start, const uint8_t *slots,
                           const uint8_t *runends, uint64_t n) {
  return shk::shard_apply_spill(own_blocks, g, nshards, qb, spill_start, slots, runends, n);
}
// the chunker as an iterator (the multi-GPU launcher shk/count.py reads its parts through this)
void *shkh_batch_open(const char **paths, int nfiles, int mode, uint64_t part_size, uint32_t overhead) {
  std::vector<std::string> f(paths, paths + nfiles);
  return new shk::seqFile_batch(f, shk::FASTQ, mode == 1 ? shk::GZIP : mode == 2 ? shk::BZIP2 : shk::TEXT, part_size, overhead);
}
// 1 = a part (malloc'ed: shkh_free it), 0 = end, -1 = error (wrong input file)
int shkh_batch_next(void *h, char **data, uint64_t *size) {
  shk::seqFile_batch *b = (shk::seqFile_batch *)h;
  shk::chunk c;
  if (b->getDataChunk(c)) { *data = c.get_reads(); *size = c.get_size(); return 1; }
  return b->bad() ? -1 : 0;
}
int shkh_batch_files(void *h) { return ((shk::seqFile_batch *)h)->num_files(); }
void shkh_free(void *p) { free(p); }
void shkh_batch_close(void *h) { delete (shk::seqFile_batch *)h; }
// alpha = -1: the true:false ratio comes from the error profile file
void shkh_size_filter_profile(int K, uint64_t n_true, uint64_t N_total, const char *profile, int num_denoise, double fr, uint64_t *out) {
  shk::Sizing s = shk::size_filter(K, n_true, N_total, -1, profile, num_denoise, fr);
  out[0] = s.qb; out[1] = s.hb; out[2] = (uint64_t)s.num_deNoise; out[3] = s.n_distinct_elts_for_DeNoise;
  out[4] = s.num_true_kmers; out[5] = s.num_false_kmers; out[6] = (uint64_t)s.lower_bound; out[7] = (uint64_t)s.upper_bound;
}
double shkh_true_to_false_ratio(const double *rates, uint64_t n, uint64_t K) {
  return shk::ErrorProfile(std::vector<double>(rates, rates + n)).true_to_false_ratio(K);
}
int shkh_rounds_for_loss_rate(double mean, double fr) { return shk::rounds_for_loss_rate(mean, fr); }
// the cut fastq_read_parts would make in this buffer (0 = none found)
uint64_t shkh_record_cut(const char *text, uint64_t n, uint32_t overhead) { return shk::fastq_record_cut(text, n, overhead); }
void shkh_size_filter(int K, uint64_t n_true, uint64_t N_total, double alpha, int num_denoise, double fr, uint64_t *out) {
  shk::Sizing s = shk::size_filter(K, n_true, N_total, alpha, "", num_denoise, fr);
  out[0] = s.qb; out[1] = s.hb; out[2] = (uint64_t)s.num_deNoise; out[3] = s.n_distinct_elts_for_DeNoise;
  out[4] = s.num_true_kmers; out[5] = s.num_false_kmers; out[6] = (uint64_t)s.lower_bound; out[7] = (uint64_t)s.upper_bound;
}
}
