// class CQF_mt: the reference's wrapper (cqf/CQF_mt.h:414-558) with the same member names
// and argument meaning, implemented on the shk C ABI (the filter lives on the GPU).
#pragma once
#include <stdint.h>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/shk.h"
#include "fastq_chunker.hpp"

namespace shk {

class CQF_mt {
 public:
  uint64_t qb = 0, hb = 0;
  uint16_t t = 1;  // kept for interface compatibility; the device does the parallel work

  CQF_mt() {}
  // cqf/CQF_mt.h:427-445
  CQF_mt(uint64_t qb_, uint64_t hb_, uint16_t t_ = 1, uint32_t seed_ = 2038074761) : qb(qb_), hb(hb_), t(t_), seed(seed_) {
    if (qb > 40 || hb != qb + 8) throw std::invalid_argument("Only qb <= 40 with hb == qb+8 is supported.");
  }
  ~CQF_mt() { if (ctx) shk_destroy(ctx); }

  // test hooks (not in the reference): part geometry and deNoise range length
  uint64_t part_size = 1ULL << 23;
  uint32_t overhead = 65535;
  uint64_t min_denoise_len = 0;
  uint32_t parts_per_call = 64;
  int device = 0;

  // cqf/CQF_mt.h:959-995. Returns through the members nelts()/ndistinct_elts().
  void build_KmerSpectrum(const std::vector<std::string> &fnames, const FILE_TYPE ftype, const FILE_MODE fmode, int ksize,
                          uint64_t n_distinct_true_elts, uint64_t n_distinct_elts_for_DeNoise, uint32_t n_deNoise,
                          bool end_deNoise, double p_true_kmer_singleton);
  void save(const std::string &filename);   // CQF_mt.h:521 -> qf_serialize
  void load(const std::string &filename);   // CQF_mt.h:514 -> qf_deserialize
  uint64_t count(uint64_t key);             // CQF_mt.h:462
  bool count_key_value_set_traveled(uint64_t key, uint64_t &count);   // CQF_mt.h:506
  bool count_key_value_is_traveled(uint64_t key, uint64_t &count);    // CQF_mt.h:509
  // batched forms for device-friendly callers
  void count_batch(const std::vector<uint64_t> &keys, int mode, std::vector<uint64_t> &counts, std::vector<uint8_t> &trav);
  void print_metadata();                    // CQF_mt.h:530-545
  uint64_t nelts() { shk_totals tt; chk(shk_stats(ctx, &tt)); return tt.nelts; }
  uint64_t ndistinct_elts() { shk_totals tt; chk(shk_stats(ctx, &tt)); return tt.ndistinct; }
  uint32_t denoise_rounds_done = 0;
  uint64_t removed_total = 0;

 private:
  shk_ctx *ctx = nullptr;
  uint32_t seed = 2038074761;
  void chk(int rc) { if (rc) throw std::runtime_error(std::string("libshk: ") + shk_strerror(rc)); }
  void ensure_ctx(int ksize, uint64_t trigger, uint32_t rounds, uint64_t max_bytes);
};

}  // namespace shk
