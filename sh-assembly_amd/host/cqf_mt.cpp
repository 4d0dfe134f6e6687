#include "cqf_mt.hpp"

#include <stdlib.h>
#include <string.h>
#include <iostream>

namespace shk {

void CQF_mt::ensure_ctx(int ksize, uint64_t trigger, uint32_t rounds, uint64_t max_bytes) {
  if (ctx) return;
  shk_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.qb = (uint32_t)qb; cfg.hb = (uint32_t)hb; cfg.seed = seed; cfg.k = (uint32_t)ksize;
  cfg.ndistinct_for_denoise = trigger; cfg.num_denoise = rounds; cfg.min_denoise_len = min_denoise_len;
  cfg.max_batch_bytes = max_bytes;
  cfg.max_batch_keys = max_bytes;        // one k-mer needs at least one input byte
  cfg.max_batch_reads = max_bytes / 8 + 1024;
  cfg.device = device;
  chk(shk_create(&cfg, &ctx));
}

void CQF_mt::build_KmerSpectrum(const std::vector<std::string> &fnames, const FILE_TYPE ftype, const FILE_MODE fmode,
                                int ksize, uint64_t, uint64_t n_distinct_elts_for_DeNoise, uint32_t n_deNoise,
                                bool end_deNoise, double) {
  time_t start_time = time(NULL);
  const uint64_t max_bytes = (uint64_t)parts_per_call * (part_size + overhead);
  ensure_ctx(ksize, n_distinct_elts_for_DeNoise, n_deNoise, max_bytes);
  seqFile_batch files(fnames, ftype, fmode, part_size, overhead);
  std::vector<char> text;
  std::vector<uint64_t> off, len;
  text.reserve(max_bytes);
  auto flush = [&]() {
    shk_batch_stats st;
    chk(shk_count_chunks(ctx, text.data(), 0, text.size(), off.data(), len.data(), (uint32_t)off.size(), &st));
    for (uint32_t r = 0; r < st.denoise_rounds; r++) {
      // CQF_mt.h:868/912 print these lines around every round
      std::cerr << "Finished DeNoise: ndistinct_elts/total_elts." << ndistinct_elts() << "/" << nelts() << std::endl;
    }
    denoise_rounds_done += st.denoise_rounds;
    removed_total += st.removed;
    text.clear(); off.clear(); len.clear();
  };
  chunk c;
  while (files.getDataChunk(c)) {
    off.push_back(text.size());
    len.push_back(c.get_size());
    text.insert(text.end(), c.get_reads(), c.get_reads() + c.get_size());
    free(c.get_reads());
    if (off.size() == parts_per_call) flush();
  }
  if (files.bad()) throw std::runtime_error("Error: Wrong input file!");
  if (!off.empty()) flush();
  if (end_deNoise) {   // CQF_mt.h:860: one more round after the last part, not counted in n_deNoise
    uint64_t removed = 0;
    chk(shk_denoise(ctx, &removed));
    removed_total += removed;
    denoise_rounds_done++;
  }
  std::cerr << "Time for building K-mer spectrum without dumping to disk: " << difftime(time(NULL), start_time) << " seconds."
            << std::endl;
}

void CQF_mt::save(const std::string &filename) { chk(shk_export_cqf(ctx, filename.c_str())); }

void CQF_mt::load(const std::string &filename) {
  FILE *f = fopen(filename.c_str(), "rb");
  if (!f) throw std::runtime_error("Error opening file for deserializing");
  uint8_t hdr[128];
  if (fread(hdr, 128, 1, f) != 1) { fclose(f); throw std::runtime_error("short .cqf header"); }
  fclose(f);
  uint64_t nslots, key_bits, krb;
  memcpy(&nslots, hdr + 16, 8); memcpy(&key_bits, hdr + 32, 8); memcpy(&krb, hdr + 48, 8);
  memcpy(&seed, hdr + 8, 4);
  hb = key_bits; qb = hb - krb;        // CQF_mt.h:517-518
  if (ctx) { shk_destroy(ctx); ctx = nullptr; }
  ensure_ctx(21, ~0ULL >> 2, 0, 1 << 20);
  chk(shk_import_cqf(ctx, filename.c_str()));
}

void CQF_mt::count_batch(const std::vector<uint64_t> &keys, int mode, std::vector<uint64_t> &counts, std::vector<uint8_t> &trav) {
  counts.resize(keys.size()); trav.resize(keys.size());
  chk(shk_lookup(ctx, keys.data(), keys.size(), 0, mode, counts.data(), trav.data()));
}
uint64_t CQF_mt::count(uint64_t key) {
  uint64_t c = 0; uint8_t tr = 0;
  chk(shk_lookup(ctx, &key, 1, 0, 2, &c, &tr));
  return c;
}
bool CQF_mt::count_key_value_set_traveled(uint64_t key, uint64_t &count) {
  uint8_t tr = 0;
  chk(shk_lookup(ctx, &key, 1, 0, 1, &count, &tr));
  return tr != 0;
}
bool CQF_mt::count_key_value_is_traveled(uint64_t key, uint64_t &count) {
  uint8_t tr = 0;
  chk(shk_lookup(ctx, &key, 1, 0, 0, &count, &tr));
  return tr != 0;
}

void CQF_mt::print_metadata() {
  uint8_t h[128];
  chk(shk_header(ctx, h));
  auto u64 = [&](int off) { uint64_t v; memcpy(&v, h + off, 8); return v; };
  uint32_t sd; memcpy(&sd, h + 8, 4);
  std::cerr << "#metadata" << std::endl
            << "size: " << u64(0) << std::endl << "seed: " << sd << std::endl << "nslots: " << u64(16) << std::endl
            << "xnslots: " << u64(24) << std::endl << "key_bits: " << u64(32) << std::endl << "value_bits: " << u64(40) << std::endl
            << "key_remainder_bits: " << u64(48) << std::endl << "bits_per_slots: " << u64(56) << std::endl
            << "nelts: " << u64(88) << std::endl << "ndistinct_elts: " << u64(96) << std::endl
            << "noccupied_slots: " << u64(104) << std::endl << "num_locks: " << u64(112) << std::endl;
}

}  // namespace shk
