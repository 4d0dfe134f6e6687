#include "cqf_mt.hpp"

#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <iostream>
#include <mutex>
#include <thread>

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
  // Three page-locked batch buffers rotate: a reader thread fills one from the files while the copy of the
  // previous one runs on the context's copy stream (shk_upload_text) and the batch before that is counted.
  // The parts reach the filter in exactly the order the single-threaded loop would present them.
  struct Batch { char *text = nullptr; uint64_t bytes = 0; std::vector<uint64_t> off, len; bool last = false; };
  Batch ring[3];
  for (auto &bt : ring) { void *p = nullptr; chk(shk_host_alloc(max_bytes + 64, &p)); bt.text = (char *)p; }
  std::mutex mu;
  std::condition_variable cv;
  int filled = 0, taken = 0;          // batches produced / consumed (ring slot = index % 3)
  bool reader_bad = false;
  std::thread reader([&]() {
    chunk c;
    bool more = true;
    while (more) {
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return filled - taken < 3; }); }
      Batch &bt = ring[filled % 3];
      bt.bytes = 0; bt.off.clear(); bt.len.clear(); bt.last = false;
      while (bt.off.size() < parts_per_call) {
        if (!files.getDataChunk(c)) { more = false; break; }
        bt.off.push_back(bt.bytes);
        bt.len.push_back(c.get_size());
        memcpy(bt.text + bt.bytes, c.get_reads(), c.get_size());
        bt.bytes += c.get_size();
        free(c.get_reads());
      }
      bt.last = !more;
      if (files.bad()) reader_bad = true;
      { std::lock_guard<std::mutex> lk(mu); filled++; }
      cv.notify_all();
    }
  });
  auto pop = [&]() -> Batch * {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return filled > taken; });
    return &ring[taken % 3];
  };
  auto release = [&]() { { std::lock_guard<std::mutex> lk(mu); taken++; } cv.notify_all(); };
  std::string err;
  try {
    Batch *cur = pop();
    void *dcur = nullptr;
    if (cur->bytes) chk(shk_upload_text(ctx, cur->text, cur->bytes, &dcur));
    for (;;) {
      Batch *nxt = nullptr;
      void *dnxt = nullptr;
      if (!cur->last) {
        // the next batch is ring[(taken + 1) % 3]: wait for it and start its copy before counting the current one
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return filled > taken + 1; }); }
        nxt = &ring[(taken + 1) % 3];
        if (nxt->bytes) chk(shk_upload_text(ctx, nxt->text, nxt->bytes, &dnxt));
      }
      if (!cur->off.empty()) {
        shk_batch_stats st;
        chk(shk_count_chunks(ctx, dcur, 1, cur->bytes, cur->off.data(), cur->len.data(), (uint32_t)cur->off.size(), &st));
        for (uint32_t r = 0; r < st.denoise_rounds; r++) {
          // CQF_mt.h:868/912 print these lines around every round
          std::cerr << "Finished DeNoise: ndistinct_elts/total_elts." << ndistinct_elts() << "/" << nelts() << std::endl;
        }
        denoise_rounds_done += st.denoise_rounds;
        removed_total += st.removed;
      }
      const bool was_last = cur->last;
      release();
      if (was_last) break;
      cur = nxt; dcur = dnxt;
    }
  } catch (const std::exception &e) {
    err = e.what();
    // let the reader run to the end of the files so that it can be joined
    for (;;) {
      std::unique_lock<std::mutex> lk(mu);
      if (filled > taken) { const bool l = ring[taken % 3].last; taken++; lk.unlock(); cv.notify_all(); if (l) break; }
      else cv.wait(lk, [&] { return filled > taken; });
    }
  }
  reader.join();
  for (auto &bt : ring) shk_host_free(bt.text);
  if (!err.empty()) throw std::runtime_error(err);
  if (reader_bad || files.bad()) throw std::runtime_error("Error: Wrong input file!");
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
