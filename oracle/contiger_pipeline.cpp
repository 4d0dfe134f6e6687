/* TEST INFRASTRUCTURE ONLY -- a second, complete, independent reading of Contiger (src/contig_assembly.cpp) as ONE
 * sequential program on top of the oracle's filter (cqf_oracle.c). Nothing under sh-assembly_amd/ includes, links or
 * calls this.
 *
 * Restated, in the order the reference runs them:
 *   main                      src/contig_assembly.cpp:246-303, 586-629   (load, find, check, renumber, links, write)
 *   find_unitigs_mt_master    :2034-2172   seeds in read order, seed position len/2                        (rule 0)
 *   processDataChunk          :1839-1924   the workers' copy of that loop, seed position len/2 - K/2       (rule 1)
 *   find_unitigs_mt_worker    :2254-2269   queued contigs: ONE get_unitig_forward each, no turn-round
 *   WorkQueue                 :847-882     concurrent_queue = FIFO
 *   insert_or_replace         :3018-3025   "smaller id wins"
 *   get_unitig_forward        :3028-3218   incl. traveled bits, known nodes, candidates, pure circles
 *   check_unitig              :935-954
 *   track_kmer_worker         :956-1010    renumbering 1..M, +id / -id map values (stale entries stay, as there)
 *   build_graph_worker        :1012-1084   afterNodes in A,C,G,T order, beforeNodes in T,G,C,A order
 *   writer                    :600-629     ">i LN:i: KC:i: km:f: L:..." with the int-truncated median
 *   chunk::readLine/skipLines cqf/chunk.h:74-93; Contig base/Utility.h:28-55; median base/Utility.cpp:27-40;
 *   to_upper_DNA base/Utility.h:120-154; DNAString's 2-bit code (every non-ACGT byte becomes 'A') base/DNA_string.h:23
 *
 * What a sequential program has to decide that the reference leaves to its threads (master + t workers on TBB
 * containers, ids = concurrent_vector::push_back order): which loop reads a chunk (`rule`), and when queued contigs are
 * taken. Here one loop reads every chunk with the given rule, and the queue is drained (FIFO) after every read
 * (policy bit 0 clear; the master's throttle :2145-2148 keeps the backlog at <= t) or only at the end of each chunk (bit 0
 * set). Tests run several schedules; what they compare with the device output is order-free: the canonical sequence
 * set, per-unitig median / KC, the canonicalised link set.
 *
 * Every k-mer is hashed from scratch (orc_nthash); the reference rolls with swapped arguments, which
 * tests/test_oracle.py::test_contiger_roll_sequence... pins to the same values on the real NTPC64.
 *
 * PARITY UNPINNED: src/contig_assembly.cpp needs boost and TBB (absent here: unbuildable) and the reference ships no
 * fixtures, so nothing pins this reading to the reference's OUTPUT. It replaces "one function restated" by "the whole
 * pipeline restated a second time, independently of the device code". */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <deque>
#include <string>
#include <unordered_map>
#include <vector>

extern "C" {
#include "cqf_oracle.h"
}

namespace {

const char DNA_bases[4] = {'A', 'C', 'G', 'T'}; /* base/global.h:110 */

char rc_base(char c) { /* RC_DNAbase, base/Utility.h:96-118 (input is ACGT by construction) */
  switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    default: return 'A';
  }
}
std::string rc(const std::string &s) {
  std::string r(s.rbegin(), s.rend());
  for (auto &c : r) c = rc_base(c);
  return r;
}
/* what DNAString keeps of a string: two bits per base, every byte that is not ACGT/acgt is 00 = 'A' (base2bits) */
std::string dna(const std::string &s) {
  std::string r = s;
  for (auto &c : r) {
    switch (c) {
      case 'A': case 'a': c = 'A'; break;
      case 'C': case 'c': c = 'C'; break;
      case 'G': case 'g': c = 'G'; break;
      case 'T': case 't': c = 'T'; break;
      default: c = 'A';
    }
  }
  return r;
}
double median(std::vector<int> &v) { /* base/Utility.cpp:27-40 */
  if (v.empty()) return 0;
  if (v.size() == 1) return v[0];
  std::sort(v.begin(), v.end());
  size_t t = v.size() / 2;
  return v.size() % 2 == 0 ? (v[t - 1] + v[t]) / 2.0 : v[t];
}

struct Contig { /* base/Utility.h:28-55 */
  std::string seq;
  int median_abundance;
};

struct Run {
  orc_qf *qf;
  unsigned K;
  uint64_t abundance_min, solid_min, solid_max;
  std::vector<Contig> contigs;                            /* concurrent_vector<Contig>; [0] is the dummy of main :269 */
  std::unordered_map<std::string, long> start;            /* startKmer2unitig */
  std::deque<size_t> queue;                               /* WorkQueue::jobQueue */
  uint64_t lookups = 0, seeds = 0, queued = 0, cleared = 0;

  /* cqf.count_key_value_set_traveled(kmer_hash % range, count): was-traveled, sets the bit (CQF_mt.h:506-508) */
  bool lookup_set_traveled(const char *kmer, uint64_t *count) {
    uint64_t fh, rh;
    orc_nthash(kmer, K, &fh, &rh);
    uint64_t h = rh < fh ? rh : fh;
    if (qf->hb < 64) h &= (1ULL << qf->hb) - 1;
    lookups++;
    return orc_qf_count_set_traveled(qf, h, count) != 0;
  }
  bool insert_or_replace(const std::string &km, size_t idx) { /* :3018-3025 */
    auto it = start.find(km);
    if (it == start.end()) { start.emplace(km, (long)idx); return true; }
    if (it->second >= (long)idx) { it->second = (long)idx; return true; }
    return false;
  }
  void clear(size_t id) { contigs[id].seq.clear(); cleared++; }

  void get_unitig_forward(size_t id) { /* :3028-3218 */
    std::string first_kmer = contigs[id].seq.substr(0, K);
    std::string current_kmer = contigs[id].seq.substr(contigs[id].seq.size() - K);
    std::string current_kmer_RC = rc(current_kmer);
    std::vector<int> abundances(contigs[id].seq.size() - K + 1, contigs[id].median_abundance);
    for (;;) {
      std::string current_kmer_fix = current_kmer.substr(1);
      bool candidates_before[4] = {false, false, false, false}, candidates_after[4] = {false, false, false, false};
      uint64_t ab_before[4] = {0, 0, 0, 0}, ab_after[4] = {0, 0, 0, 0};
      int candidates_before_num = 0, candidates_after_num = 0, nodes_before_num = 0, nodes_after_num = 0;
      uint64_t kmer_count;
      /* k-mers with current_kmer_fix as prefix (:3064-3088) */
      for (int x = 0; x < 4; x++) {
        std::string kmer = current_kmer_fix + DNA_bases[x];
        bool isTraveled = lookup_set_traveled(kmer.c_str(), &kmer_count);
        if (kmer_count >= abundance_min) {
          if (isTraveled && start.count(kmer)) {
            nodes_after_num++;
            ab_after[x] = kmer_count;
          } else {
            ab_after[x] = kmer_count;
            candidates_after[x] = true;
            candidates_after_num++;
          }
        }
      }
      /* k-mers with RC(current_kmer_fix) as prefix, except RC(current_kmer) itself (:3090-3120) */
      std::string kmer = current_kmer_RC;
      for (int x = 0; x < 4; x++) {
        if (DNA_bases[x] == current_kmer_RC[K - 1]) continue;
        kmer[K - 1] = DNA_bases[x];
        bool isTraveled = lookup_set_traveled(kmer.c_str(), &kmer_count);
        if (kmer_count >= abundance_min) {
          if (isTraveled && start.count(kmer)) {
            nodes_before_num++;
          } else {
            ab_before[x] = kmer_count;
            candidates_before[x] = true;
            candidates_before_num++;
          }
        }
      }
      if ((nodes_before_num + candidates_before_num) || (nodes_after_num + candidates_after_num) > 1) { /* :3122 */
        if (!insert_or_replace(current_kmer_RC, id)) { clear(id); break; }
        contigs[id].median_abundance = (int)median(abundances);
        for (int x = 0; x < 4; x++)
          if (candidates_after[x]) push_candidate(current_kmer_fix + DNA_bases[x], ab_after[x]);
        kmer = current_kmer_RC;
        for (int x = 0; x < 4; x++)
          if (candidates_before[x]) { kmer[K - 1] = DNA_bases[x]; push_candidate(kmer, ab_before[x]); }
        break;
      } else if (candidates_after_num == 1) { /* :3162 */
        int x = 0;
        for (x = 0; x < 4; x++) if (candidates_after[x]) break;
        current_kmer = current_kmer_fix + DNA_bases[x];
        if (current_kmer == first_kmer) { /* a pure circle (:3176-3183) */
          if (!insert_or_replace(first_kmer, id) || !insert_or_replace(current_kmer_RC, id)) clear(id);
          else contigs[id].median_abundance = (int)median(abundances);
          break;
        }
        current_kmer_RC = rc_base(DNA_bases[x]) + current_kmer_RC.substr(0, K - 1);
        contigs[id].seq += DNA_bases[x];
        abundances.push_back((int)ab_after[x]);
      } else { /* one known node behind the end (:3192-3199), or nothing solid (:3200-3210) */
        if (!insert_or_replace(current_kmer_RC, id)) clear(id);
        else contigs[id].median_abundance = (int)median(abundances);
        break;
      }
    }
  }
  void push_candidate(const std::string &kmer, uint64_t abundance) { /* :3133-3160 */
    if (start.find(kmer) != start.end()) return;            /* insert(access, kmer) false: already a key */
    contigs.push_back(Contig{kmer, (int)abundance});
    start.emplace(kmer, (long)(contigs.size() - 1));
    queue.push_back(contigs.size() - 1);
    queued++;
  }
  void drain() { /* find_unitigs_mt_worker :2254-2269 */
    while (!queue.empty()) {
      size_t id = queue.front();
      queue.pop_front();
      get_unitig_forward(id);
    }
  }

  /* one data chunk: the master's loop (:2049-2149, rule 0) or processDataChunk (:1839-1924, rule 1) */
  void chunk(const char *p, uint64_t n, int rule, bool drain_per_read) {
    const char *end = p + n;
    auto readLine = [&](std::string &s) -> bool { /* chunk::readLine */
      if (p == end) return false;
      const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
      if (!nl) nl = end;              /* the reference dereferences NULL here; a chunk always ends in '\n' (CQF_mt.h:793-808) */
      s.assign(p, (size_t)(nl - p));
      p = nl < end ? nl + 1 : end;
      return true;
    };
    auto skipLines = [&](int num) {
      while (num--) {
        if (p == end) return;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        p = nl ? nl + 1 : end;
      }
    };
    std::string line, seq;
    while (readLine(line)) {
      if (line.empty()) continue;
      if (line[0] != '@') continue;
      if (!readLine(seq)) break;
      if (seq.length() < K) { skipLines(2); continue; }
      int seq_len = (int)seq.length();
      int middle = rule == 0 ? seq_len / 2 : seq_len / 2 - (int)K / 2;
      if (middle <= seq_len - (int)K) {
        std::string kmer = seq.substr((size_t)middle, K);
        for (auto &c : kmer) if (c > 0x60) c -= 32;          /* to_upper_DNA */
        /* `continue` in the reference skips skipLines(2): the '+' and quality lines are then read as candidate
         * headers; only a quality line that starts with '@' would change anything. Kept as written. */
        if (kmer.find_first_of("nN") != std::string::npos) continue;
        uint64_t kmer_count;
        if (lookup_set_traveled(kmer.c_str(), &kmer_count)) continue;
        if (kmer_count < solid_min || kmer_count > solid_max) continue;
        contigs.push_back(Contig{dna(kmer), (int)kmer_count});
        size_t contig_id = contigs.size() - 1;
        seeds++;
        std::string key = dna(kmer);
        get_unitig_forward(contig_id);
        if (!contigs[contig_id].seq.empty()) {
          auto it = start.find(key);
          if (it != start.end()) {
            if (it->second > (long)contig_id) {
              contigs[contig_id].seq = rc(contigs[contig_id].seq);
              get_unitig_forward(contig_id);
            } else if (it->second < (long)contig_id) {
              clear(contig_id);
            }
          } else {
            contigs[contig_id].seq = rc(contigs[contig_id].seq);
            get_unitig_forward(contig_id);
          }
        }
      }
      skipLines(2);
      if (drain_per_read) drain();
    }
    drain();
  }

  std::string finish() {
    /* check_unitig :935-954 */
    for (size_t id = 1; id < contigs.size(); id++) {
      if (contigs[id].seq.empty()) continue;
      auto it = start.find(contigs[id].seq.substr(0, K));
      if (it != start.end() && it->second != (long)id) clear(id);
    }
    /* track_kmer_worker :956-1010 */
    long counter = 1;
    for (size_t id = 1; id < contigs.size(); id++) {
      if (contigs[id].seq.empty()) continue;
      std::string first_kmer = contigs[id].seq.substr(0, K);
      std::string last_kmer_RC = rc(contigs[id].seq.substr(contigs[id].seq.size() - K));
      if (first_kmer == last_kmer_RC) {
        auto it = start.find(first_kmer);
        if (it != start.end()) it->second = counter;
      } else {
        auto it = start.find(last_kmer_RC);
        if (it != start.end()) it->second = -counter;
        it = start.find(first_kmer);
        if (it != start.end()) it->second = counter;
      }
      counter++;
    }
    /* build_graph_worker :1012-1084 + writer :600-629 */
    std::string out;
    char buf[160];
    size_t n = 0;
    for (size_t id = 1; id < contigs.size(); id++) {
      const Contig &c = contigs[id];
      if (c.seq.empty()) continue;
      /* KC: int * (size_t - int + 1) is evaluated in size_t */
      size_t kc = (size_t)c.median_abundance * (c.seq.size() - K + 1);
      snprintf(buf, sizeof buf, ">%zu LN:i:%zu KC:i:%zu km:f:%d", n, c.seq.size(), kc, c.median_abundance);
      out += buf;
      std::string fix = c.seq.substr(c.seq.size() - K + 1, K - 1);
      for (int x = 0; x < 4; x++) {
        auto it = start.find(fix + DNA_bases[x]);
        if (it == start.end()) continue;
        long t = it->second;
        if (t > 0) snprintf(buf, sizeof buf, " L:+:%ld:+", t - 1);
        else snprintf(buf, sizeof buf, " L:+:%ld:-", -t - 1);
        out += buf;
      }
      fix = rc(c.seq.substr(0, K - 1));
      for (int x = 3; x >= 0; x--) {
        auto it = start.find(fix + DNA_bases[x]);
        if (it == start.end()) continue;
        long t = it->second;
        if (t > 0) snprintf(buf, sizeof buf, " L:-:%ld:+", t - 1);
        else snprintf(buf, sizeof buf, " L:-:%ld:-", -t - 1);
        out += buf;
      }
      out += "\n";
      out += c.seq;
      out += "\n";
      n++;
    }
    return out;
  }
};

} /* namespace */

extern "C" {

/* The whole of Contiger on one filter (its traveled bits are set, as the reference's are). `text` holds the reads; the
 * chunks (offset, length) are what seqFile_batch::getDataChunk hands out, in order. policy: bit 0 = drain the queue only
 * at the end of each chunk; bits 1.. = seed rule per chunk: 0 every chunk by the master's loop (rule 0), 1 every chunk by
 * processDataChunk (rule 1), 2 alternate (even chunks master). Returns the text of unitigs.fa (malloc'ed; free with
 * orc_contiger_free) and stats = {seeds taken, contigs queued, contigs cleared, filter lookups, contigs in total, map entries}. */
char *orc_contiger_run(orc_qf *qf, const char *text, const uint64_t *chunk_off, const uint64_t *chunk_len, uint32_t nchunks,
                       unsigned k, uint64_t abundance_min, uint64_t solid_min, uint64_t solid_max, uint32_t policy,
                       uint64_t *out_len, uint64_t stats[6]) {
  Run r;
  r.qf = qf;
  r.K = k;
  r.abundance_min = abundance_min;
  r.solid_min = solid_min;
  r.solid_max = solid_max;
  r.contigs.resize(1);
  const uint32_t who = policy >> 1;
  for (uint32_t c = 0; c < nchunks; c++)
    r.chunk(text + chunk_off[c], chunk_len[c], who == 2 ? (int)(c & 1) : (int)who, !(policy & 1));
  std::string out = r.finish();
  if (stats) {
    stats[0] = r.seeds; stats[1] = r.queued; stats[2] = r.cleared; stats[3] = r.lookups;
    stats[4] = r.contigs.size() - 1; stats[5] = r.start.size();
  }
  char *p = (char *)malloc(out.size() + 1);
  memcpy(p, out.data(), out.size());
  p[out.size()] = 0;
  *out_len = out.size();
  return p;
}
void orc_contiger_free(char *p) { free(p); }

} /* extern "C" */
