// Contiger's inner step on the device (first slice of SURVEY.md 8 a-13): the forward extension of
// get_unitig_forward (src/contig_assembly.cpp:3054-3190) for many open unitig ends at once, in the
// case where no other unitig interferes (startKmer2unitig has no entry for any neighbour). That part
// is a pure function of (filter, end k-mer): it is what costs the time in the reference (up to 7 filter
// lookups + ntHash rolls per extended base); seed selection, the traveled-bit / start-k-mer protocol
// between concurrent walks, duplicate removal and the graph passes stay with the host (next rounds).
//
// One thread per open end. The k-mer window lives in registers as 2 bits per base (A,C,G,T = 0..3,
// the order of DNA_bases, base/global.h:110), k <= 64. Per step:
//   after:  the 4 k-mers window[1..] + x                      (contig_assembly.cpp:3067-3088)
//   before: the 3 siblings z + window[1..], z != window[0]    (:3090-3120: "kmers with RC(current_kmer_fix)
//           as prefix", minus the current k-mer itself)
//   stop when a sibling is solid or more than one successor is (:3122), extend when exactly one
//   successor is solid (:3167-3190; stop on a pure circle :3176), stop when none is (:3201).
// Hashes are rolled: for s = s0 s1..s(k-1), fh = XOR rol(seed(si), k-1-i), rh = XOR rol(seedc(si), i):
//   successor  fh' = rol1(fh) ^ rol(seed(s0), k) ^ seed(x);   rh' = ror1(rh ^ seedc(s0)) ^ rol(seedc(x), k-1)
//   sibling    fh" = fh ^ rol(seed(s0) ^ seed(z), k-1);       rh" = rh ^ seedc(s0) ^ seedc(z)
#define SHK_WALK_MAX_K 64   // stop reasons: SHK_STOP_* of include/shk.h

__device__ __forceinline__ uint64_t shk_code_seed(unsigned c) {      // seedTab column 0 by 2-bit code
  return c == 0 ? SHK_SEED_A : c == 1 ? SHK_SEED_C : c == 2 ? SHK_SEED_G : SHK_SEED_T;
}
__device__ __forceinline__ uint64_t shk_code_seed_rc(unsigned c) { return shk_code_seed(3 - c); }

__global__ void k_extend_forward(uint8_t *tab, uint64_t q_lo, uint64_t nslots, uint32_t hb, const char *cur_kmers,
                                 const char *first_kmers, uint32_t n, uint32_t k, uint64_t abundance_min, int mark,
                                 uint32_t max_ext, char *out_bases, uint32_t *out_counts, uint32_t *out_n,
                                 uint8_t *out_stop, uint8_t *out_branch, uint32_t *out_ncount) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t kmask = hb >= 64 ? ~0ULL : ((1ULL << hb) - 1);
  // pack the window and the first k-mer; hash the window once
  unsigned __int128 win = 0, first = 0;
  uint64_t fh = 0, rh = 0;
  bool bad = false;
  for (uint32_t j = 0; j < k; j++) {
    const char c = cur_kmers[(size_t)i * k + j], f = first_kmers[(size_t)i * k + j];
    const unsigned cc = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
    const unsigned fc = f == 'A' ? 0u : f == 'C' ? 1u : f == 'G' ? 2u : f == 'T' ? 3u : 4u;
    if (cc > 3 || fc > 3) { bad = true; break; }
    win = (win << 2) | cc;
    first = (first << 2) | fc;
    fh ^= shk_rol64(shk_code_seed(cc), (k - 1 - j) & 63);
    rh ^= shk_rol64(shk_code_seed_rc(cc), j & 63);
  }
  if (bad) { out_n[i] = 0; out_stop[i] = SHK_STOP_BAD_SEED; out_branch[i] = 0; return; }
  uint32_t ncount[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // filter counts of the neighbours named in `branch`
  const unsigned __int128 wmask = k == 64 ? ~(unsigned __int128)0 : (((unsigned __int128)1 << (2 * k)) - 1);
  uint32_t nout = 0;
  uint8_t stop = 0, trav, branch = 0;   // branch: solid successors (bits 0-3, by base) and solid siblings (bits 4-7) at the stop
  while (!stop) {
    const unsigned s0 = (unsigned)(win >> (2 * (k - 1))) & 3u;
    // successors
    const uint64_t fbase = shk_rol64(fh, 1) ^ shk_rol64(shk_code_seed(s0), k & 63);
    const uint64_t rbase = shk_ror64(rh ^ shk_code_seed_rc(s0), 1);
    uint32_t ncand = 0, xc = 0;
    uint8_t masks = 0;
    uint32_t nc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t cnt_x = 0, fh_x = 0, rh_x = 0;
    for (unsigned x = 0; x < 4; x++) {
      const uint64_t f = fbase ^ shk_code_seed(x);
      const uint64_t r = rbase ^ shk_rol64(shk_code_seed_rc(x), (k - 1) & 63);
      const uint64_t cnt = shk_lookup_one(tab, (f < r ? f : r) & kmask, q_lo, nslots, mark, &trav);
      if (cnt >= abundance_min) { ncand++; xc = x; cnt_x = cnt; fh_x = f; rh_x = r; masks |= (uint8_t)(1u << x); nc[x] = cnt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cnt; }
    }
    // siblings (other predecessors of my successors)
    uint32_t nbefore = 0;
    for (unsigned z = 0; z < 4; z++) {
      if (z == s0) continue;
      const uint64_t f = fh ^ shk_rol64(shk_code_seed(s0) ^ shk_code_seed(z), (k - 1) & 63);
      const uint64_t r = rh ^ shk_code_seed_rc(s0) ^ shk_code_seed_rc(z);
      const uint64_t cnt = shk_lookup_one(tab, (f < r ? f : r) & kmask, q_lo, nslots, mark, &trav);
      if (cnt >= abundance_min) { nbefore++; masks |= (uint8_t)(16u << z); nc[4 + z] = cnt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cnt; }
    }
    if (nbefore || ncand > 1) {
      stop = SHK_STOP_BRANCH; branch = masks;
      for (int j = 0; j < 8; j++) ncount[j] = nc[j];
      break;
    }
    if (ncand == 0) { stop = SHK_STOP_DEAD_END; break; }
    const unsigned __int128 next = ((win << 2) | xc) & wmask;
    if (next == first) { stop = SHK_STOP_CIRCLE; break; }
    if (nout >= max_ext) { stop = SHK_STOP_BUFFER; break; }
    out_bases[(size_t)i * max_ext + nout] = "ACGT"[xc];
    out_counts[(size_t)i * max_ext + nout] = cnt_x > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cnt_x;
    nout++;
    win = next; fh = fh_x; rh = rh_x;
  }
  out_n[i] = nout;
  out_stop[i] = stop;
  out_branch[i] = branch;
  for (int j = 0; j < 8; j++) out_ncount[(size_t)i * 8 + j] = ncount[j];
}

// Seed of a read (processDataChunk, contig_assembly.cpp:1856-1876): the k-mer at len/2 - k/2, upper-cased, no 'N';
// looked up (and marked traveled when mark == 1, as count_key_value_set_traveled does); kept when it was not
// traveled before and its count lies in [count_min, count_max]. One thread per read; out_counts[r] = 0 means none.
__global__ void k_select_seeds(uint8_t *tab, uint64_t q_lo, uint64_t nslots, uint32_t hb, const uint8_t *text,
                               const uint64_t *rd_start, const uint64_t *rd_end, uint64_t read_lo, uint64_t nreads, uint32_t k, uint64_t count_min,
                               uint64_t count_max, int mark, char *out_seeds, uint32_t *out_counts) {
  // reads [read_lo, nreads) of the batch
  const uint64_t r = read_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nreads) return;
  out_counts[r] = 0;
  const uint64_t st = rd_start[r], en = rd_end[r];
  if (en < st || en - st < k || en - st > SHK_MAX_READ) return;
  const uint32_t len = (uint32_t)(en - st);
  const int middle = (int)(len / 2) - (int)(k / 2);
  if (middle < 0 || (uint32_t)middle > len - k) return;
  uint64_t fh = 0, rh = 0;
  for (uint32_t j = 0; j < k; j++) {
    unsigned ch = text[st + (uint32_t)middle + j];
    if (ch >= 'a' && ch <= 'z') ch -= 32;        // to_upper_DNA
    const unsigned cc = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
    if (cc > 3) return;                          // 'N' (any byte that is not a base has no seed here)
    out_seeds[r * k + j] = (char)ch;
    fh ^= shk_rol64(shk_code_seed(cc), (k - 1 - j) & 63);
    rh ^= shk_rol64(shk_code_seed_rc(cc), j & 63);
  }
  const uint64_t kmask = hb >= 64 ? ~0ULL : ((1ULL << hb) - 1);
  uint8_t trav = 0;
  const uint64_t cnt = shk_lookup_one(tab, (fh < rh ? fh : rh) & kmask, q_lo, nslots, mark, &trav);
  if (mark == 1 && trav) return;
  if (cnt < count_min || cnt > count_max) return;
  out_counts[r] = cnt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cnt;
}
