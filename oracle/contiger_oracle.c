/* TEST INFRASTRUCTURE -- CPU restatement of Contiger's unitig extension, first slice.
 *
 * Restates, for one walk that meets no other unitig (startKmer2unitig never has an entry for a neighbour):
 *   get_unitig_forward            src/contig_assembly.cpp:3028-3218
 *   the two calls per seed        src/contig_assembly.cpp:1886-1904 (processDataChunk)
 *   median                        base/Utility.cpp:27-40 (stored in an int: Contig::median_abundance)
 * on top of the oracle's filter lookups (cqf_oracle.c). Every k-mer is hashed from scratch with the full
 * NTPC64 (the reference rolls; the device kernel rolls too -- hashing from scratch here keeps this side
 * independent of both).
 *
 * PARITY UNPINNED: the reference ships no fixtures for this path and src/contig_assembly.cpp cannot be
 * compiled in this image (boost, TBB), so nothing pins this restatement to the reference's output; it pins
 * the device kernel to this reading of the source. Not restated (next rounds): seed selection over the
 * reads, the traveled-bit / start-k-mer protocol between concurrent walks, duplicate removal, the graph
 * passes and the FASTA writer. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "cqf_oracle.h"

enum { ORC_STOP_BRANCH = 1, ORC_STOP_DEAD_END = 2, ORC_STOP_CIRCLE = 3, ORC_STOP_BUFFER = 4 };
static const char DNA_bases[4] = {'A', 'C', 'G', 'T'}; /* base/global.h:110 */

static char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c; }
static void rc_inplace(char *s, size_t n) {
  for (size_t i = 0; i < n / 2; i++) { char a = comp(s[i]), b = comp(s[n - 1 - i]); s[i] = b; s[n - 1 - i] = a; }
  if (n & 1) s[n / 2] = comp(s[n / 2]);
}
static uint64_t kmer_count(const orc_qf *qf, const char *kmer, unsigned k) {
  uint64_t fh, rh;
  orc_nthash(kmer, k, &fh, &rh);
  const uint64_t hv = fh < rh ? fh : rh;
  return orc_qf_count(qf, qf->hb >= 64 ? hv : (hv & ((1ULL << qf->hb) - 1))); /* kmer_hash % range */
}
static int cmp_int(const void *a, const void *b) { return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b); }
static int median_int(int *v, size_t n) {
  if (n == 0) return 0;
  if (n == 1) return v[0];
  qsort(v, n, sizeof(int), cmp_int);
  size_t t = n / 2;
  return n % 2 == 0 ? (int)((v[t - 1] + v[t]) / 2.0) : v[t];
}

/* one get_unitig_forward: seq (length *len, capacity max_len) grows at its end; *median is the contig's
 * median_abundance on entry and on exit; returns the stop reason */
/* branch (may be NULL): at a branch stop, bit x = solid successor fix+x, bit 4+z = solid sibling z+fix (x, z index
 * DNA_bases); ncount (may be NULL, 8 entries): their filter counts -- the contigs the reference queues (:3133-3160) */
int orc_extend_forward(const orc_qf *qf, char *seq, uint32_t *len, unsigned k, uint64_t abundance_min, uint32_t max_len,
                       int *median, uint8_t *branch, uint32_t *ncount) {
  char first[256], cur[256], cur_rc[256], kmer[256];
  if (k >= sizeof(first) || *len < k) return -1;
  memcpy(first, seq, k);
  memcpy(cur, seq + *len - k, k);
  size_t nab = *len - k + 1, cap = nab + (max_len - *len) + 1;
  int *ab = (int *)malloc(cap * sizeof(int));
  for (size_t i = 0; i < nab; i++) ab[i] = *median;
  int stop = 0;
  if (branch) *branch = 0;
  if (ncount) memset(ncount, 0, 8 * sizeof(uint32_t));
  while (!stop) {
    memcpy(cur_rc, cur, k);
    rc_inplace(cur_rc, k);
    int cand_after = 0, cand_before = 0, xa = 0;
    uint64_t count_after[4] = {0, 0, 0, 0}, count_before[4] = {0, 0, 0, 0};
    unsigned mask = 0;
    /* kmers with current_kmer_fix as prefix (:3067-3088) */
    memcpy(kmer, cur + 1, k - 1);
    for (int x = 0; x < 4; x++) {
      kmer[k - 1] = DNA_bases[x];
      uint64_t c = kmer_count(qf, kmer, k);
      if (c >= abundance_min) { cand_after++; count_after[x] = c; xa = x; mask |= 1u << x; }
    }
    /* kmers with RC(current_kmer_fix) as prefix (:3090-3120), except the current k-mer itself */
    memcpy(kmer, cur_rc, k);
    for (int x = 0; x < 4; x++) {
      if (DNA_bases[x] == cur_rc[k - 1]) continue;
      kmer[k - 1] = DNA_bases[x];
      uint64_t c = kmer_count(qf, kmer, k);
      /* RC(fix)+x is the reverse complement of the sibling comp(x)+fix; comp(DNA_bases[x]) = DNA_bases[3-x] */
      if (c >= abundance_min) { cand_before++; count_before[3 - x] = c; mask |= 16u << (3 - x); }
    }
    if (cand_before || cand_after > 1) {                                     /* :3122 */
      stop = ORC_STOP_BRANCH;
      if (branch) *branch = (uint8_t)mask;
      if (ncount) for (int j = 0; j < 4; j++) { ncount[j] = (uint32_t)count_after[j]; ncount[4 + j] = (uint32_t)count_before[j]; }
      break;
    }
    if (cand_after == 0) { stop = ORC_STOP_DEAD_END; break; }               /* :3201 */
    memmove(cur, cur + 1, k - 1);                                            /* :3167-3190 */
    cur[k - 1] = DNA_bases[xa];
    if (memcmp(cur, first, k) == 0) { stop = ORC_STOP_CIRCLE; break; }      /* :3176 */
    if (*len >= max_len) { stop = ORC_STOP_BUFFER; break; }
    seq[(*len)++] = DNA_bases[xa];
    ab[nab++] = (int)count_after[xa];
  }
  *median = median_int(ab, nab);
  free(ab);
  return stop;
}

/* processDataChunk's two extensions of one seed (:1886-1904, the branch where the map has no entry) */
int orc_unitig_from_seed(const orc_qf *qf, const char *seed, uint32_t seed_count, unsigned k, uint64_t abundance_min,
                         uint32_t max_len, char *seq, uint32_t *len, int *median, uint8_t stops[2]) {
  if (max_len < k + 1) return -1;
  memcpy(seq, seed, k);
  *len = k;
  *median = (int)seed_count;
  int s = orc_extend_forward(qf, seq, len, k, abundance_min, max_len, median, NULL, NULL);
  if (s < 0) return s;
  stops[0] = (uint8_t)s;
  rc_inplace(seq, *len);
  s = orc_extend_forward(qf, seq, len, k, abundance_min, max_len, median, NULL, NULL);
  if (s < 0) return s;
  stops[1] = (uint8_t)s;
  return 0;
}
