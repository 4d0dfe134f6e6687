// TEST INFRASTRUCTURE ONLY. The reference's gqf.c under AddressSanitizer (built by `make -C oracle asan`, run by
// tests/test_oracle.py): fills the LAST slots of a filter's overflow tail through the reference's own insert, looks the
// keys up, sweeps and serialises. With the driver's guard blocks (ref_driver.cpp: guard_table) this is clean; with
// REF_NO_GUARD=1 the reference's 8-byte slot access (gqf.c:542-574) runs past its calloc and ASan reports it -- the
// reason the guard exists.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
struct RefQF;
extern "C" {
RefQF *ref_qf_new(uint64_t qb, uint64_t hb, uint32_t seed);
void ref_qf_free(RefQF *h);
int ref_qf_insert(RefQF *h, uint64_t key, uint64_t count);
uint64_t ref_qf_count(RefQF *h, uint64_t key);
uint64_t ref_find_first_empty_slot(RefQF *h, uint64_t from);
uint64_t ref_denoise_round_t1(RefQF *h, uint64_t min_len);
int ref_qf_check_offset(RefQF *h);
}
int main(int argc, char **argv) {
  // qb 10: nslots 1024, xnslots 1344 = 21 blocks exactly, so slot 1343 is the allocation's last byte
  RefQF *q = ref_qf_new(10, 18, 2038074761u);
  uint64_t n = 0;
  for (uint64_t i = 0; i < 442; i++) {              // one cluster from slot 900 to slot 1341
    uint64_t key = ((900 + i / 4) << 8) | ((i % 4) * 60);
    ref_qf_insert(q, key, 1);
    n++;
  }
  uint64_t e = ref_find_first_empty_slot(q, 900);
  if (e != 1342) { fprintf(stderr, "probe: cluster ends at %llu, expected 1342\n", (unsigned long long)e); return 2; }
  ref_qf_insert(q, (900ull << 8) | 0, 1);           // remainder 0 at count 2 takes two more slots (0, 1): the cluster now
                                                    // ends at 1343, the table's last slot
  if (ref_find_first_empty_slot(q, 900) != 1344) { fprintf(stderr, "probe: tail not filled\n"); return 2; }
  for (uint64_t i = 0; i < 442; i++) {
    uint64_t key = ((900 + i / 4) << 8) | ((i % 4) * 60);
    uint64_t want = key == ((900ull << 8) | 0) ? 2 : 1;
    if (ref_qf_count(q, key) != want) { fprintf(stderr, "probe: wrong count\n"); return 3; }
  }
  if (!ref_qf_check_offset(q)) { fprintf(stderr, "probe: check_offset failed\n"); return 4; }
  uint64_t removed = ref_denoise_round_t1(q, 64);
  printf("probe ok: %llu keys, %llu removed\n", (unsigned long long)n, (unsigned long long)removed);
  ref_qf_free(q);
  return 0;
}
