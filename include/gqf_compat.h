/* The gqf-named host surface: the subset of the reference's `extern "C"` filter API (cqf/gqf.h:106-225) that
 * its two programs and their helpers reach, with the reference's names, signatures and struct layouts, implemented in
 * sh-assembly_amd/host/gqf_compat.cpp over the same packed table (128-byte qfmetadata + 89-byte qfblocks at
 * bits_per_slot = 8) and exported from sh-assembly_amd/libshkhost.so. A reference-side caller that wants PER-KEY
 * semantics next to the batched device ABI of include/shk.h (import a .cqf the GPU wrote, look keys up, iterate, patch a
 * few counts, sweep a cluster, write it back) links these unchanged. Host code, one thread at a time per filter: the
 * `lock` / `spin` arguments are accepted and ignored (the device path has no per-key locks).
 *
 * Only the geometry CQF-deNoise uses is supported: value_bits = 0 and key_bits = log2(nslots) + 8 (cqf/CQF_mt.h:442);
 * anything else ends the program like the reference's own fatal paths do (perror + exit, gqf.c:2244-2247).
 * Not provided (not reached from either program): qf_copy, qf_read, qf_merge, qf_multi_merge,
 * qf_inner_product, qf_magnitude, the *_with_lock sweeps, and the prototypes gqf.h declares but gqf.c never defines. */
#ifndef GQF_COMPAT_H
#define GQF_COMPAT_H
#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct quotient_filter_mem {            /* cqf/gqf.h:53-58 */
  int fd;
  volatile int metadata_lock;
  volatile int *locks;
  void *wait_times;
} qfmem;

typedef struct quotient_filter_metadata {       /* cqf/gqf.h:62-77: 128 bytes, written as the .cqf header */
  uint64_t size;
  uint32_t seed;
  uint64_t nslots;
  uint64_t xnslots;
  uint64_t key_bits;
  uint64_t value_bits;
  uint64_t key_remainder_bits;
  uint64_t bits_per_slot;
  __uint128_t range;
  uint64_t nblocks;
  uint64_t nelts;
  uint64_t ndistinct_elts;
  uint64_t noccupied_slots;
  uint64_t num_locks;
} qfmetadata;

typedef struct quotient_filter {                /* cqf/gqf.h:81-85 */
  qfmem *mem;
  qfmetadata *metadata;
  void *blocks;                                 /* nblocks packed 89-byte blocks (cqf/gqf.c:63-86) */
} QF;

typedef struct { uint64_t start_index; uint16_t length; } cluster_data;   /* cqf/gqf.h:88-91 */
typedef struct quotient_filter_iterator {       /* cqf/gqf.h:93-101 */
  QF *qf;
  uint64_t run;
  uint64_t current;
  uint64_t cur_start_index;
  uint16_t cur_length;
  uint32_t num_clusters;
  cluster_data *c_info;
} QFi;

void qf_init(QF *qf, uint64_t nslots, uint64_t key_bits, uint64_t value_bits, bool mem, const char *path, uint32_t seed); /* gqf.c:2187 */
void qf_reset(QF *qf);                                                                   /* :2360 */
void qf_destroy(QF *qf, bool mem);                                                       /* :2306 */
bool qf_insert(QF *qf, uint64_t key, uint64_t value, uint64_t count, bool lock, bool spin); /* :2422 */
#ifdef __cplusplus
bool qf_insert_advance(QF *qf, uint64_t key, uint64_t value, uint64_t count, bool lock, bool spin, bool &isNew); /* :2432 (a C++ reference in the reference too) */
#endif
uint64_t qf_count_key_value(const QF *qf, uint64_t key, uint64_t value);                 /* :2442 */
bool qf_iterator(QF *qf, QFi *qfi, uint64_t position);                                   /* :2474 */
int qfi_get(QFi *qfi, uint64_t *key, uint64_t *value, uint64_t *count);                  /* :2506: 0 = valid */
int qfi_next(QFi *qfi);                                                                  /* :2529 */
int qfi_end(QFi *qfi);                                                                   /* :2593 */
void qf_serialize(const QF *qf, const char *filename);                                   /* :2379 */
void qf_deserialize(QF *qf, const char *filename);                                       /* :2396 */
uint64_t find_first_empty_slot(const QF *qf, uint64_t from);                             /* :738 */
uint64_t find_first_nonempty_slot(const QF *qf, uint64_t from);                          /* :751 */
void qf_clean_singleton(const QF *qf, uint64_t start_bucket_id, uint64_t end_bucket_id, uint64_t *removed_elts); /* :2792 */
uint64_t popcnt_runends(const QF *qf);                                                   /* :3042 */
uint64_t popcnt_occupieds(const QF *qf);                                                 /* :3049 */
bool check_offset(const QF *qf);                                                         /* :3056 */
bool qf_is_traveled(const QF *qf, uint64_t index);                                       /* :3071 */
void qf_set_traveled(const QF *qf, uint64_t index);                                      /* :3075 */
int qfi_next_untraveled(QFi *qfi);                                                       /* :3082 */
bool qf_count_key_value_set_traveled(const QF *qf, uint64_t key, uint64_t value, uint64_t *count);  /* :3092: returns WAS traveled */
bool qf_count_key_value_is_traveled(const QF *qf, uint64_t key, uint64_t value, uint64_t *count);   /* :3132 */

#ifdef __cplusplus
}
#endif
#endif
