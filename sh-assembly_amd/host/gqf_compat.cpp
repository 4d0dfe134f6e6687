// The gqf-named host surface of include/gqf_compat.h: per-key operations on the packed filter table for callers of the
// reference's `extern "C"` API (cqf/gqf.h:106-225). NOT a transliteration of cqf/gqf.c: the reference keeps runs in place
// by shifting slots under rank/select over saturating block offsets; here everything is derived from one quantity,
// the FREE POINTER (first slot behind the runs of all smaller quotients) -- a block's offset byte is that pointer on
// entering the block, a run starts at max(quotient, free pointer), a slot is empty when the free pointer has not reached
// it -- and a mutation re-lays the runs it displaces from their decoded entries. Since the filter's bytes are a function
// of the key multiset alone the result is the reference's table, byte for byte
// (tests/test_host_logic.py against the compiled reference).
#include "../../include/gqf_compat.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace {

constexpr uint64_t BLK = 89, OCC = 1, RUN = 9, TRAV = 17, SLOTS = 25;   // qfblock, cqf/gqf.c:63-86

struct Tab {
  uint8_t *p;
  uint64_t nslots, xnslots, nblocks;
  explicit Tab(const QF *qf) : p((uint8_t *)qf->blocks), nslots(qf->metadata->nslots), xnslots(qf->metadata->xnslots), nblocks(qf->metadata->nblocks) {}
  uint64_t word(uint64_t b, uint64_t field) const { uint64_t v; memcpy(&v, p + b * BLK + field, 8); return v; }
  void set_word(uint64_t b, uint64_t field, uint64_t v) const { memcpy(p + b * BLK + field, &v, 8); }
  bool bit(uint64_t i, uint64_t field) const { return i / 64 < nblocks && ((word(i / 64, field) >> (i % 64)) & 1); }
  void set_bit(uint64_t i, uint64_t field, bool v) const {
    uint64_t w = word(i / 64, field);
    w = v ? (w | (1ULL << (i % 64))) : (w & ~(1ULL << (i % 64)));
    set_word(i / 64, field, w);
  }
  bool occ(uint64_t q) const { return bit(q, OCC); }
  bool runend(uint64_t s) const { return bit(s, RUN); }
  uint8_t slot(uint64_t s) const { return s / 64 < nblocks ? p[(s / 64) * BLK + SLOTS + s % 64] : 0; }
  void set_slot(uint64_t s, uint8_t v) const { p[(s / 64) * BLK + SLOTS + s % 64] = v; }
  uint8_t off(uint64_t b) const { return p[b * BLK]; }
  // first run-end bit at or behind slot s (xnslots when there is none: a damaged table)
  uint64_t next_runend(uint64_t s) const {
    for (uint64_t b = s / 64; b < nblocks; b++) {
      uint64_t w = word(b, RUN);
      if (b == s / 64) w &= ~0ULL << (s % 64);
      if (w) return b * 64 + (uint64_t)__builtin_ctzll(w);
    }
    return xnslots;
  }
  // free pointer behind the runs of the occupied quotients 64 b .. 64 b + upto - 1, entered with fp
  uint64_t walk(uint64_t b, unsigned upto, uint64_t fp) const {
    uint64_t o = word(b, OCC);
    if (upto < 64) o &= (1ULL << upto) - 1;
    while (o) {
      const uint64_t q = b * 64 + (uint64_t)__builtin_ctzll(o);
      o &= o - 1;
      fp = next_runend(q > fp ? q : fp) + 1;
    }
    return fp;
  }
  // free pointer on entering block b: its offset byte says so unless it is saturated at 255 (gqf.c:580-591, 599-601);
  // then the nearest earlier block with an exact byte is walked forward
  uint64_t fp_enter(uint64_t b) const {
    if (b == 0) return 0;
    if (off(b) < 255) return b * 64 + off(b);
    uint64_t b0 = b - 1;
    while (b0 > 0 && off(b0) == 255) b0--;
    uint64_t fp = b0 == 0 ? 0 : b0 * 64 + off(b0);
    for (uint64_t x = b0; x < b; x++) fp = walk(x, 64, fp);
    return fp;
  }
  uint64_t fp_before(uint64_t q) const { return q / 64 < nblocks ? walk(q / 64, (unsigned)(q % 64), fp_enter(q / 64)) : xnslots; }
  uint64_t run_start(uint64_t q) const { const uint64_t f = fp_before(q); return q > f ? q : f; }
  uint64_t next_occupied(uint64_t after) const {           // smallest occupied quotient > after; xnslots if none
    for (uint64_t b = (after + 1) / 64; b < nblocks; b++) {
      uint64_t w = word(b, OCC);
      if (b == (after + 1) / 64) w &= ~0ULL << ((after + 1) % 64);
      if (w) return b * 64 + (uint64_t)__builtin_ctzll(w);
    }
    return xnslots;
  }
};

// one entry: remainder + the counter code of cqf/gqf.c:1225-1299 (count 1 = the bare remainder; count c + 1 >= 2 = remainder,
// an escape 0 if the top digit exceeds it, the base-128 digits of c with 0x80 on all but the last)
struct Entry { uint8_t rem; uint64_t count; };
uint64_t decode(const Tab &t, uint64_t s, Entry *e) {        // returns the entry's last slot
  e->rem = t.slot(s);
  e->count = 1;
  if (t.runend(s)) return s;
  uint8_t d = t.slot(s + 1);
  if (d > e->rem) return s;
  uint64_t n = 1, c = 0;
  if (d == 0) { n++; d = t.slot(s + n); }
  while (d & 0x80) { c = c * 128 + (d & 0x7f); n++; d = t.slot(s + n); }
  e->count = c * 128 + d + 1;
  return s + n;
}
void encode(const Entry &e, std::vector<uint8_t> *out) {
  out->push_back(e.rem);
  if (e.count <= 1) return;
  uint64_t c = e.count - 1;
  uint8_t dig[12];
  int nd = 0;
  do { dig[nd++] = (uint8_t)(c & 0x7f); c >>= 7; } while (c);
  uint8_t top = dig[nd - 1] | (nd > 1 ? 0x80 : 0);
  if (top > e.rem) out->push_back(0);
  for (int i = nd - 1; i >= 1; i--) out->push_back(dig[i] | 0x80);
  out->push_back(dig[0]);
}
// the entries of quotient q's run
void run_entries(const Tab &t, uint64_t q, std::vector<Entry> *out, uint64_t *start, uint64_t *end) {
  uint64_t s = t.run_start(q);
  *start = s;
  for (;;) {
    Entry e;
    const uint64_t last = decode(t, s, &e);
    out->push_back(e);
    if (t.runend(last) || last + 1 >= t.xnslots) { *end = last; return; }
    s = last + 1;
  }
}
bool find_entry(const Tab &t, uint64_t key, uint64_t *first_slot, uint64_t *count) {
  const uint64_t q = key >> 8, r = key & 0xff;
  if (q >= t.nslots || !t.occ(q)) return false;
  uint64_t s = t.run_start(q);
  for (;;) {
    Entry e;
    const uint64_t last = decode(t, s, &e);
    if (e.rem == r) { *first_slot = s; *count = e.count; return true; }
    if (t.runend(last) || last + 1 >= t.xnslots) return false;
    s = last + 1;
  }
}

struct Placed { uint64_t q, start; std::vector<uint8_t> bytes; };
// write re-laid runs and the offset bytes of the blocks their span crosses
void write_runs(const Tab &t, const std::vector<Placed> &runs, uint64_t old_lo, uint64_t old_hi) {
  for (uint64_t s = old_lo; s < old_hi; s++) { t.set_bit(s, RUN, false); t.set_slot(s, 0); }
  for (const Placed &r : runs) {
    for (size_t i = 0; i < r.bytes.size(); i++) t.set_slot(r.start + i, r.bytes[i]);
    if (!r.bytes.empty()) t.set_bit(r.start + r.bytes.size() - 1, RUN, true);
  }
}
void fix_offsets(const Tab &t, uint64_t b_from, uint64_t b_to) {     // blocks b_from .. b_to inclusive, from their predecessors
  for (uint64_t b = b_from; b <= b_to && b < t.nblocks; b++) {
    if (b == 0) continue;
    // strictly from the layout (block_offset_strict, gqf.c:599-601): walk block b - 1 from ITS entering pointer
    const uint64_t fp = t.walk(b - 1, 64, t.fp_enter(b - 1));
    const uint64_t o = fp > b * 64 ? fp - b * 64 : 0;
    t.p[b * BLK] = (uint8_t)(o > 255 ? 255 : o);
  }
}

bool insert_count(QF *qf, uint64_t key, uint64_t count, bool *is_new) {
  const Tab t(qf);
  const uint64_t q = key >> 8;
  const uint8_t r = (uint8_t)(key & 0xff);
  *is_new = false;
  if (count == 0) return true;
  if (q >= t.nslots) return false;
  // q's run with the key folded in
  std::vector<Entry> ents;
  uint64_t fp_old = t.fp_before(q), old_lo = q > fp_old ? q : fp_old, old_hi = old_lo;
  if (t.occ(q)) { uint64_t s, e; run_entries(t, q, &ents, &s, &e); old_hi = e + 1; }
  size_t i = 0;
  while (i < ents.size() && ents[i].rem < r) i++;
  if (i < ents.size() && ents[i].rem == r) ents[i].count += count;
  else { ents.insert(ents.begin() + (long)i, Entry{r, count}); *is_new = true; }
  std::vector<Placed> runs;
  runs.push_back(Placed{q, old_lo, {}});
  for (const Entry &e : ents) encode(e, &runs.back().bytes);
  uint64_t fp_new = old_lo + runs.back().bytes.size();
  uint64_t fp_was = t.occ(q) ? old_hi : fp_old;          // the free pointer behind q before the insert
  // the runs this one displaces: a later run moves iff the new free pointer has passed its old start
  for (uint64_t p = t.next_occupied(q); p < t.xnslots; p = t.next_occupied(p)) {
    const uint64_t was = p > fp_was ? p : fp_was;
    const uint64_t now = p > fp_new ? p : fp_new;
    if (now == was) break;
    const uint64_t e = t.next_runend(was);
    Placed pr{p, now, {}};
    for (uint64_t s = was; s <= e; s++) pr.bytes.push_back(t.slot(s));
    runs.push_back(std::move(pr));
    fp_was = e + 1;
    fp_new = now + (e - was + 1);
    old_hi = e + 1;
  }
  if (fp_new > t.xnslots) { *is_new = false; return false; }   // full: the reference would run off its table here
  write_runs(t, runs, old_lo, old_hi);
  t.set_bit(q, OCC, true);
  fix_offsets(t, q / 64 + 1, (fp_new - 1) / 64 + 1);
  return true;
}

}  // namespace

extern "C" {

void qf_init(QF *qf, uint64_t nslots, uint64_t key_bits, uint64_t value_bits, bool mem, const char *, uint32_t seed) {
  uint64_t lg = 0;
  while ((1ULL << lg) < nslots) lg++;
  if (!mem || value_bits != 0 || (1ULL << lg) != nslots || key_bits != lg + 8) {
    fprintf(stderr, "gqf_compat: only in-memory filters with value_bits = 0 and key_bits = log2(nslots) + 8 are supported\n");
    exit(EXIT_FAILURE);
  }
  qf->mem = (qfmem *)calloc(1, sizeof(qfmem));
  qfmetadata *m = qf->metadata = (qfmetadata *)calloc(1, sizeof(qfmetadata));
  m->seed = seed;
  m->nslots = nslots;
  m->xnslots = nslots + (uint64_t)(10 * sqrt((double)nslots));
  m->key_bits = key_bits; m->value_bits = 0; m->key_remainder_bits = 8; m->bits_per_slot = 8;
  m->range = (__uint128_t)nslots << 8;
  m->nblocks = (m->xnslots + 63) / 64;
  m->size = m->nblocks * BLK;
  m->num_locks = m->xnslots / (1ULL << 16) + 2;
  qf->blocks = calloc(m->size + 2 * BLK, 1);
  qf->mem->locks = (volatile int *)calloc(m->num_locks, sizeof(int));
}
void qf_reset(QF *qf) {
  qf->metadata->nelts = qf->metadata->ndistinct_elts = qf->metadata->noccupied_slots = 0;
  memset(qf->blocks, 0, qf->metadata->size);
}
void qf_destroy(QF *qf, bool) {
  if (qf->mem) free((void *)qf->mem->locks);
  free(qf->mem); free(qf->metadata); free(qf->blocks);
  qf->mem = nullptr; qf->metadata = nullptr; qf->blocks = nullptr;
}
bool qf_insert(QF *qf, uint64_t key, uint64_t, uint64_t count, bool, bool) {
  bool nw;
  return insert_count(qf, key, count, &nw);
}
bool qf_insert_advance(QF *qf, uint64_t key, uint64_t, uint64_t count, bool, bool, bool &isNew) {
  return insert_count(qf, key, count, &isNew);
}
uint64_t qf_count_key_value(const QF *qf, uint64_t key, uint64_t) {
  uint64_t s, c;
  return find_entry(Tab(qf), key, &s, &c) ? c : 0;
}
bool qf_is_traveled(const QF *qf, uint64_t index) { return Tab(qf).bit(index, TRAV); }
void qf_set_traveled(const QF *qf, uint64_t index) { Tab(qf).set_bit(index, TRAV, true); }
bool qf_count_key_value_set_traveled(const QF *qf, uint64_t key, uint64_t, uint64_t *count) {
  const Tab t(qf);
  uint64_t s, c;
  if (!find_entry(t, key, &s, &c)) { *count = 0; return false; }
  *count = c;
  if (t.bit(s, TRAV)) return true;
  t.set_bit(s, TRAV, true);
  return false;
}
bool qf_count_key_value_is_traveled(const QF *qf, uint64_t key, uint64_t, uint64_t *count) {
  const Tab t(qf);
  uint64_t s, c;
  if (!find_entry(t, key, &s, &c)) { *count = 0; return false; }
  *count = c;
  return t.bit(s, TRAV);
}

uint64_t find_first_empty_slot(const QF *qf, uint64_t from) {
  const Tab t(qf);
  uint64_t x = from;
  while (x < t.xnslots) {
    // behind the runs of all quotients <= x, worked out inside x's own block (the entering pointer of the NEXT block
    // is only known as max(its first slot, pointer))
    const uint64_t fp = t.walk(x / 64, (unsigned)(x % 64) + 1, t.fp_enter(x / 64));
    if (fp <= x) return x;
    x = fp;
  }
  return x;
}
uint64_t find_first_nonempty_slot(const QF *qf, uint64_t from) {   // (the next OCCUPIED quotient, as in the reference)
  const Tab t(qf);
  if (t.occ(from)) return from;
  const uint64_t q = t.next_occupied(from);
  return q < t.xnslots ? q : t.xnslots;
}

void qf_clean_singleton(const QF *qf, uint64_t start_bucket_id, uint64_t end_bucket_id, uint64_t *removed_elts) {
  const Tab t(qf);
  std::vector<Placed> runs;
  std::vector<uint64_t> emptied;
  uint64_t fp = t.fp_before(start_bucket_id), old_lo = t.xnslots, old_hi = 0, removed = 0;
  uint64_t fp_new = fp;
  for (uint64_t q = t.occ(start_bucket_id) ? start_bucket_id : t.next_occupied(start_bucket_id); q <= end_bucket_id && q < t.xnslots;
       q = t.next_occupied(q)) {
    const uint64_t s = q > fp ? q : fp;
    std::vector<Entry> ents;
    uint64_t s2, e;
    // (run_entries recomputes the start from the table, which this loop has not changed yet)
    run_entries(t, q, &ents, &s2, &e);
    (void)s;
    old_lo = std::min(old_lo, s2);
    old_hi = std::max(old_hi, e + 1);
    fp = e + 1;
    Placed pr{q, q > fp_new ? q : fp_new, {}};
    for (const Entry &en : ents) { if (en.count >= 2) encode(en, &pr.bytes); else removed++; }
    if (pr.bytes.empty()) emptied.push_back(q);          // (the table is read until the loop is over: nothing changes before)
    else { fp_new = pr.start + pr.bytes.size(); runs.push_back(std::move(pr)); }
  }
  for (uint64_t q : emptied) t.set_bit(q, OCC, false);
  if (old_hi > old_lo) {
    write_runs(t, runs, old_lo, old_hi);
    fix_offsets(t, start_bucket_id / 64 + 1, (old_hi - 1) / 64 + 1);
  }
  if (removed_elts) *removed_elts += removed;
}

uint64_t popcnt_runends(const QF *qf) {
  const Tab t(qf);
  uint64_t n = 0;
  for (uint64_t b = 0; b < t.nblocks; b++) n += (uint64_t)__builtin_popcountll(t.word(b, RUN));
  return n;
}
uint64_t popcnt_occupieds(const QF *qf) {
  const Tab t(qf);
  uint64_t n = 0;
  for (uint64_t b = 0; b < t.nblocks; b++) n += (uint64_t)__builtin_popcountll(t.word(b, OCC));
  return n;
}
bool check_offset(const QF *qf) {
  const Tab t(qf);
  uint64_t fp = 0;
  for (uint64_t b = 1; b < t.nblocks; b++) {
    fp = t.walk(b - 1, 64, fp);
    const uint64_t real = fp > b * 64 ? fp - b * 64 : 0;
    if (real <= 255 && t.off(b) != real) return false;
    if (real > 255 && t.off(b) != 255) return false;
  }
  return true;
}

bool qf_iterator(QF *qf, QFi *qfi, uint64_t position) {
  const Tab t(qf);
  if (!t.occ(position)) position = t.next_occupied(position);   // (the reference indexes a block with `position` here, gqf.c:2478: only 0 is safe there)
  memset(qfi, 0, sizeof(*qfi));
  qfi->qf = qf;
  qfi->run = position;
  qfi->current = position < t.xnslots ? t.run_start(position) : t.xnslots;
  return qfi->current < t.nslots;
}
int qfi_end(QFi *qfi) { return qfi->current >= qfi->qf->metadata->xnslots ? 1 : 0; }
int qfi_get(QFi *qfi, uint64_t *key, uint64_t *value, uint64_t *count) {
  if (qfi_end(qfi)) return 1;
  Entry e;
  decode(Tab(qfi->qf), qfi->current, &e);
  *key = (qfi->run << 8) | e.rem;
  *value = 0;
  *count = e.count;
  return 0;
}
int qfi_next(QFi *qfi) {
  if (qfi_end(qfi)) return 1;
  const Tab t(qfi->qf);
  Entry e;
  const uint64_t last = decode(t, qfi->current, &e);
  if (!t.runend(last)) {
    qfi->current = last + 1;
    return qfi->current > t.nslots ? 1 : 0;                // (sic: the reference compares with nslots, gqf.c:2543)
  }
  const uint64_t nq = t.next_occupied(qfi->run);
  if (nq >= t.xnslots) { qfi->run = qfi->current = t.xnslots; return 1; }
  qfi->run = nq;
  qfi->current = last + 1 < nq ? nq : last + 1;
  return 0;
}
int qfi_next_untraveled(QFi *qfi) {
  int end = qfi_next(qfi);
  while (!end && qf_is_traveled(qfi->qf, qfi->current)) end = qfi_next(qfi);
  return end;
}

void qf_serialize(const QF *qf, const char *filename) {
  FILE *f = fopen(filename, "wb+");
  if (!f) { perror("Error opening file for serializing\n"); exit(EXIT_FAILURE); }
  fwrite(qf->metadata, sizeof(qfmetadata), 1, f);
  fwrite(qf->blocks, qf->metadata->size, 1, f);
  fclose(f);
}
void qf_deserialize(QF *qf, const char *filename) {
  FILE *f = fopen(filename, "rb");
  if (!f) { perror("Error opening file for deserializing\n"); exit(EXIT_FAILURE); }
  qf->mem = (qfmem *)calloc(1, sizeof(qfmem));
  qf->metadata = (qfmetadata *)calloc(1, sizeof(qfmetadata));
  if (fread(qf->metadata, sizeof(qfmetadata), 1, f) != 1 || qf->metadata->bits_per_slot != 8) {
    fprintf(stderr, "gqf_compat: %s is not a .cqf with 8-bit slots\n", filename);
    exit(EXIT_FAILURE);
  }
  qf->metadata->num_locks = qf->metadata->xnslots / (1ULL << 16) + 2;
  qf->mem->locks = (volatile int *)calloc(qf->metadata->num_locks, sizeof(int));
  qf->blocks = calloc(qf->metadata->size + 2 * BLK, 1);
  if (fread(qf->blocks, qf->metadata->size, 1, f) != 1) { fprintf(stderr, "gqf_compat: %s is truncated\n", filename); exit(EXIT_FAILURE); }
  fclose(f);
}

}  // extern "C"
