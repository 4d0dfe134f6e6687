// Contiger on the device: the unitig set, its start-k-mer map, the walk with the reference's
// stop rules, duplicate removal, numbering, the graph pass and the sequence text all live in HBM; the host only
// launches rounds and writes the final FASTA text.
//
// Reference (src/contig_assembly.cpp)                       here
//   contigs (concurrent_vector<Contig>)                     ShkUG contig arrays (first/last k-mer as 2 bits per base in
//                                                           128 bits, rolling hashes, length, state)
//   startKmer2unitig (tbb::concurrent_hash_map)  :3018-3025 open-addressing table keyed by the packed k-mer, value = contig id,
//                                                           "smaller id wins" as an atomic minimum (ug_put)
//   WorkQueue of branch neighbours               :847-882   `next` list filled by the walk kernel, swapped by the host per round
//   get_unitig_forward                           :3028-3218 k_ug_walk: per step 4 successor + 3 sibling lookups that mark the
//                                                           traveled bit, known-node test (traveled AND in the map), stop /
//                                                           extend / pure circle exactly as the reference decides
//   seeds' second call on the reverse complement :1886-1904 in the same kernel, as soon as the first call has closed
//   check_unitig                                 :935-954   k_ug_check (first k-mer and RC(last k-mer) must map to the contig)
//   track_kmer_worker                            :956-1010  k_ug_number + k_ug_map2 (first k-mer -> +id, RC(last) -> -id)
//   build_graph_worker                           :1012-1084 k_ug_links
//   Contig::median_abundance, median()           Utility.cpp:27-40   k_ug_median (two-stage, as the seeds' two calls compute it)
//   writer                                       :600-629   k_ug_emit writes the bases; the host formats the records
//
// A walk stores nothing but its end state: the bases of the kept unitigs are produced afterwards by walking each once
// more from its first k-mer into an exactly sized arena (the path is unambiguous: a walk only ever extends through a
// k-mer with one solid successor that has one solid predecessor). Sequences never travel to the host before the end.
#include "shk_device.h"

typedef unsigned __int128 shk_u128;

#define SHK_UG_UNUSED 0
#define SHK_UG_OPEN 1
#define SHK_UG_CLOSED 2
#define SHK_UG_CLEARED 3
#define SHK_UG_BUSY 0xFFFFFFFFu
#define SHK_UG_E_CONTIGS 1u      // contig arrays full (host grows them before a round: cannot happen unless miscounted)
#define SHK_UG_E_MAP 2u          // start-k-mer map full

struct ShkUG {
  // contigs (index = contig id, ids start at 1 like the reference's contigs.resize(1))
  uint64_t *first_lo, *first_hi, *cur_lo, *cur_hi, *rc_lo, *rc_hi, *fh, *rh, *hmin;
  uint32_t *len, *l1, *cnt0;
  uint8_t *state, *kind, *stop;      // kind: bit 0 = seed (walked both ways), bit 1 = second call running, bit 2 = hashes not yet computed,
                                     // bit 3 + bits 4-5 = the base onto RC(seed) (k_ug_emit), bit 6 = a pure circle (the circle set decides who keeps it)
  uint32_t cap;
  uint32_t *ncontigs;                // next free id
  // start-k-mer map
  uint64_t *mk_lo, *mk_hi;
  uint32_t *mv;
  uint32_t mmask;
  // pure circles: minimum canonical hash over the circle's k-mers -> contig id (the same circle cut elsewhere collides)
  uint64_t *ck;
  uint32_t *cv;
  uint32_t cmask;
  // work lists
  uint32_t *next, *next_n;
  uint32_t *flags;
  unsigned long long *stats;         // 0 extensions, 1 duplicates (cleared), 2 truncated, 3 candidates queued
};

__device__ __forceinline__ uint64_t shk_ug_mix(uint64_t lo, uint64_t hi) {
  uint64_t h = lo * 0x9E3779B97F4A7C15ULL ^ (hi + 0x7F4A7C159E3779B9ULL) * 0xC2B2AE3D27D4EB4FULL;
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
  return h;
}
// reverse complement of a k-mer packed 2 bits per base, first base in the highest used bits
__device__ __forceinline__ shk_u128 shk_ug_rc(shk_u128 w, uint32_t k) {
  uint64_t lo = ~(uint64_t)w, hi = ~(uint64_t)(w >> 64);
  // reverse the 2-bit groups of each half, then swap the halves
#define SHK_REV2(x) do { x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2); \
                         x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4); \
                         x = __builtin_bswap64(x); } while (0)
  SHK_REV2(lo); SHK_REV2(hi);
#undef SHK_REV2
  const shk_u128 r = ((shk_u128)lo << 64) | hi;
  return k == 64 ? r : (r >> (128 - 2 * k));
}
__device__ __forceinline__ shk_u128 shk_ug_mask(uint32_t k) { return k == 64 ? ~(shk_u128)0 : (((shk_u128)1 << (2 * k)) - 1); }

// value of `key` in the start-k-mer map, 0 when absent
__device__ __forceinline__ uint32_t shk_ug_find(const ShkUG &G, shk_u128 key) {
  const uint64_t lo = (uint64_t)key, hi = (uint64_t)(key >> 64);
  uint32_t i = (uint32_t)shk_ug_mix(lo, hi) & G.mmask;
  for (uint32_t probes = 0; probes <= G.mmask; probes++) {
    const uint32_t v = __hip_atomic_load(&G.mv[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    // a slot that is being filled right now counts as the end of the probe sequence: every key inserted BEFORE it was
    // claimed sits at or in front of it, so "absent" is a correct answer for this moment
    if (v == 0 || v == SHK_UG_BUSY) return 0;
    if (G.mk_lo[i] == lo && G.mk_hi[i] == hi) return v;
    i = (i + 1) & G.mmask;
  }
  return 0;
}
// insert_or_replace (contig_assembly.cpp:3018-3025): the key ends up mapped to min(existing, id); true when that is `id`.
// only_if_absent: a plain insert that reports whether the key was new (the work queue's test, :3137, :3150).
__device__ __forceinline__ bool shk_ug_put(const ShkUG &G, shk_u128 key, uint32_t id, bool only_if_absent) {
  const uint64_t lo = (uint64_t)key, hi = (uint64_t)(key >> 64);
  uint32_t i = (uint32_t)shk_ug_mix(lo, hi) & G.mmask;
  for (uint32_t probes = 0; probes <= G.mmask;) {
    const uint32_t v = __hip_atomic_load(&G.mv[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if (v == 0) {
      if (atomicCAS(&G.mv[i], 0u, SHK_UG_BUSY) == 0u) {       // claimed: key first, then the value makes it visible
        G.mk_lo[i] = lo; G.mk_hi[i] = hi;
        __hip_atomic_store(&G.mv[i], id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return true;
      }
      continue;                                               // somebody else claimed it: look again
    }
    if (v == SHK_UG_BUSY) continue;                           // its owner publishes within its own loop iteration
    if (G.mk_lo[i] == lo && G.mk_hi[i] == hi) {
      if (only_if_absent) return false;
      uint32_t old = v;
      while (old > id) {
        const uint32_t prev = atomicCAS(&G.mv[i], old, id);
        if (prev == old) return true;
        old = prev;
      }
      return old == id;
    }
    i = (i + 1) & G.mmask;
    probes++;
  }
  atomicOr(G.flags, SHK_UG_E_MAP);
  return false;
}
__device__ __forceinline__ bool shk_ug_circle_put(const ShkUG &G, uint64_t h, uint32_t id) {
  uint32_t i = (uint32_t)(h * 0x9E3779B97F4A7C15ULL >> 32) & G.cmask;
  for (uint32_t probes = 0; probes <= G.cmask;) {
    const uint32_t v = __hip_atomic_load(&G.cv[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if (v == 0) {
      if (atomicCAS(&G.cv[i], 0u, SHK_UG_BUSY) == 0u) {
        G.ck[i] = h;
        __hip_atomic_store(&G.cv[i], id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return true;
      }
      continue;
    }
    if (v == SHK_UG_BUSY) continue;
    if (G.ck[i] == h) {
      uint32_t old = v;
      while (old > id) {
        const uint32_t prev = atomicCAS(&G.cv[i], old, id);
        if (prev == old) return true;
        old = prev;
      }
      return old == id;
    }
    i = (i + 1) & G.cmask;
    probes++;
  }
  atomicOr(G.flags, SHK_UG_E_MAP);
  return false;
}
__device__ __forceinline__ uint32_t shk_ug_circle_find(const ShkUG &G, uint64_t h) {
  uint32_t i = (uint32_t)(h * 0x9E3779B97F4A7C15ULL >> 32) & G.cmask;
  for (uint32_t probes = 0; probes <= G.cmask; probes++) {
    const uint32_t v = G.cv[i];
    if (v == 0) return 0;
    if (v != SHK_UG_BUSY && G.ck[i] == h) return v;
    i = (i + 1) & G.cmask;
  }
  return 0;
}

// canonical ntHash of a packed k-mer from scratch (base/nthash.hpp:295-302)
__device__ __forceinline__ void shk_ug_hash(shk_u128 w, uint32_t k, uint64_t *fh, uint64_t *rh) {
  uint64_t f = 0, r = 0;
  for (uint32_t j = 0; j < k; j++) {
    const unsigned c = (unsigned)(w >> (2 * (k - 1 - j))) & 3u;
    f ^= shk_rol64(shk_code_seed(c), (k - 1 - j) & 63);
    r ^= shk_rol64(shk_code_seed_rc(c), j & 63);
  }
  *fh = f; *rh = r;
}

// a new contig that consists of one k-mer (Contig(kmer, count), :3139, :3152, :1878)
__device__ __forceinline__ void shk_ug_init(const ShkUG &G, uint32_t id, shk_u128 kmer, uint32_t k, uint32_t count, uint8_t kind) {
  const shk_u128 rc = shk_ug_rc(kmer, k);
  G.first_lo[id] = (uint64_t)kmer; G.first_hi[id] = (uint64_t)(kmer >> 64);
  G.cur_lo[id] = (uint64_t)kmer; G.cur_hi[id] = (uint64_t)(kmer >> 64);
  G.rc_lo[id] = (uint64_t)rc; G.rc_hi[id] = (uint64_t)(rc >> 64);
  G.fh[id] = 0; G.rh[id] = 0; G.hmin[id] = ~0ULL;
  G.len[id] = k; G.l1[id] = k; G.cnt0[id] = count;
  G.kind[id] = (uint8_t)(kind | 4u);
  G.stop[id] = 0;
  G.state[id] = SHK_UG_OPEN;
}

// seeds given as text (n * k upper-case bases): contigs first_id .. first_id + n - 1, in order; a seed with a byte that
// is not a base gives an unused contig
__global__ void k_ug_add_seeds(ShkUG G, const char *seeds, const uint32_t *counts, uint32_t n, uint32_t k, uint32_t first_id,
                               uint32_t *active) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t id = first_id + i;
  shk_u128 w = 0;
  bool bad = false;
  for (uint32_t j = 0; j < k; j++) {
    const char c = seeds[(size_t)i * k + j];
    const unsigned cc = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
    if (cc > 3) bad = true;
    w = (w << 2) | (cc & 3u);
  }
  active[i] = id;
  if (bad) { G.state[id] = SHK_UG_UNUSED; G.len[id] = 0; return; }
  shk_ug_init(G, id, w, k, counts[i], 1);
}
// seeds straight from the reads of a batch (k_select_seeds' per-read output: count 0 = no seed): compacted into new contigs
__global__ void k_ug_seeds_from_reads(ShkUG G, const char *seeds, const uint32_t *counts, uint64_t read_lo, uint64_t nreads, uint32_t k,
                                      uint32_t *active, uint32_t *nactive) {
  const uint64_t r = read_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nreads || counts[r] == 0) return;
  shk_u128 w = 0;
  for (uint32_t j = 0; j < k; j++) {
    const char c = seeds[r * k + j];
    w = (w << 2) | (c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : 3u);
  }
  const uint32_t id = atomicAdd(G.ncontigs, 1u);
  if (id >= G.cap) { atomicOr(G.flags, SHK_UG_E_CONTIGS); return; }
  shk_ug_init(G, id, w, k, counts[r], 1);
  active[atomicAdd(nactive, 1u)] = id;
}

// get_unitig_forward for every contig of `active`, at most max_steps extensions each per launch (a contig that is not
// done goes onto the `next` list again, as do the neighbours queued at a branch and seeds that turn round).
// EIGHT LANES PER CONTIG: a step's seven filter lookups are independent of each other, and a walk is one long chain
// of dependent steps, so lane j of a group looks up neighbour j (0-3: the successors current[1..]+ACGT, 4-7: the
// siblings, i.e. RC(current) with its last base replaced by ACGT; the lane of current itself idles). The group's
// verdicts meet in a wave ballot; every lane keeps the walk's state and decides alike; lane 0 owns the contig's
// records, a lane that found a candidate queues it itself.
#define SHK_UG_LANES 8
__global__ void __launch_bounds__(SHK_WAVE) k_ug_walk(ShkUG G, const uint32_t *active, uint32_t nactive, uint8_t *tab, uint64_t q_lo,
                                                      uint64_t nslots, uint32_t hb, uint32_t k, uint64_t amin, int mark, uint32_t max_steps,
                                                      uint32_t max_len) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t grp = t / SHK_UG_LANES;
  const unsigned sub = t % SHK_UG_LANES;                       // my neighbour
  const unsigned gshift = (threadIdx.x & (SHK_WAVE - 1)) & ~(SHK_UG_LANES - 1u);   // my group's first lane in the wave
  bool alive = grp < nactive;
  const uint32_t id = alive ? active[grp] : 0;
  if (alive && G.state[id] != SHK_UG_OPEN) alive = false;
  const uint64_t kmask = hb >= 64 ? ~0ULL : ((1ULL << hb) - 1);
  const shk_u128 wmask = shk_ug_mask(k);
  shk_u128 win = 0, rcw = 0, first = 0;
  uint64_t fh = 0, rh = 0, hmin = ~0ULL;
  uint32_t len = 0;
  uint8_t kind = 0;
  if (alive) {
    win = ((shk_u128)G.cur_hi[id] << 64) | G.cur_lo[id];
    rcw = ((shk_u128)G.rc_hi[id] << 64) | G.rc_lo[id];
    first = ((shk_u128)G.first_hi[id] << 64) | G.first_lo[id];
    fh = G.fh[id]; rh = G.rh[id]; hmin = G.hmin[id];
    len = G.len[id]; kind = G.kind[id];
  }
  unsigned long long ext = 0;
  uint32_t steps = 0;
  while (__ballot(alive)) {                                    // wave-uniform: groups that are done idle along
    bool solid = false, node = false;
    uint32_t mycnt = 0;
    unsigned s0 = 0;
    if (alive) {
      if (kind & 4u) {                                         // a fresh k-mer: hash it once
        shk_ug_hash(win, k, &fh, &rh);
        kind &= ~4u;
        const uint64_t hc = fh < rh ? fh : rh;
        if (hc < hmin) hmin = hc;
      }
      if (steps >= max_steps) {                                // goes on in the next launch
        if (sub == 0) {
          G.cur_lo[id] = (uint64_t)win; G.cur_hi[id] = (uint64_t)(win >> 64);
          G.rc_lo[id] = (uint64_t)rcw; G.rc_hi[id] = (uint64_t)(rcw >> 64);
          G.fh[id] = fh; G.rh[id] = rh; G.hmin[id] = hmin; G.len[id] = len; G.kind[id] = kind;
          G.next[atomicAdd(G.next_n, 1u)] = id;
        }
        alive = false;
      }
    }
    if (alive) {
      steps++;
      s0 = (unsigned)(win >> (2 * (k - 1))) & 3u;
      uint64_t f, r;
      shk_u128 key;
      bool mine = true;
      if (sub < 4) {
        // k-mers with current[1..] as prefix (:3064-3087)
        f = shk_rol64(fh, 1) ^ shk_rol64(shk_code_seed(s0), k & 63) ^ shk_code_seed(sub);
        r = shk_ror64(rh ^ shk_code_seed_rc(s0), 1) ^ shk_rol64(shk_code_seed_rc(sub), (k - 1) & 63);
        key = ((win << 2) | sub) & wmask;
      } else {
        // k-mers with RC(current[1..]) as prefix: the other predecessors of my successors (:3090-3120). In the reverse
        // orientation they are RC(current) with its last base replaced; base y there <=> sibling base 3 - y here
        const unsigned y = sub - 4, z = 3u - y;
        mine = y != 3u - s0;
        f = fh ^ shk_rol64(shk_code_seed(s0) ^ shk_code_seed(z), (k - 1) & 63);
        r = rh ^ shk_code_seed_rc(s0) ^ shk_code_seed_rc(z);
        key = (rcw & ~(shk_u128)3) | y;
      }
      if (mine) {
        uint8_t trav;
        const uint64_t cnt = shk_lookup_one(tab, (f < r ? f : r) & kmask, q_lo, nslots, mark, &trav);
        if (cnt >= amin) {
          solid = true;
          mycnt = cnt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cnt;
          node = trav && shk_ug_find(G, key) != 0;             // a known unitig start
        }
      }
    }
    const unsigned solid_m = (unsigned)(__ballot(solid) >> gshift) & 0xFFu;
    const unsigned node_m = (unsigned)(__ballot(node) >> gshift) & 0xFFu;
    // ---- the verdict (every lane of the group computes the same); lane 0 registers what has to be registered
    uint8_t stop = 0;
    bool extend = false, branch = false, ok = false;
    unsigned xc = 0;
    const unsigned cand_m = solid_m & ~node_m;
    if (alive) {
      const uint32_t ncand = (uint32_t)__popc(cand_m & 0xFu), nnode_after = (uint32_t)__popc(node_m & 0xFu), nbefore = (uint32_t)__popc(solid_m >> 4);
      if (nbefore || ncand + nnode_after > 1) {
        // no linear extension (:3122-3165): my end is registered, the candidates become contigs of their own
        stop = SHK_STOP_BRANCH; branch = true;
        if (sub == 0) ok = shk_ug_put(G, rcw, id, false);
      } else if (ncand == 1) {
        xc = (unsigned)__ffs((int)(cand_m & 0xFu)) - 1;
        const shk_u128 nxt = ((win << 2) | xc) & wmask;
        if (nxt == first) {
          // a pure circle (:3176-3183). The reference registers first k-mer and RC(last k-mer) here, keys that depend on
          // where the seed cut the circle. Seeds of one batch that lie on the same circle all get this far at the same
          // time, each with its own cut; their keys would then stop each other's later steps as "known nodes" and leave
          // fragments and several rotations behind (seen on the GPU; the emulator runs workgroups one after another and
          // never did). A pure circle has no solid neighbour outside itself, so nobody else ever asks the map for its
          // k-mers: it registers NOTHING there and is owned through the circle set alone (minimum canonical hash over
          // its k-mers -> smallest id); the graph pass's own map (k_ug_map2) gives the kept one its two self links.
          stop = SHK_STOP_CIRCLE;
          if (sub == 0) { shk_ug_circle_put(G, hmin, id); ok = true; }
        } else if (len >= max_len) {
          stop = SHK_STOP_BUFFER;
          if (sub == 0) { atomicAdd(&G.stats[2], 1ULL); ok = shk_ug_put(G, rcw, id, false); }
        } else extend = true;
      } else {
        // one known unitig start ahead, or nothing (:3191-3216)
        stop = nnode_after == 1 ? SHK_STOP_BRANCH : SHK_STOP_DEAD_END;
        if (sub == 0) ok = shk_ug_put(G, rcw, id, false);
      }
    }
    const bool closed_ok = __shfl((int)ok, (int)gshift) != 0;   // (every lane of the wave shuffles here, once per step)
    uint32_t v = 0;
    bool closing = alive && !extend;
    if (alive && extend) {
      const uint64_t f = shk_rol64(fh, 1) ^ shk_rol64(shk_code_seed(s0), k & 63) ^ shk_code_seed(xc);
      const uint64_t r = shk_ror64(rh ^ shk_code_seed_rc(s0), 1) ^ shk_rol64(shk_code_seed_rc(xc), (k - 1) & 63);
      win = ((win << 2) | xc) & wmask;
      rcw = (rcw >> 2) | ((shk_u128)(3u - xc) << (2 * (k - 1)));
      fh = f; rh = r;
      const uint64_t hc = fh < rh ? fh : rh;
      if (hc < hmin) hmin = hc;
      len++;
      if (sub == 0) ext++;
    }
    if (closing) {
      if (branch && closed_ok && ((cand_m >> sub) & 1u)) {
        // my neighbour is a candidate: it becomes a contig of one k-mer unless somebody has queued it already (:3133-3160)
        const shk_u128 key = sub < 4 ? (((win << 2) | sub) & wmask) : ((rcw & ~(shk_u128)3) | (sub - 4));
        if (!shk_ug_find(G, key)) {                            // (cheap pre-check: most neighbours are known already)
          const uint32_t nid = atomicAdd(G.ncontigs, 1u);
          if (nid >= G.cap) atomicOr(G.flags, SHK_UG_E_CONTIGS);
          else if (shk_ug_put(G, key, nid, true)) {
            shk_ug_init(G, nid, key, k, mycnt, 0);
            G.next[atomicAdd(G.next_n, 1u)] = nid;
            atomicAdd(&G.stats[3], 1ULL);
          } else { G.state[nid] = SHK_UG_UNUSED; G.len[nid] = 0; }
        }
      }
      // a seed after its first call (:1886-1904) looks its seed k-mer up
      if (closed_ok && (kind & 1u) && !(kind & 2u) && stop != SHK_STOP_CIRCLE && sub == 0) v = shk_ug_find(G, first);
    }
    v = (uint32_t)__shfl((int)v, (int)gshift);
    if (!closing) continue;
    // ---- this call of get_unitig_forward is over
    bool turn = false;
    uint8_t nstate = SHK_UG_CLOSED;
    if (!closed_ok) nstate = SHK_UG_CLEARED;
    else if ((kind & 1u) && !(kind & 2u) && stop != SHK_STOP_CIRCLE) {
      // unless the seed k-mer already belongs to a smaller contig (or to this one), turn round
      if (v != 0 && v < id) nstate = SHK_UG_CLEARED;
      else if (v != id) turn = true;
    }
    if (sub == 0) {
      G.cur_lo[id] = (uint64_t)win; G.cur_hi[id] = (uint64_t)(win >> 64);
      G.rc_lo[id] = (uint64_t)rcw; G.rc_hi[id] = (uint64_t)(rcw >> 64);
      G.fh[id] = fh; G.rh[id] = rh; G.hmin[id] = hmin; G.len[id] = len;
      G.stop[id] = (uint8_t)((G.stop[id] << 4) | stop);
      if (nstate == SHK_UG_CLEARED) atomicAdd(&G.stats[1], 1ULL);
    }
    if (turn) {
      const shk_u128 nfirst = rcw, ncur = shk_ug_rc(first, k), nrc = first;
      first = nfirst; win = ncur; rcw = nrc;
      // the seed k-mer itself need not be solid for extensions (its count is tested against -x/-X, the walk's
      // neighbours against -s): k_ug_emit, which follows solid successors, is told the one base it cannot find that way --
      // the last base of RC(seed), appended at position l1 - 1 of the final orientation (bit 3 + bits 4-5 of kind)
      kind = (uint8_t)((kind & 0x07u) | 2u | 4u | 8u | ((3u - ((unsigned)(nrc >> (2 * (k - 1))) & 3u)) << 4));
      if (sub == 0) { G.first_lo[id] = (uint64_t)first; G.first_hi[id] = (uint64_t)(first >> 64); G.l1[id] = len; }
      continue;                                                // second call, same lanes, remaining step budget
    }
    if (nstate == SHK_UG_CLOSED && stop == SHK_STOP_CIRCLE) kind |= 0x40u;      // owned through the circle set (k_ug_check)
    if (sub == 0) { G.state[id] = nstate; G.kind[id] = kind; }
    alive = false;
  }
  if (ext) atomicAdd(&G.stats[0], ext);
}

// after all walks: a contig survives when both of its keys still map to it (check_unitig :935-954 tests the first k-mer; the
// walk itself gives up a contig whose end belongs to a smaller one, :3132, :3196, :3204 -- here both tests are final) and,
// for a pure circle, when it owns the circle. keep[id] = 1 / 0.
__global__ void k_ug_check(ShkUG G, uint32_t n, uint32_t *keep, uint32_t *lens) {
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n) return;
  uint32_t kp = 0;
  if (id >= 1 && G.state[id] == SHK_UG_CLOSED) {
    const shk_u128 first = ((shk_u128)G.first_hi[id] << 64) | G.first_lo[id];
    const shk_u128 rcw = ((shk_u128)G.rc_hi[id] << 64) | G.rc_lo[id];
    if (G.kind[id] & 0x40u) kp = shk_ug_circle_find(G, G.hmin[id]) == id;               // a pure circle: the circle set's owner
    else kp = shk_ug_find(G, first) == id && shk_ug_find(G, rcw) == id;
    if (!kp) { G.state[id] = SHK_UG_CLEARED; atomicAdd(&G.stats[1], 1ULL); }
  }
  keep[id] = kp;
  lens[id] = kp ? G.len[id] : 0;
}

// the kept contigs once more from their first k-mer: bases (text) and the filter count of every k-mer, into the arena
__global__ void k_ug_emit(ShkUG G, uint32_t n, const uint32_t *keep, const uint64_t *newid, const uint64_t *off, uint8_t *tab,
                          uint64_t q_lo, uint64_t nslots, uint32_t hb, uint32_t k, uint64_t amin, char *bases, uint32_t *counts,
                          uint64_t *out_off, uint32_t *out_len, uint32_t *out_l1) {
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n || !keep[id]) return;
  const uint64_t kmask = hb >= 64 ? ~0ULL : ((1ULL << hb) - 1);
  const shk_u128 wmask = shk_ug_mask(k);
  shk_u128 win = ((shk_u128)G.first_hi[id] << 64) | G.first_lo[id];
  const uint32_t len = G.len[id], l1 = G.l1[id];
  const uint8_t kind = G.kind[id];
  const uint64_t o = off[id];
  const uint64_t u = newid[id];                  // 0-based final number
  out_off[u] = o; out_len[u] = len; out_l1[u] = l1;
  for (uint32_t j = 0; j < k; j++) bases[o + j] = "ACGT"[(unsigned)(win >> (2 * (k - 1 - j))) & 3u];
  uint64_t fh, rh;
  shk_ug_hash(win, k, &fh, &rh);
  uint8_t trav;
  {
    const uint64_t c0 = shk_lookup_one(tab, (fh < rh ? fh : rh) & kmask, q_lo, nslots, 2, &trav);
    counts[o] = c0 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)c0;
  }
  for (uint32_t p = k; p < len; p++) {
    const unsigned s0 = (unsigned)(win >> (2 * (k - 1))) & 3u;
    const uint64_t fbase = shk_rol64(fh, 1) ^ shk_rol64(shk_code_seed(s0), k & 63);
    const uint64_t rbase = shk_ror64(rh ^ shk_code_seed_rc(s0), 1);
    unsigned xc = 0;
    uint64_t cx = 0, fx = 0, rx = 0;
    if ((kind & 8u) && p == l1 - 1) {
      // the step onto RC(seed k-mer): taken whatever its count (see k_ug_walk's turn)
      xc = (kind >> 4) & 3u;
      fx = fbase ^ shk_code_seed(xc);
      rx = rbase ^ shk_rol64(shk_code_seed_rc(xc), (k - 1) & 63);
      cx = shk_lookup_one(tab, (fx < rx ? fx : rx) & kmask, q_lo, nslots, 2, &trav);
    } else {
      for (unsigned x = 0; x < 4; x++) {
        const uint64_t f = fbase ^ shk_code_seed(x);
        const uint64_t r = rbase ^ shk_rol64(shk_code_seed_rc(x), (k - 1) & 63);
        const uint64_t cnt = shk_lookup_one(tab, (f < r ? f : r) & kmask, q_lo, nslots, 2, &trav);
        if (cnt >= amin) { xc = x; cx = cnt; fx = f; rx = r; }
      }
    }
    bases[o + p] = "ACGT"[xc];
    counts[o + p - k + 1] = cx > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cx;
    win = ((win << 2) | xc) & wmask;
    fh = fx; rh = rx;
  }
}

// Contig::median_abundance (an int) as the reference arrives at it: median() of the first call's abundances (the seed's
// count and one count per appended base), then -- for a seed's second call -- median() of (length - K + 1) copies of
// that value followed by the second call's counts (:3049, :3141, base/Utility.cpp:27-40). One wave per unitig; counts
// below 4096 go through an LDS histogram, the (rare) rest through a bounded number of compare-and-count passes.
__device__ __forceinline__ uint32_t shk_ug_kth(const uint32_t *c, uint32_t n, uint32_t extra_val, uint32_t extra_n, uint32_t kth,
                                               uint32_t *hist) {
  // k-th smallest (0-based) of c[0..n) plus extra_n copies of extra_val; all lanes of the wave call it together
  const unsigned lane = shk_lane();
  uint32_t lo = 0, hi = 0xFFFFFFFFu;
  // histogram pass over the low range
  for (uint32_t i = lane; i < 4096; i += SHK_WAVE) hist[i] = 0;
  __syncthreads();
  uint32_t big = 0;
  for (uint32_t i = lane; i < n; i += SHK_WAVE) {
    const uint32_t v = c[i];
    if (v < 4096) atomicAdd(&hist[v], 1u); else big++;
  }
  if (extra_n && lane == 0) { if (extra_val < 4096) atomicAdd(&hist[extra_val], extra_n); else big += extra_n; }
  __syncthreads();
  // prefix over the histogram: lane l sums its 64 bins
  uint32_t mine = 0;
  for (uint32_t j = 0; j < 64; j++) mine += hist[lane * 64 + j];
  const uint32_t incl = shk_wave_incl_add(mine);
  const uint32_t below = __shfl(incl, SHK_WAVE - 1);
  if (kth < below) {
    const unsigned long long m = __ballot(incl > kth);
    const unsigned owner = (unsigned)(__ffsll((long long)m) - 1);
    uint32_t res = 0;
    if (lane == owner) {
      uint32_t run = incl - mine;
      for (uint32_t j = 0; j < 64; j++) { run += hist[lane * 64 + j]; if (run > kth) { res = lane * 64 + j; break; } }
    }
    return __shfl(res, owner);
  }
  (void)big;
  // the k-th value is >= 4096: bisect on the value with counting passes (32 at most)
  lo = 4096;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    uint32_t le = 0;                                  // how many values are <= mid
    for (uint32_t i = lane; i < n; i += SHK_WAVE) le += c[i] <= mid;
    le = __shfl(shk_wave_incl_add(le), SHK_WAVE - 1);
    if (extra_n && extra_val <= mid) le += extra_n;
    if (le > kth) hi = mid; else lo = mid + 1;
  }
  return lo;
}
__device__ __forceinline__ int shk_ug_median(const uint32_t *c, uint32_t n, uint32_t extra_val, uint32_t extra_n, uint32_t *hist) {
  const uint32_t tot = n + extra_n;
  if (tot == 0) return 0;
  const uint32_t t = tot / 2;
  const uint32_t hiv = shk_ug_kth(c, n, extra_val, extra_n, t, hist);
  if (tot % 2) return (int)hiv;
  const uint32_t lov = shk_ug_kth(c, n, extra_val, extra_n, t - 1, hist);
  return (int)(((double)lov + (double)hiv) / 2.0);
}
__global__ void __launch_bounds__(SHK_WAVE) k_ug_median(uint32_t nunits, const uint64_t *off, const uint32_t *len, const uint32_t *l1,
                                                        const uint32_t *counts, uint32_t k, int32_t *median) {
  __shared__ uint32_t hist[4096];
  const uint32_t u = blockIdx.x;
  if (u >= nunits) return;
  const uint32_t *c = counts + off[u];
  const uint32_t nk = len[u] - k + 1, n1 = l1[u] - k + 1;
  const int m1 = shk_ug_median(c, n1, 0, 0, hist);
  int m = m1;
  if (nk > n1) m = shk_ug_median(c + n1, nk - n1, (uint32_t)m1, n1, hist);
  if (threadIdx.x == 0) median[u] = m;
}

// the graph pass's map (track_kmer_worker :956-1010): first k-mer -> +number, RC(last k-mer) -> -number (numbers from 1),
// one key only for a unitig whose two keys coincide. Built into a fresh table (G.m*), value = number << 1 | minus.
__global__ void k_ug_map2(ShkUG G, uint32_t n, const uint32_t *keep, const uint64_t *newid) {
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n || !keep[id]) return;
  const uint32_t num = (uint32_t)newid[id] + 1;
  const shk_u128 f = ((shk_u128)G.first_hi[id] << 64) | G.first_lo[id], e = ((shk_u128)G.rc_hi[id] << 64) | G.rc_lo[id];
  shk_ug_put(G, f, num << 1, false);
  if (f != e) shk_ug_put(G, e, (num << 1) | 1u, false);
}
// build_graph_worker (:1012-1084): successors of my last k-1 bases in A,C,G,T order, then of RC(my first k-1 bases) in
// T,G,C,A order; links[u*8 + i] = signed number (0 = none)
__global__ void k_ug_links(ShkUG G, uint32_t n, const uint32_t *keep, const uint64_t *newid, uint32_t k, int32_t *links) {
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n || !keep[id]) return;
  const shk_u128 wmask = shk_ug_mask(k);
  const shk_u128 first = ((shk_u128)G.first_hi[id] << 64) | G.first_lo[id], last = ((shk_u128)G.cur_hi[id] << 64) | G.cur_lo[id];
  int32_t *out = links + (size_t)newid[id] * 8;
  for (unsigned x = 0; x < 4; x++) {
    const uint32_t v = shk_ug_find(G, ((last << 2) | x) & wmask);
    out[x] = v ? ((v & 1u) ? -(int32_t)(v >> 1) : (int32_t)(v >> 1)) : 0;
  }
  // RC(first k-1 bases) + x = RC(first k-mer) without ITS first base, + x
  const shk_u128 rcf = shk_ug_rc(first, k);
  for (unsigned i = 0; i < 4; i++) {
    const unsigned x = 3u - i;
    const uint32_t v = shk_ug_find(G, ((rcf << 2) | x) & wmask);
    out[4 + i] = v ? ((v & 1u) ? -(int32_t)(v >> 1) : (int32_t)(v >> 1)) : 0;
  }
}

__global__ void k_ug_rehash(ShkUG G, const uint64_t *ok_lo, const uint64_t *ok_hi, const uint32_t *ov, uint32_t ocap) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ocap) return;
  const uint32_t v = ov[i];
  if (v == 0 || v == SHK_UG_BUSY) return;
  shk_ug_put(G, ((shk_u128)ok_hi[i] << 64) | ok_lo[i], v, false);
}
__global__ void k_ug_rehash_circles(ShkUG G, const uint64_t *ok, const uint32_t *ov, uint32_t ocap) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ocap) return;
  const uint32_t v = ov[i];
  if (v == 0 || v == SHK_UG_BUSY) return;
  shk_ug_circle_put(G, ok[i], v);
}
