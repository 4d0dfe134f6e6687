// Filter-to-filter kernels: the GPU form of the reference's iterator and merge utilities
//   qf_iterator / qfi_get / qfi_next          cqf/gqf.c:2474-2601  -> k_region_dump
//   qf_merge / qf_multi_merge                 cqf/gqf.c:2614-2704  -> k_region_merge2
// and of the stitch of quotient-range shards into one table, which is the same
// rebuild with a different second source.
//
// Both read a region's entries straight out of the packed table: one wave per region of 256
// quotients stages the region's bytes in LDS, finds its runs with rank/select over the masked
// runends words (one lane per run), and every lane walks the entries of its four quotients in
// (quotient, remainder) order -- that is the iterator's order, and two such walks merge like
// qf_merge's two iterators. The merged table is laid out by the usual two launches (lengths ->
// k_region_scan_* -> write), so it is the canonical table of the summed multiset: exactly what
// qf_merge's qf_insert calls produce.
#include "shk_device.h"

#define SHK_MAX_SRC 64

// the second source of a two-source rebuild: regions [s * regions_per_src, (s+1) * regions_per_src) of the
// destination take their second operand from table s (one table of the same geometry for a merge; the
// shards' tables, each with its own overflow tail, for a stitch)
struct ShkSrc2 {
  const uint8_t *tab[SHK_MAX_SRC];
  const uint64_t *fin[SHK_MAX_SRC];   // free pointer at every region start of that table (k_build_fin)
  uint64_t nblocks;                   // blocks of one source table
  uint32_t regions_per_src;
  uint32_t nsrc;
};

// A region of one table, staged and indexed (lives in LDS).
template <int IMGB>
struct ShkRegionView {
  __attribute__((aligned(16))) uint8_t img[IMGB * SHK_BLOCK_BYTES + 16];
  uint64_t occ[SHK_REGION_BLOCKS];
  uint64_t runw[IMGB];
  uint32_t orank[SHK_REGION_BLOCKS + 1];
  uint32_t rrank[IMGB + 1];
  uint16_t rend[SHK_REGION];          // slot (image relative) of the j-th runend of the region
  uint32_t olo, nruns;
};

// Wave-collective (64 threads, every lane calls it): stage region `r` of `tab` and index its runs.
// Returns false (after raising `err`) when the region's runs reach further than the image.
template <int IMGB>
__device__ __forceinline__ bool shk_view_load(ShkRegionView<IMGB> &V, const uint8_t *tab, const uint64_t *fin, uint32_t r,
                                              uint32_t nq, uint64_t nblocks, unsigned tid, uint32_t *err) {
  constexpr unsigned IMG_SLOTS = IMGB * 64;
  const uint64_t q0 = (uint64_t)r * SHK_REGION, b0 = q0 / 64;
  const uint32_t nown = (nq + 63) / 64;
  bool ok = true;
  uint32_t olo = 0, ohi = 0, nblk = nown;
  bool any = false;
  if (tab) {
    const uint64_t fa0 = fin[r], fa1 = fin[r + 1];
    const uint64_t olo_abs = fa0 > q0 ? fa0 : q0;
    any = fa1 > olo_abs;
    olo = (uint32_t)(olo_abs - q0);
    ohi = any ? (uint32_t)((fa1 - q0) > 0xFFFFFFF ? 0xFFFFFFF : (fa1 - q0)) : olo;
    if (any && (ohi + 63) / 64 > nblk) nblk = (ohi + 63) / 64;
    if (nblk > (uint32_t)IMGB || ohi > IMG_SLOTS) {
      if (tid == 0) atomicOr(err, SHK_E_OLD_EXTENT);
      ok = false; any = false; nblk = nown;
    }
    if (b0 + nblk > nblocks) nblk = (uint32_t)(nblocks - b0);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(tab + b0 * SHK_BLOCK_BYTES);
    uint32_t *dst = reinterpret_cast<uint32_t *>(V.img);
    for (uint32_t i = tid; i < (nblk * SHK_BLOCK_BYTES + 3) / 4; i += SHK_WAVE) dst[i] = src[i];
  } else {
    nblk = 0;
  }
  __syncthreads();
  uint64_t ow = 0;
  if (tab && tid < nown) ow = shk_ld64(V.img + tid * SHK_BLOCK_BYTES + SHK_OFF_OCC);
  if (!ok) ow = 0;
  if (tid < SHK_REGION_BLOCKS) V.occ[tid] = ow;
  const uint32_t pc = (uint32_t)__popcll(ow);
  const uint32_t inc = shk_wave_incl_add(pc);
  if (tid < SHK_REGION_BLOCKS) V.orank[tid] = inc - pc;
  if (tid == SHK_REGION_BLOCKS - 1) V.orank[SHK_REGION_BLOCKS] = inc;
  uint64_t rw = 0;
  if (any && tid < nblk) {
    rw = shk_ld64(V.img + tid * SHK_BLOCK_BYTES + SHK_OFF_RUN);
    const uint32_t s0 = tid * 64;
    if (s0 + 64 <= olo || s0 >= ohi) rw = 0;
    else {
      if (olo > s0) rw &= ~((1ULL << (olo - s0)) - 1);
      if (ohi < s0 + 64) rw &= ((1ULL << (ohi - s0)) - 1);
    }
  }
  const uint32_t rc = (uint32_t)__popcll(rw);
  const uint32_t rinc = shk_wave_incl_add(rc);
  if (tid < (unsigned)IMGB) { V.rrank[tid] = rinc - rc; V.runw[tid] = rw; }
  if (tid == SHK_WAVE - 1) { V.rrank[IMGB] = rinc; V.olo = olo; }
  __syncthreads();
  const uint32_t nruns = V.rrank[IMGB];
  if (nruns != V.orank[SHK_REGION_BLOCKS] || nruns > SHK_REGION) {
    if (tid == 0) atomicOr(err, SHK_E_CORRUPT);
    ok = false;
  }
  if (ok)
    for (uint32_t j = tid; j < nruns; j += SHK_WAVE) {
      uint32_t lo = 0, hi = IMGB;   // last word w with rrank[w] <= j
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (V.rrank[mid] <= j) lo = mid; else hi = mid;
      }
      V.rend[j] = (uint16_t)(lo * 64 + shk_select64(V.runw[lo], j - V.rrank[lo]));
    }
  if (tid == 0) V.nruns = ok ? nruns : 0;
  __syncthreads();
  return ok;
}

// One lane's walk over the entries of its `per` consecutive quotients starting at qa, in (quotient, remainder) order
struct ShkCur {
  uint32_t occ, jr, q, pos, end, rem, n;
  uint64_t cnt;
  bool has, first;              // first: the current entry opens its run
};
template <int IMGB>
__device__ __forceinline__ void shk_cur_open_run(ShkCur &c, const ShkRegionView<IMGB> &V, uint32_t qa, uint32_t prev_end_plus1) {
  c.q = qa + (uint32_t)__ffs((int)c.occ) - 1;
  c.occ &= c.occ - 1;
  c.end = V.rend[c.jr];
  c.pos = prev_end_plus1 > c.q ? prev_end_plus1 : c.q;
  c.n = shk_img_dec(V.img, c.pos, c.end, &c.rem, &c.cnt);
  c.has = true; c.first = true;
}
template <int IMGB>
__device__ __forceinline__ void shk_cur_init(ShkCur &c, const ShkRegionView<IMGB> &V, uint32_t qa, uint32_t per, bool active) {
  c.has = false; c.first = false; c.occ = 0; c.jr = 0; c.q = 0; c.pos = 0; c.end = 0; c.rem = 0; c.n = 0; c.cnt = 0;
  if (!active || V.nruns == 0) return;
  const uint64_t ow = V.occ[qa >> 6];
  c.occ = (uint32_t)(ow >> (qa & 63)) & ((1u << per) - 1);
  c.jr = V.orank[qa >> 6] + (uint32_t)__popcll(ow & ((1ULL << (qa & 63)) - 1));
  if (c.occ) shk_cur_open_run(c, V, qa, c.jr ? (uint32_t)V.rend[c.jr - 1] + 1 : V.olo);
}
template <int IMGB>
__device__ __forceinline__ void shk_cur_next(ShkCur &c, const ShkRegionView<IMGB> &V, uint32_t qa) {
  c.pos += c.n;
  if (c.pos <= c.end) { c.n = shk_img_dec(V.img, c.pos, c.end, &c.rem, &c.cnt); c.first = false; }
  else if (c.occ) { c.jr++; shk_cur_open_run(c, V, qa, c.end + 1); }
  else c.has = false;
}

// two cursors merged; f(quotient, remainder, total count, present in 1, count in 2, slot of the entry in 1, opens its run in 1)
template <int IMGB, typename F>
__device__ __forceinline__ void shk_walk2(const ShkRegionView<IMGB> &V1, const ShkRegionView<IMGB> &V2, uint32_t qa, uint32_t per,
                                          bool active, F &&f) {
  ShkCur a, b;
  shk_cur_init(a, V1, qa, per, active);
  shk_cur_init(b, V2, qa, per, active);
  constexpr uint32_t NONE = 0xFFFFFFu;
  while (a.has || b.has) {
    const uint32_t ca = a.has ? ((a.q << 8) | a.rem) : NONE, cb = b.has ? ((b.q << 8) | b.rem) : NONE;
    const uint32_t comp = ca < cb ? ca : cb;
    uint64_t total = 0, c2 = 0;
    const bool in1 = ca == comp, in2 = cb == comp;
    const uint32_t pos1 = a.pos;
    const bool first1 = a.first;
    if (in1) { total += a.cnt; shk_cur_next(a, V1, qa); }
    if (in2) { c2 = b.cnt; total += c2; shk_cur_next(b, V2, qa); }
    f(comp >> 8, comp & 0xff, total, in1, c2, pos1, first1);
  }
}

// WRITE = false: run lengths -> (T, c) + statistics into the summary (then k_region_scan_*).
// WRITE = true : lay the merged runs out from the free pointers in finB and store table B.
template <bool WRITE, int IMGB>
__global__ void __launch_bounds__(SHK_WAVE) k_region_merge2(ShkMergeArgs A, ShkSrc2 S) {
  constexpr unsigned IMG_SLOTS = IMGB * 64, IMG_BYTES = IMGB * SHK_BLOCK_BYTES;
  __shared__ ShkRegionView<IMGB> V1, V2;
  __shared__ __attribute__((aligned(16))) uint8_t nimg[IMG_BYTES + 16];
  __shared__ uint32_t qlen[SHK_REGION];
  const unsigned tid = threadIdx.x;
  const uint32_t r = blockIdx.x + A.r0;
  const uint32_t nregions = (uint32_t)((A.nslots + SHK_REGION - 1) / SHK_REGION);
  const uint64_t q0 = (uint64_t)r * SHK_REGION;
  const uint32_t nq = (uint32_t)((A.nslots - q0) < SHK_REGION ? (A.nslots - q0) : SHK_REGION);
  const uint32_t nown = (nq + 63) / 64;
  const uint64_t b0 = q0 / 64;
  const uint32_t s = r / S.regions_per_src, r2 = r - s * S.regions_per_src;
  const bool have2 = s < S.nsrc && S.tab[s] != nullptr;
  bool ok = shk_view_load<IMGB>(V1, A.tabA, A.finA, r, nq, A.nblocks, tid, A.err);
  ok = shk_view_load<IMGB>(V2, have2 ? S.tab[s] : nullptr, have2 ? S.fin[s] : nullptr, r2, nq, S.nblocks, tid, A.err) && ok;
  for (uint32_t i = tid; i < SHK_REGION; i += SHK_WAVE) qlen[i] = 0;
  if (WRITE) {
    uint32_t *z = reinterpret_cast<uint32_t *>(nimg);
    for (uint32_t i = tid; i < (IMG_BYTES + 16) / 4; i += SHK_WAVE) z[i] = 0;
  }
  __syncthreads();
  constexpr uint32_t per = SHK_REGION / SHK_WAVE;
  const uint32_t qa = tid * per;
  const bool active = ok && qa < nq;
  // ---- lengths
  ShkMP mine; mine.a = 0; mine.b = SHK_NEG_INF;
  uint32_t my_new = 0;
  unsigned long long my_added = 0;
  {
    uint32_t curq = 0xFFFFFFFFu, len = 0;
    auto close_run = [&]() {
      if (len) {
        qlen[curq] = len;
        ShkMP m; m.a = len; m.b = (long long)curq + len;
        mine = shk_mp_compose(mine, m);
      }
    };
    shk_walk2<IMGB>(V1, V2, qa, per, active, [&](uint32_t q, uint32_t rem, uint64_t total, bool in1, uint64_t c2, uint32_t, bool) {
      if (q != curq) { close_run(); curq = q; len = 0; }
      len += shk_enc_len(rem, total);
      if (!in1) my_new++;
      my_added += c2;
    });
    close_run();
  }
  ShkMP incl = mine;
  for (int d = 1; d < SHK_WAVE; d <<= 1) {
    ShkMP y;
    y.a = __shfl_up(incl.a, d);
    y.b = __shfl_up(incl.b, d);
    if (tid >= (unsigned)d) incl = shk_mp_compose(y, incl);
  }
  ShkMP tot, pre;
  tot.a = __shfl(incl.a, SHK_WAVE - 1);
  tot.b = __shfl(incl.b, SHK_WAVE - 1);
  pre.a = __shfl_up(incl.a, 1);
  pre.b = __shfl_up(incl.b, 1);
  if (tid == 0) { pre.a = 0; pre.b = SHK_NEG_INF; }
  if (!WRITE) {
    const uint32_t t_new = shk_wave_incl_add(my_new);
    const uint64_t t_added = shk_wave_incl_add64(my_added);
    if (tid == SHK_WAVE - 1) {
      uint32_t *sm = A.summary + (size_t)SHK_SUM_STRIDE * r;
      sm[0] = ok ? (uint32_t)tot.a : 0;
      sm[1] = (ok && tot.b > 0) ? (uint32_t)tot.b : 0;
      sm[2] = t_new; sm[3] = 0; sm[4] = 0; sm[5] = 0; sm[6] = 0;
      if (tot.a > 0xFFFF) atomicOr(A.err, SHK_E_RUN_TOO_LONG);
      if (t_added) atomicAdd(&A.counters[1], (unsigned long long)t_added);   // may exceed 32 bits per region: not via the summary
    }
    return;
  }
  if (!ok) return;
  // ---- placement
  const long long fin_rel = (long long)A.finB[r] - (long long)q0;
  const long long fout_rel = (long long)A.finB[r + 1] - (long long)q0;
  const uint32_t out_lo = fin_rel > 0 ? (uint32_t)fin_rel : 0;
  const bool new_any = tot.a > 0;
  const uint32_t out_hi = new_any ? (uint32_t)fout_rel : out_lo;
  if (new_any && (fout_rel > (long long)IMG_SLOTS || fout_rel < 0)) {
    if (tid == 0) atomicOr(A.err, SHK_E_NEW_EXTENT);
    return;
  }
  uint32_t *nimg32 = reinterpret_cast<uint32_t *>(nimg);
  uint32_t rstart[per];
  {
    long long f = shk_mp_apply(pre, fin_rel);
    for (uint32_t j = 0; j < per; j++) {
      const uint32_t q = qa + j;
      rstart[j] = 0;
      if (q < nq && (q & 63) == 0) {
        long long o = f - (long long)q;   // block_offset_strict, gqf.c:599-601, saturating like the 8-bit field
        nimg[(q >> 6) * SHK_BLOCK_BYTES] = (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o));
      }
      const uint32_t len = q < SHK_REGION ? qlen[q] : 0;
      if (len) {
        const long long stt = f > (long long)q ? f : (long long)q;
        rstart[j] = (uint32_t)stt;
        f = stt + len;
      }
    }
  }
  {
    uint32_t curq = 0xFFFFFFFFu, wp = 0;
    auto close_run = [&]() {
      if (curq != 0xFFFFFFFFu) {
        const uint32_t last = wp - 1;   // runend bit on the run's last slot, occupied bit on its quotient
        const uint32_t bo = (last >> 6) * SHK_BLOCK_BYTES + SHK_OFF_RUN + ((last & 63) >> 3);
        atomicOr(&nimg32[bo >> 2], 1u << (((bo & 3) << 3) + (last & 7)));
        const uint32_t oo = (curq >> 6) * SHK_BLOCK_BYTES + SHK_OFF_OCC + ((curq & 63) >> 3);
        atomicOr(&nimg32[oo >> 2], 1u << (((oo & 3) << 3) + (curq & 7)));
      }
    };
    shk_walk2<IMGB>(V1, V2, qa, per, active, [&](uint32_t q, uint32_t rem, uint64_t total, bool, uint64_t, uint32_t, bool) {
      if (q != curq) { close_run(); curq = q; wp = rstart[q - qa]; }
      uint8_t enc[12];
      const unsigned n = shk_enc_write(enc, rem, total);
      for (unsigned i = 0; i < n; i++) nimg[shk_img_slot_off(wp + i)] = enc[i];
      wp += n;
    });
    close_run();
  }
  __syncthreads();
  shk_store_image<IMGB>(A, r, nregions, nimg, tid, nown, b0, q0, out_lo, out_hi, new_any, fout_rel);
}

// (key, count) of every entry in iterator order. PASS 0: entries per region -> nper[r]. PASS 1: write them at
// offs[r] (the exclusive scan of nper). key = (global quotient << 8) | remainder, as qfi_get composes it (gqf.c:2516).
// `stop` (PASS 1, may be null) receives the index at which the reference's qfi_next ends the iteration early: it
// returns 1 when it steps INSIDE a run onto a slot behind nslots (gqf.c:2537-2539), so entries that sit in the
// overflow tail without opening their run -- and everything behind them -- are never visited by the reference.
template <int PASS, int IMGB>
__global__ void __launch_bounds__(SHK_WAVE) k_region_dump(ShkMergeArgs A, uint32_t *nper, const uint64_t *offs, uint64_t *keys,
                                                          uint64_t *counts, uint64_t cap, unsigned long long *stop) {
  __shared__ ShkRegionView<IMGB> V1, V2;
  const unsigned tid = threadIdx.x;
  const uint32_t r = blockIdx.x + A.r0;
  const uint64_t q0 = (uint64_t)r * SHK_REGION;
  const uint32_t nq = (uint32_t)((A.nslots - q0) < SHK_REGION ? (A.nslots - q0) : SHK_REGION);
  const bool ok = shk_view_load<IMGB>(V1, A.tabA, A.finA, r, nq, A.nblocks, tid, A.err);
  shk_view_load<IMGB>(V2, nullptr, nullptr, 0, nq, 0, tid, A.err);   // an empty second operand
  constexpr uint32_t per = SHK_REGION / SHK_WAVE;
  const uint32_t qa = tid * per;
  const bool active = ok && qa < nq;
  uint32_t mine = 0;
  shk_walk2<IMGB>(V1, V2, qa, per, active, [&](uint32_t, uint32_t, uint64_t, bool, uint64_t, uint32_t, bool) { mine++; });
  const uint32_t incl = shk_wave_incl_add(mine);
  if (PASS == 0) {
    if (tid == SHK_WAVE - 1) nper[r] = incl;
    return;
  }
  uint64_t o = offs[r] + (incl - mine);
  shk_walk2<IMGB>(V1, V2, qa, per, active, [&](uint32_t q, uint32_t rem, uint64_t total, bool, uint64_t, uint32_t pos, bool first) {
    if (keys && o < cap) { keys[o] = ((A.q_lo + q0 + q) << 8) | rem; counts[o] = total; }
    if (stop && !first && q0 + pos > A.nslots) atomicMin(stop, (unsigned long long)o);
    o++;
  });
}

// (key, count) pairs -> key words for the counted form of the rebuild: a pair becomes ceil(count / 4096) words whose
// 12-bit chunk field carries (multiplicity - 1). One thread per pair; nwords[i] first (PASS 0), then the words.
template <int PASS>
__global__ void k_expand_counted(const uint64_t *keys, const uint64_t *counts, uint64_t n, uint64_t skip, uint64_t take, uint32_t hb,
                                 uint32_t *nwords, const uint64_t *offs, uint64_t *words, uint32_t *err, uint64_t key_lo, uint64_t key_hi,
                                 unsigned long long *max_count) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // this call inserts occurrences [skip, skip + take) of every pair (counts beyond 2^30 take several calls: the
  // rebuild adds a batch's occurrences of one key in 32 bits)
  const uint64_t c = counts[i];
  uint64_t mine = c > skip ? c - skip : 0;
  if (mine > take) mine = take;
  const uint32_t nw = (uint32_t)((mine + SHK_MAX_CHUNKS - 1) / SHK_MAX_CHUNKS);
  if (PASS == 0) {
    nwords[i] = nw;
    if (mine && (keys[i] < key_lo || keys[i] >= key_hi)) atomicOr(err, SHK_E_CORRUPT);
    if (max_count && c > skip + take) atomicMax(max_count, (unsigned long long)c);
    return;
  }
  uint64_t o = offs[i];
  for (uint32_t j = 0; j < nw; j++) {
    const uint64_t part = mine - (uint64_t)j * SHK_MAX_CHUNKS < SHK_MAX_CHUNKS ? mine - (uint64_t)j * SHK_MAX_CHUNKS : SHK_MAX_CHUNKS;
    words[o + j] = keys[i] | ((part - 1) << hb);
  }
}
