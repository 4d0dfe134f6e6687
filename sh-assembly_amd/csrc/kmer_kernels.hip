// FASTQ text -> canonical ntHash key words, on device.
//
// Replaces the reference's per-thread loop reads_to_kmers (cqf/CQF_mt.h:610-731) and
// the ntHash calls it makes (base/nthash.hpp:295-309) for a whole batch of chunks:
//   k_count_lines   newline count per chunk (4-line records, CQF_mt.h:616-726)
//   k_scan_chunks   reads per chunk -> first read index of every chunk
//   k_emit_reads    [start,end) of every sequence line
//   k_count_keys    k-mers per read, honouring the reference's 'N' restart rule
//   k_hash_reads    one wave per read: keys in exactly the reference's emission order
//
// ntHash without the serial roll: with j counted from the segment start,
//   fh(p) = rol( G(p+k) ^ G(p), k-1+p ),  G(n) = XOR_{j<n} ror(seed[c_j], j)
//   rh(p) = ror( H(p+k) ^ H(p), p ),      H(n) = XOR_{j<n} rol(seed[comp c_j], j)
// (rotations mod 64) which is the closed form of nthash.hpp:295-309, so a wave gets
// 64 k-mers from one 64-lane XOR scan instead of 64 dependent rolls.
#include "shk_device.h"

#define SHK_SEED_A 0x3c8bfbb395c60474ULL  // base/nthash.hpp:24-27
#define SHK_SEED_C 0x3193c18562a02b4cULL
#define SHK_SEED_G 0x20323ed082572324ULL
#define SHK_SEED_T 0x295549f54be24456ULL
#define SHK_HASH_WAVES 4                   // waves per workgroup of k_hash_reads
#define SHK_MAX_K 191                     // ring of 256 prefix values per wave
#define SHK_MAX_READ 65535                // a record fits the chunker's overhead (CQF_mt.h:764)

// seedTab / msTab column 0 (nthash.hpp:85-153): upper and lower case map, all else is 0
__device__ __forceinline__ uint64_t shk_seed_fwd(unsigned c) {
  const unsigned u = c & 0xDF;
  return (u == 'A' || c == 4) ? SHK_SEED_A : (u == 'C' || c == 7) ? SHK_SEED_C : (u == 'G' || c == 3) ? SHK_SEED_G : (u == 'T' || c == 1) ? SHK_SEED_T : 0ULL;
}
// seed on the complement strand: the reference indexes seedTab with (c & cpOff) for every byte, nthash.hpp:15,299
__device__ __forceinline__ uint64_t shk_seed_rc(unsigned c) {
  c &= 7;
  return c == 1 ? SHK_SEED_T : c == 3 ? SHK_SEED_G : c == 7 ? SHK_SEED_C : c == 4 ? SHK_SEED_A : 0ULL;
}

// exact per-byte "== '\n'" flags (0x80 in every matching byte)
__device__ __forceinline__ uint32_t shk_nl_flags(uint32_t w) {
  uint32_t x = w ^ 0x0A0A0A0Au;
  return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}

// Loads the 16-byte unit `u` of a chunk (units are 16-B aligned in the text buffer) and
// returns per-word newline flags with bytes outside [off, off+len) masked away.
__device__ __forceinline__ uint4 shk_unit_flags(const uint8_t *text, uint64_t a0, uint64_t u, uint64_t off,
                                                uint64_t end) {
  const uint4 v = *reinterpret_cast<const uint4 *>(text + a0 + 16 * u);
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint64_t b0 = a0 + 16 * u;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint32_t f = shk_nl_flags(w[i]);
    uint64_t wb = b0 + 4 * i;
    if (wb < off || wb + 4 > end) {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (wb + j < off || wb + j >= end) f &= ~(0x80u << (8 * j));
    }
    w[i] = f;
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// ---------------------------------------------------------------- lines per chunk
// A chunk (8 MiB) is cut into SHK_PARSE_SEGS segments of whole 16-byte units; one workgroup per
// segment, so a batch of a few hundred chunks fills the device.
#define SHK_PARSE_SEGS 16
__device__ __forceinline__ void shk_segment_units(uint64_t nunits, unsigned seg, uint64_t *u0, uint64_t *u1) {
  const uint64_t per = (nunits + SHK_PARSE_SEGS - 1) / SHK_PARSE_SEGS;
  *u0 = per * seg < nunits ? per * seg : nunits;
  *u1 = *u0 + per < nunits ? *u0 + per : nunits;
}

__global__ void k_count_lines(const uint8_t *text, const uint64_t *chunk_off, const uint64_t *chunk_len,
                              uint64_t *nlines_seg) {
  __shared__ uint64_t scratch[SHK_MAX_WAVES + 1];
  const unsigned c = blockIdx.x / SHK_PARSE_SEGS, seg = blockIdx.x % SHK_PARSE_SEGS;
  const uint64_t off = chunk_off[c], end = off + chunk_len[c];
  const uint64_t a0 = off & ~15ULL;
  const uint64_t nunits = (end - a0 + 15) / 16;
  uint64_t u0, u1;
  shk_segment_units(nunits, seg, &u0, &u1);
  uint64_t cnt = 0;
  for (uint64_t u = u0 + threadIdx.x; u < u1; u += blockDim.x) {
    uint4 f = shk_unit_flags(text, a0, u, off, end);
    cnt += __popc(f.x) + __popc(f.y) + __popc(f.z) + __popc(f.w);
  }
  uint64_t tot = shk_block_sum64(cnt, scratch);
  if (threadIdx.x == 0) nlines_seg[blockIdx.x] = tot;
}

// reads per chunk = #newlines with index 1 mod 4 = (nl+2)/4; first read index per chunk;
// nlines_seg[] is turned into the first line index of every segment inside its chunk
__global__ void k_scan_chunks(uint64_t *nlines_seg, unsigned nchunks, uint64_t *reads_base, uint64_t *nreads_total) {
  __shared__ uint64_t scratch[SHK_MAX_WAVES + 1];
  __shared__ uint64_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (unsigned base = 0; base < nchunks; base += blockDim.x) {
    unsigned c = base + threadIdx.x;
    uint64_t nl = 0;
    if (c < nchunks)
      for (unsigned sg = 0; sg < SHK_PARSE_SEGS; sg++) {
        const uint64_t v = nlines_seg[(uint64_t)c * SHK_PARSE_SEGS + sg];
        nlines_seg[(uint64_t)c * SHK_PARSE_SEGS + sg] = nl;
        nl += v;
      }
    uint64_t v = c < nchunks ? (nl + 2) / 4 : 0;
    uint64_t tot;
    uint64_t ex = shk_block_exscan64(v, &tot, scratch);
    uint64_t carry = carry_s;
    if (c < nchunks) reads_base[c] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    reads_base[nchunks] = carry_s;
    *nreads_total = carry_s;
  }
}

// ---------------------------------------------------------------- sequence-line extents
// Line l (0-based, by newline count from the chunk start) is a header when l%4==0 and
// the read when l%4==1 -- the same strict 4-line walk as CQF_mt.h:616-726.
__global__ void k_emit_reads(const uint8_t *text, const uint64_t *chunk_off, const uint64_t *chunk_len,
                             const uint64_t *reads_base, const uint64_t *line_base_seg, uint64_t *rd_start, uint64_t *rd_end,
                             uint16_t *rd_chunk) {
  __shared__ uint32_t scratch[SHK_MAX_WAVES + 1];
  __shared__ uint64_t line_carry;
  const unsigned c = blockIdx.x / SHK_PARSE_SEGS, seg = blockIdx.x % SHK_PARSE_SEGS;
  const uint64_t off = chunk_off[c], end = off + chunk_len[c];
  const uint64_t a0 = off & ~15ULL;
  const uint64_t nunits = (end - a0 + 15) / 16;
  uint64_t u0, u1;
  shk_segment_units(nunits, seg, &u0, &u1);
  const uint64_t rbase = reads_base[c], nreads = reads_base[c + 1] - rbase;
  if (threadIdx.x == 0) line_carry = line_base_seg[blockIdx.x];
  __syncthreads();
  for (uint64_t ub = u0; ub < u1; ub += blockDim.x) {
    uint64_t u = ub + threadIdx.x;
    uint4 f = make_uint4(0, 0, 0, 0);
    if (u < u1) f = shk_unit_flags(text, a0, u, off, end);
    uint32_t n = __popc(f.x) + __popc(f.y) + __popc(f.z) + __popc(f.w);
    uint32_t tot;
    uint32_t ex = shk_block_exscan(n, &tot, scratch);
    uint64_t line = line_carry + ex;
    if (n) {
      uint32_t w[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        uint32_t m = w[i];
        while (m) {
          int bit = __ffs(m) - 1;  // 7, 15, 23 or 31
          m &= m - 1;
          uint64_t pos = a0 + 16 * u + 4 * i + (bit >> 3);
          uint64_t rd = line >> 2;
          if (rd < nreads) {
            if ((line & 3) == 0) { rd_start[rbase + rd] = pos + 1; rd_chunk[rbase + rd] = (uint16_t)c; }
            else if ((line & 3) == 1) rd_end[rbase + rd] = pos;
          }
          line++;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) line_carry += tot;
    __syncthreads();
  }
}

// ---------------------------------------------------------------- the 'N' restart rule
// reads_to_kmers hashes the first window of a (sub)read without looking at it, then
// restarts behind the first 'N' it meets at index >= k (CQF_mt.h:627-676). For the
// (sub)read starting at s this returns e = index of that 'N', or len when there is none.
// All lanes of the wave call it together; the result is wave-uniform.
__device__ __forceinline__ uint32_t shk_segment_end(const uint8_t *rd, uint32_t len, uint32_t s, uint32_t k) {
  const unsigned lane = shk_lane();
  for (uint32_t base = (s + k) & ~63u; base < len; base += 64) {
    uint32_t pos = base + lane;
    bool isN = pos >= s + k && pos < len && rd[pos] == 'N';
    unsigned long long m = __ballot(isN);
    if (m) return base + (uint32_t)(__ffsll((long long)m) - 1);
  }
  return len;
}

// The first 256 bases of a read, fetched with four independent loads (lane l holds bases l, 64+l, 128+l,
// 192+l; 0 beyond the end) and their 'N' masks: reads up to 256 bases need no further memory access for
// the restart rule. These kernels are bound by the latency of dependent loads, not by bandwidth.
__device__ __forceinline__ void shk_read_bytes(const uint8_t *text, uint64_t st, uint64_t en, unsigned lane, uint32_t c[4]) {
  const uint64_t len = en > st ? en - st : 0;
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const uint32_t pos = (uint32_t)t * 64 + lane;
    c[t] = pos < len ? text[st + pos] : 0u;
  }
}
__device__ __forceinline__ void shk_read_masks(const uint32_t c[4], unsigned long long nm[4]) {
#pragma unroll
  for (int t = 0; t < 4; t++) nm[t] = __ballot(c[t] == 'N');
}
__device__ __forceinline__ uint32_t shk_segment_end_pre(const unsigned long long nm[4], const uint8_t *rd, uint32_t len,
                                                        uint32_t s, uint32_t k) {
  if (len > 256) return shk_segment_end(rd, len, s, k);
  const uint32_t from = s + k;
#pragma unroll
  for (int t = 0; t < 4; t++) {
    if ((uint32_t)t < (from >> 6)) continue;
    unsigned long long m = nm[t];
    if ((uint32_t)t == (from >> 6)) m &= ~0ULL << (from & 63);
    if (m) return (uint32_t)t * 64 + (uint32_t)(__ffsll((long long)m) - 1);
  }
  return len;
}

// k-mers per read, one THREAD per read. The wave-per-read form of this kernel (four byte loads, four ballots and the
// restart loop per read: about 80 wave instructions for 150 bases) was bound by instruction issue, 1.6 ms for 8 M reads;
// here a thread looks at its read 16 aligned bytes at a time (SWAR compare with 'N'), and only a read that really has an
// 'N' behind its first window walks the restart rule byte by byte (CQF_mt.h:627-676: the first window of a (sub)read is
// hashed without being looked at; the read restarts behind the first 'N' at index >= k).
__device__ __forceinline__ uint32_t shk_bytes_below(uint32_t n) { return n >= 4 ? 0xFFFFFFFFu : ((1u << (8 * n)) - 1u); }
__global__ void k_count_keys(const uint8_t *text, const uint64_t *rd_start, const uint64_t *rd_end,
                             const uint64_t *nreads_p, uint32_t k, uint32_t *nkeys, uint32_t *err) {
  const uint64_t nreads = *nreads_p;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += stride) {
    const uint64_t st = rd_start[r], en = rd_end[r];
    uint32_t cnt = 0;
    if (en - st > SHK_MAX_READ) atomicOr(err, SHK_E_BAD_FASTQ);
    else if (en - st >= k) {
      const uint32_t len = (uint32_t)(en - st);
      const uint8_t *rd = text + st;
      // is there an 'N' at an index >= k at all?
      // (offsets into the text buffer, whose base is 16-byte aligned as for k_count_lines: units never start in front of it)
      const uint64_t a = st + k, e = en;
      uint32_t any = 0;
      for (uint64_t u = a & ~15ULL; u < e; u += 16) {
        const uint4 v = *reinterpret_cast<const uint4 *>(text + u);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const uint64_t b = u + 4 * i;                      // bytes [lo, hi) of this word belong to the inspected part
          const uint32_t lo = a > b ? (uint32_t)(a - b > 4 ? 4 : a - b) : 0u;
          const uint32_t hi = e > b ? (uint32_t)(e - b > 4 ? 4 : e - b) : 0u;
          const uint32_t valid = shk_bytes_below(hi) & ~shk_bytes_below(lo);
          const uint32_t x = (w[i] ^ 0x4E4E4E4Eu) | ~valid;   // zero byte = an 'N' that counts
          any |= (x - 0x01010101u) & ~x & 0x80808080u;
        }
      }
      if (!any) cnt = len - k + 1;
      else {
        uint32_t s0 = 0;
        while (len >= s0 + k) {
          uint32_t e0 = s0 + k;
          while (e0 < len && rd[e0] != 'N') e0++;
          cnt += e0 - s0 - k + 1;
          if (e0 == len) break;
          s0 = e0 + 1;
        }
      }
    }
    nkeys[r] = cnt;
  }
}

// ---------------------------------------------------------------- hash
// word = key | chunk << hb, key = min(fh, rh) mod 2^hb (CQF_mt.h:636-637, gqf.c:2230).
__global__ void k_hash_reads(const uint8_t *text, const uint64_t *rd_start, const uint64_t *rd_end,
                             const uint64_t *nreads_p, const uint16_t *rd_chunk,
                             uint32_t chunk_first, uint32_t chunk_mul, const uint64_t *key_base, uint32_t k, uint32_t hb,
                             uint64_t *words, uint64_t cap, uint32_t *err, uint64_t *hist0, uint32_t dig_shift, uint32_t dig_bits,
                             uint64_t q_lo, uint32_t ng_log2) {
  // hist0 != null: also count the keys per first partition digit (digit = (region >> dig_shift) & (2^dig_bits - 1)),
  // which saves the partition its first pass over the keys; with ng_log2 > 0 per (digit, window group of the key's position):
  // ShkRpLevel::ng_log2 (dig_bits + ng_log2 <= 10)
  __shared__ uint32_t lh0[1024];
  if (hist0) {
    for (uint32_t d = threadIdx.x; d < (1u << (dig_bits + ng_log2)); d += blockDim.x) lh0[d] = 0;
    __syncthreads();
  }
  __shared__ uint64_t ringG[SHK_HASH_WAVES][256];  // launched with at most SHK_HASH_WAVES waves per group
  __shared__ uint64_t ringH[SHK_HASH_WAVES][256];
  const uint64_t nreads = *nreads_p;
  const unsigned lane = shk_lane(), wv = shk_wave();
  const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x / SHK_WAVE);
  const uint64_t mask = hb >= 64 ? ~0ULL : ((1ULL << hb) - 1);
  uint64_t *rg = ringG[wv], *rh_ = ringH[wv];
  // Strips are 64 bases wide, so base j of a segment always sits in lane j & 63: every rotation
  // amount of the closed form is a per-lane constant. Seeds are pre-rotated once per lane.
  const uint64_t fA = shk_ror64(SHK_SEED_A, lane), fC = shk_ror64(SHK_SEED_C, lane), fG = shk_ror64(SHK_SEED_G, lane),
                 fT = shk_ror64(SHK_SEED_T, lane);
  const uint64_t rA = shk_rol64(SHK_SEED_T, lane), rC = shk_rol64(SHK_SEED_G, lane), rG = shk_rol64(SHK_SEED_C, lane),
                 rT = shk_rol64(SHK_SEED_A, lane);   // complement seeds (nthash.hpp:15,299)
  const unsigned rot_f = lane;                        // (k - 1 + p) & 63 with p = j + 1 - k
  const unsigned rot_r = (lane + 1 + 64 * 4 - k) & 63;  // p & 63 (k <= 191 < 256)
  uint64_t r = (uint64_t)blockIdx.x * (blockDim.x / SHK_WAVE) + wv;
  // two reads ahead for the table entries, one read ahead for the bases (see k_count_keys)
  uint64_t st_n = 0, en_n = 0, kb_n = 0, st_nn = 0, en_nn = 0, kb_nn = 0;
  uint32_t ch_n = 0, ch_nn = 0;
  uint32_t pc[4], pc_n[4];
  if (r < nreads) { st_n = rd_start[r]; en_n = rd_end[r]; kb_n = key_base[r]; ch_n = rd_chunk[r]; }
  if (r + nwaves < nreads) { st_nn = rd_start[r + nwaves]; en_nn = rd_end[r + nwaves]; kb_nn = key_base[r + nwaves]; ch_nn = rd_chunk[r + nwaves]; }
  shk_read_bytes(text, st_n, en_n, lane, pc_n);
  for (; r < nreads; r += nwaves) {
    const uint64_t st = st_n, en = en_n;
    uint64_t out = kb_n;
    const uint64_t chunk_tag = (uint64_t)(chunk_first + ch_n * chunk_mul) << hb;
#pragma unroll
    for (int t = 0; t < 4; t++) pc[t] = pc_n[t];
    st_n = st_nn; en_n = en_nn; kb_n = kb_nn; ch_n = ch_nn;
    if (r + nwaves < nreads) shk_read_bytes(text, st_n, en_n, lane, pc_n);
    if (r + 2 * nwaves < nreads) {
      st_nn = rd_start[r + 2 * nwaves]; en_nn = rd_end[r + 2 * nwaves]; kb_nn = key_base[r + 2 * nwaves]; ch_nn = rd_chunk[r + 2 * nwaves];
    }
    if (en - st > SHK_MAX_READ) continue;
    const uint32_t len = (uint32_t)(en - st);
    if (len < k) continue;
    const uint8_t *rd = text + st;
    unsigned long long nm[4];
    shk_read_masks(pc, nm);
    uint32_t s = 0;
    while (len >= s + k) {
      const uint32_t e = shk_segment_end_pre(nm, rd, len, s, k);
      const uint32_t L = e - s;             // bases in this segment
      const uint32_t nk = L - k + 1;        // its k-mers
      if (out + nk > cap) {
        if (lane == 0) atomicOr(err, SHK_E_KEYS_FULL);
        break;
      }
      uint64_t carryG = 0, carryH = 0;
      if (lane == 0) { rg[0] = 0; rh_[0] = 0; }
      for (uint32_t t = 0; t * 64 < L; t++) {
        const uint32_t j = t * 64 + lane;   // index in the segment
        uint64_t a = 0, c = 0;
        if (j < L) {
          // the first segment's first four strips are already in registers
          const uint32_t raw = (s == 0 && t < 4) ? (t == 0 ? pc[0] : t == 1 ? pc[1] : t == 2 ? pc[2] : pc[3]) : (uint32_t)rd[s + j];
          // seedTab (nthash.hpp:120-153): the bases in either case and the bytes 1 3 4 7; the complement strand's
          // seed is seedTab[byte & 7] for EVERY byte (nthash.hpp:299), base or not
          const unsigned b8 = raw & 0xFF, ch = b8 & 0xDF, lo = b8 & 7;
          a = (ch == 'A' || b8 == 4) ? fA : (ch == 'C' || b8 == 7) ? fC : (ch == 'G' || b8 == 3) ? fG : (ch == 'T' || b8 == 1) ? fT : 0ULL;
          c = lo == 1 ? rA : lo == 3 ? rC : lo == 7 ? rG : lo == 4 ? rT : 0ULL;
        }
        const uint64_t G1 = shk_wave_incl_xor64(a) ^ carryG;  // G(j+1)
        const uint64_t H1 = shk_wave_incl_xor64(c) ^ carryH;  // H(j+1)
        carryG = shk_last_lane64(G1);
        carryH = shk_last_lane64(H1);
        if (j < L) { rg[(j + 1) & 255] = G1; rh_[(j + 1) & 255] = H1; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the k-mer that ends at base j starts at p = j + 1 - k
        if (j < L && j + 1 >= k) {
          const uint32_t p = j + 1 - k;
          const uint64_t G0 = rg[p & 255], H0 = rh_[p & 255];
          const uint64_t fh = shk_rol64(G1 ^ G0, rot_f);
          const uint64_t rv = shk_ror64(H1 ^ H0, rot_r);
          const uint64_t hv = fh < rv ? fh : rv;
          words[out + p] = (hv & mask) | chunk_tag;
          if (hist0) {   // the digit exactly as k_rp_scatter computes it (shk_word_region)
            const uint32_t reg = (uint32_t)((((hv & mask) >> 8) - q_lo) >> SHK_REGION_LOG2);
            atomicAdd(&lh0[(((reg >> dig_shift) & ((1u << dig_bits) - 1)) << ng_log2) | ((uint32_t)((out + p) >> SHK_RP_TILE0_LOG2) & ((1u << ng_log2) - 1))], 1u);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      out += nk;
      if (e == len) break;
      s = e + 1;
    }
  }
  if (hist0) {
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < (1u << (dig_bits + ng_log2)); d += blockDim.x)
      if (lh0[d]) atomicAdd((unsigned long long *)&hist0[d], (unsigned long long)lh0[d]);
  }
}
