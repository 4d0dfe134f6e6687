// Device-side helpers shared by the kernels: wave/workgroup scans, unaligned
// access to the packed 89-byte qfblock image, the counter codec.
// gfx950 only: wavefront = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SHK_WAVE 64
#define SHK_MAX_WAVES 16      // workgroups are at most 1024 threads
#define SHK_BLOCK_BYTES 89    // reference qfblock at bits_per_slot = 8: cqf/gqf.c:63-86
#define SHK_OFF_OCC 1         // occupieds word
#define SHK_OFF_RUN 9         // runends word
#define SHK_OFF_TRAV 17       // traveled word
#define SHK_OFF_SLOTS 25      // 64 one-byte slots

// One region = the quotients one workgroup owns in the merge kernels.
#define SHK_REGION_LOG2 8
#define SHK_REGION (1u << SHK_REGION_LOG2)             // 256 quotients = 4 blocks, rebuilt by ONE wave
#define SHK_REGION_BLOCKS (SHK_REGION / 64)
#define SHK_IMG_BLOCKS 24                              // LDS image: own 4 blocks + 20 spill blocks
#define SHK_IMG_SLOTS (SHK_IMG_BLOCKS * 64)            // 1536 slots
#define SHK_IMG_BLOCKS_BIG 64                          // retry size for longer clusters (4096 slots)
#define SHK_IMG_BYTES (SHK_IMG_BLOCKS * SHK_BLOCK_BYTES)
#define SHK_HCAP_LOG2 9
#define SHK_HCAP (1u << SHK_HCAP_LOG2)                 // LDS hash capacity (distinct new keys per region)
#define SHK_MERGE_THREADS 64                           // the rebuild kernels run one wave per region ...
#define SHK_MERGE_GROUP 128                            // ... helped by one more wave while staging and folding the batch keys (the kernel is bound by instruction issue: more helpers only add instructions)
#define SHK_CHUNK_BITS 12                              // chunk index field of a key word
#define SHK_MAX_CHUNKS (1u << SHK_CHUNK_BITS)
#define SHK_HIST_BINS 32

// error bits raised by kernels (ctx->d_err), reported through the C ABI
#define SHK_E_OLD_EXTENT   (1u << 0)   // a region's old runs spill further than the LDS image
#define SHK_E_NEW_EXTENT   (1u << 1)   // a region's new runs spill further than the LDS image
#define SHK_E_HASH_FULL    (1u << 2)   // more distinct new keys in one region than SHK_HCAP
#define SHK_E_TABLE_FULL   (1u << 3)   // runs would pass xnslots
#define SHK_E_CORRUPT      (1u << 4)   // occupieds/runends disagree
#define SHK_E_BAD_FASTQ    (1u << 5)   // read longer than 65535 / k out of range
#define SHK_E_KEYS_FULL    (1u << 6)   // batch produced more keys than the key buffer holds
#define SHK_E_RUN_TOO_LONG (1u << 7)
#define SHK_E_LOOKBACK     (1u << 8)   // single-launch rebuild gave up waiting for a predecessor (host falls back)
#define SHK_E_FUSED        (1u << 9)   // the one-pass deNoise point met a region it does not handle (host takes the three-pass path)
#define SHK_E_SLOT_FULL    (1u << 10)  // last partition level with fixed-capacity region slots: a region got more words (host redoes the level with exact bases)

__device__ __forceinline__ unsigned shk_lane() { return threadIdx.x & (SHK_WAVE - 1); }
__device__ __forceinline__ unsigned shk_wave() { return threadIdx.x / SHK_WAVE; }

__device__ __forceinline__ uint64_t shk_rol64(uint64_t v, unsigned s) {
  s &= 63;
  return s ? (v << s) | (v >> (64 - s)) : v;
}
__device__ __forceinline__ uint64_t shk_ror64(uint64_t v, unsigned s) {
  s &= 63;
  return s ? (v >> s) | (v << (64 - s)) : v;
}

// keys per window of the FIRST partition level: 16384 = what one workgroup can stage in LDS. Its 128 output streams lie
// pages apart and three of four digit-run stores of a 4096-key window missed the first-level TLB (19.5 M misses per
// launch against 30 k in the later levels, whose streams stay inside one bucket): longer runs = fewer misses per key.
// Measured on 832 M keys: 4.7 ms (4096) -> 3.6 (8192) -> 2.7 ms (16384). The later levels keep 4096-key windows
// (SHK_RP_TILE): 8192 made them slower.
#define SHK_RP_TILE0_LOG2 14

// Workgroup barrier that orders LDS accesses only: global stores issued before it may still be in flight behind it (a
// __syncthreads() waits for them too: vmcnt(0)). For the scatter kernels, whose windows reuse LDS buffers while the runs
// they have just written drain to HBM.
__device__ __forceinline__ void shk_lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#else
  __syncthreads();
#endif
}

// ---- wave scans (64 lanes, shuffle based)
// LDS writes of this wave's lanes become visible to its other lanes (a barrier among the 64 lanes only)
__device__ __forceinline__ void shk_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Inclusive scans over the 64 lanes with DPP moves (VALU only: __shfl_up would be a ds_bpermute per step, and the LDS
// pipeline is what the rebuild kernel keeps busiest): Hillis-Steele inside each row of 16 lanes (row_shr 1,2,4,8), then
// row_bcast15 / row_bcast31 carry the row totals across rows (GFX9 DPP modes, present on gfx950). A lane without a source
// (or in a row the step does not touch) receives the operation's identity.
#define SHK_DPP_STEPS(STEP) STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)
__device__ __forceinline__ uint32_t shk_wave_incl_add(uint32_t x) {
#define SHK_STEP_(ctrl, rows) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rows, 0xf, true);
  SHK_DPP_STEPS(SHK_STEP_)
#undef SHK_STEP_
  return x;
}
__device__ __forceinline__ uint64_t shk_wave_incl_add64(uint64_t x) {
#define SHK_STEP_(ctrl, rows) { const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, ctrl, rows, 0xf, true); \
                              const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), ctrl, rows, 0xf, true); \
                              x += ((uint64_t)hi_ << 32) | lo_; }
  SHK_DPP_STEPS(SHK_STEP_)
#undef SHK_STEP_
  return x;
}
// Inclusive XOR scan over the 64 lanes with DPP moves (no LDS crossbar): Hillis-Steele inside
// each row of 16 lanes (row_shr 1,2,4,8), then row_bcast15 / row_bcast31 carry the row totals
// across rows (GFX9 DPP modes, present on gfx950).
__device__ __forceinline__ uint32_t shk_dpp_xor_scan32(uint32_t x) {
  x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);  // row_shr:1
  x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);  // row_shr:2
  x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);  // row_shr:4
  x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);  // row_shr:8
  x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, true);  // row_bcast:15 -> rows 1,3
  x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, true);  // row_bcast:31 -> rows 2,3
  return x;
}
__device__ __forceinline__ uint64_t shk_wave_incl_xor64(uint64_t x) {
  const uint32_t lo = shk_dpp_xor_scan32((uint32_t)x), hi = shk_dpp_xor_scan32((uint32_t)(x >> 32));
  return ((uint64_t)hi << 32) | lo;
}
// value of lane 63, as a wave-uniform scalar
__device__ __forceinline__ uint64_t shk_last_lane64(uint64_t x) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, 63);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), 63);
  return ((uint64_t)hi << 32) | lo;
}

// ---- workgroup exclusive scan of one u32 per thread. `scratch` is SHK_MAX_WAVES+1 words of LDS.
// Returns the exclusive prefix; *total gets the workgroup sum. Ends with a barrier.
__device__ __forceinline__ uint32_t shk_block_exscan(uint32_t v, uint32_t *total, uint32_t *scratch) {
  unsigned lane = shk_lane(), wave = shk_wave(), nw = blockDim.x / SHK_WAVE;
  uint32_t incl = shk_wave_incl_add(v);
  if (lane == SHK_WAVE - 1) scratch[wave] = incl;
  __syncthreads();
  if (wave == 0) {
    uint32_t w = lane < nw ? scratch[lane] : 0;
    uint32_t wi = shk_wave_incl_add(w);
    if (lane < nw) scratch[lane] = wi - w;
    if (lane == nw - 1) scratch[SHK_MAX_WAVES] = wi;
  }
  __syncthreads();
  uint32_t res = scratch[wave] + incl - v;
  *total = scratch[SHK_MAX_WAVES];
  __syncthreads();
  return res;
}
__device__ __forceinline__ uint64_t shk_block_exscan64(uint64_t v, uint64_t *total, uint64_t *scratch) {
  unsigned lane = shk_lane(), wave = shk_wave(), nw = blockDim.x / SHK_WAVE;
  uint64_t incl = shk_wave_incl_add64(v);
  if (lane == SHK_WAVE - 1) scratch[wave] = incl;
  __syncthreads();
  if (wave == 0) {
    uint64_t w = lane < nw ? scratch[lane] : 0;
    uint64_t wi = shk_wave_incl_add64(w);
    if (lane < nw) scratch[lane] = wi - w;
    if (lane == nw - 1) scratch[SHK_MAX_WAVES] = wi;
  }
  __syncthreads();
  uint64_t res = scratch[wave] + incl - v;
  *total = scratch[SHK_MAX_WAVES];
  __syncthreads();
  return res;
}
// workgroup sum of one u64 per thread (result valid in every thread). Ends with a barrier.
__device__ __forceinline__ uint64_t shk_block_sum64(uint64_t v, uint64_t *scratch) {
  uint64_t t;
  shk_block_exscan64(v, &t, scratch);
  return t;
}

// ---- "free pointer" functions f -> max(f + a, b): how a sequence of runs moves the
// first free slot (a = total run length, b = where it ends when nothing spills in).
// Composition is associative; identity is (0, -inf).
#define SHK_NEG_INF (-(1LL << 60))
struct ShkMP {
  long long a, b;
};
__device__ __forceinline__ ShkMP shk_mp_compose(ShkMP first, ShkMP second) {
  ShkMP r;
  r.a = first.a + second.a;
  long long t = first.b + second.a;
  r.b = t > second.b ? t : second.b;
  if (r.b < SHK_NEG_INF) r.b = SHK_NEG_INF;
  return r;
}
__device__ __forceinline__ long long shk_mp_apply(ShkMP m, long long f) {
  long long t = f + m.a;
  return t > m.b ? t : m.b;
}

// the same in 32 bits, for values relative to a region start (k_region_merge: a <= slots of one region's runs, b <= 256 + a)
#define SHK_NEG_INF_W (-(1 << 28))
struct ShkMPw {
  int a, b;
};
__device__ __forceinline__ ShkMPw shk_mpw_compose(ShkMPw first, ShkMPw second) {
  ShkMPw r;
  r.a = first.a + second.a;
  const int t = first.b + second.a;
  r.b = t > second.b ? t : second.b;
  if (r.b < SHK_NEG_INF_W) r.b = SHK_NEG_INF_W;
  return r;
}

// inclusive scan of f -> max(f + a, b) functions over the 64 lanes in lane order (lane 0's function is applied first)
__device__ __forceinline__ ShkMPw shk_mpw_wave_scan(ShkMPw x) {
#define SHK_STEP_(ctrl, rows) { ShkMPw y_; y_.a = __builtin_amdgcn_update_dpp(0, x.a, ctrl, rows, 0xf, true); \
                              y_.b = __builtin_amdgcn_update_dpp(SHK_NEG_INF_W, x.b, ctrl, rows, 0xf, false); \
                              x = shk_mpw_compose(y_, x); }
  SHK_DPP_STEPS(SHK_STEP_)
#undef SHK_STEP_
  return x;
}

// ---- bytes of the packed block image (global or LDS): unaligned little-endian access
__device__ __forceinline__ uint64_t shk_ld64(const uint8_t *p) {
  uint64_t v = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i);
  return v;
}
__device__ __forceinline__ void shk_st64(uint8_t *p, uint64_t v) {
#pragma unroll
  for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i));
}

// position of the n-th (0-based) set bit of w; 64 when there are not that many
__device__ __forceinline__ unsigned shk_select64(uint64_t w, unsigned n) {
  if ((unsigned)__popcll(w) <= n) return 64;
  unsigned pos = 0;
#pragma unroll
  for (int width = 32; width >= 1; width >>= 1) {
    uint64_t low = w & ((1ULL << width) - 1);
    unsigned c = (unsigned)__popcll(low);
    if (n >= c) { n -= c; w >>= width; pos += width; } else { w = low; }
  }
  return pos;
}

// ---- counter codec, cqf/gqf.c:1225-1255 (encode_counter) at bits_per_slot = 8:
// count 1 -> [r]; count c+1 >= 2 -> [r, (0 if top digit > r), d_m|0x80 .. d_1|0x80, d_0],
// base-128 digits of c.
__device__ __forceinline__ unsigned shk_enc_len(unsigned rem, uint64_t count) {
  if (count <= 1) return (unsigned)count;
  uint64_t c = count - 1;
  unsigned nd = 1;
  uint64_t t = c >> 7;
  while (t) { nd++; t >>= 7; }
  unsigned top = (unsigned)((c >> (7 * (nd - 1))) & 0x7f);
  if (nd > 1) top |= 0x80;
  return 1 + nd + (top > rem ? 1 : 0);
}
// writes the encoding to dst[0..len) and returns len
__device__ __forceinline__ unsigned shk_enc_write(uint8_t *dst, unsigned rem, uint64_t count) {
  dst[0] = (uint8_t)rem;
  if (count <= 1) return 1;
  uint64_t c = count - 1;
  unsigned nd = 1;
  uint64_t t = c >> 7;
  while (t) { nd++; t >>= 7; }
  unsigned top = (unsigned)((c >> (7 * (nd - 1))) & 0x7f);
  if (nd > 1) top |= 0x80;
  unsigned n = 1;
  if (top > rem) dst[n++] = 0;
  for (int i = (int)nd - 1; i >= 1; i--) dst[n++] = (uint8_t)(((c >> (7 * i)) & 0x7f) | 0x80);
  dst[n++] = (uint8_t)(c & 0x7f);
  return n;
}
// decode_counter, cqf/gqf.c:1259-1299, over a byte image of slots. `is_last` = the
// entry's first slot carries the runend bit. Returns the number of slots the entry takes.
__device__ __forceinline__ unsigned shk_dec(const uint8_t *s, bool first_is_runend, uint64_t *count) {
  unsigned rem = s[0];
  if (first_is_runend) { *count = 1; return 1; }
  unsigned digit = s[1];
  if (digit > rem) { *count = 1; return 1; }
  unsigned n = 1;
  uint64_t cnt = 0;
  if (digit == 0) { n++; digit = s[n]; }
  while (digit & 0x80) { cnt = cnt * 128 + (digit & 0x7f); n++; digit = s[n]; }
  cnt = cnt * 128 + digit;
  *count = cnt + 1;
  return n + 1;
}
