// score.h — boolean queries (Or / And / min-match) as doc tiles accumulated in LDS.
//
//   k_items_count / k_items_fill   per (unit, doc tile): the list of (term, block) WORK
//                      ITEMS that reach the tile, as 32-byte records in global memory laid
//                      out per consuming wavefront
//   k_pilot            scores every P-th doc tile, derives a per-query score-bin threshold
//   k_score            decode + score + accumulate every doc tile in LDS, emit the
//                      candidates above the threshold, count hits
//
// Replaces block_disjunction::refill (disjunction.hpp:1240-1351), basic_disjunction
// (:204-358), min_match (:56-78, 1340-1342) and, for min-match / dense conjunctions, the
// per-doc match counting of Conjunction (conjunction.hpp:207-223).
//
// A workgroup owns one doc tile [lo, lo+TILE) of one query at a time.  Per-doc score
// accumulators live in LDS as FIXED-POINT integers (ACC = 64-bit, or 32-bit when the query's
// score range allows it): integer addition is associative, so the work items of a tile are
// processed by the wavefronts in any order, with no barrier between terms, and the sum is
// still bit-reproducible.  Each posting's score follows the reference's float expression;
// the fixed-point sum differs from the reference's sequential float sum by rounding only.
//
// The kernels are bound by instruction issue (VALU + SALU), not by HBM, LDS or latency
// (rocprofv3 PMC, profiles/): everything is written to minimise wave-instructions per
// 128-posting block.  Since round 2 nothing on the hot path goes through a FLAT load or a
// v_readlane any more: a work item is read by ONE scalar load (s_load_dwordx8: payload
// address, bit widths, doc base, scorer constants — all precomputed by k_items_fill), the
// payload by saddr global loads.
#pragma once
#include <cstddef>

#include "kernels.h"

namespace irs_hip {

// ------------------------------------------------------------ work items --

enum : uint32_t {
  kItemTermMask = 0x1Fu,    // aux bits 0..4: term slot of the query
  kItemSlow = 1u << 8,      // any block framing / any scorer: decoded from `.doc` by the generic path
  kItemTail = 1u << 9,      // the term's decoded vint tail / single doc (always with kItemSlow)
  kItemSqrt = 1u << 10,     // square-root score form (TF-IDF family), else reciprocal form
  kItemPair = 1u << 11,     // even item of a wavefront: it and the next one run as a fused pair
  kItemSolo = 1u << 12,     // even item of a wavefront: there is no next one
  kItemTable = 1u << 13,    // score = cs * T[tf][norm]: every tf of the block has a table row
  kItemFreqShift = 16,      // aux bits 16..31: added to every extracted frequency — the value
                            // of an ALL-EQUAL freq block (fbits == 0 extracts zeros), else 0
};
constexpr uint32_t kItemSlack = 8;     // readable records behind the last list (look-ahead loads)

struct alignas(32) ItemG {
  uint64_t addr;    // straight-line item: the doc payload in the packed-payload image (the freq
                    // payload follows 16*dbits bytes later); generic item: the block in `.doc`;
                    // tail item: some readable address (its look-ahead loads are never used)
  uint32_t dbits;   // bit widths of the doc / freq parts (block header bytes; fbits 0 = all
  uint32_t fbits;   // frequencies equal, see kItemFreqShift; tail item: 1, 1)
  uint32_t base;    // last doc of the preceding block MINUS the tile's first doc (mod 2^32):
                    // tile-relative doc = base + prefix sum of the deltas; tail item: postings
  float cs;         // the term's c0 pre-multiplied by the fixed-point scale
  uint32_t tab;     // LDS byte offset of the term's 256-entry table row; tail item: first row
                    // of the term in DevSegment::tail_docs / tail_freqs
  uint32_t aux;     // term slot | kItem* flags
};
static_assert(sizeof(ItemG) == 32, "one s_load_dwordx8 per work item");

struct alignas(16) ItemAddr {   // the first 16 bytes of an ItemG: all a payload request needs
  uint64_t addr;
  uint32_t dbits;
  uint32_t fbits;
};
struct alignas(8) ItemCalc {    // bytes 8..31 of an ItemG: all the decode + score stage needs
  uint32_t dbits;
  uint32_t fbits;
  uint32_t base;
  float cs;
  uint32_t tab;
  uint32_t aux;
};

// ---------------------------------------------------------------- scorers --
//
// Scorers of the straight-line path are all evaluated from one 256-entry table row `tab` in
// LDS (indexed by the doc's norm byte), in one of two forms:
//   reciprocal form  score = c0 - c0 / (1 + tf * tab[norm])
//     BM25, 1-byte norms  tab[n] = norm_cache[n] = 1/(norm_const + norm_length*n), [0] = 0
//                         (bm25.cpp:348-353, 404-409)
//     BM25, no norms      tab[n] = 1/(norm_const + norm_length)   (norm == 1, bm25.cpp:487-489)
//     BM15                tab[n] = 1/norm_const                    (bm25.cpp:313)
//   square-root form score = sqrt(tf) * c0 * tab[norm]
//     TF-IDF              tab[n] = 1                               (tfidf.cpp:185-187)
//     TF-IDF with norms   tab[n] = 1/sqrt(n), [0] = 0              (tfidf.cpp:251-253)
// Rows that ignore the norm are constant, so whatever byte the norm stage reads is fine.
//
// Most blocks only hold small frequencies (the bit width of the freq part says so), and for
// those the whole per-posting expression is tabulated as well: a table slot is R rows
//   row 0        tab[n]                      (above; general frequencies: one v_rcp / v_sqrt)
//   row r >= 1   T_r[n] = 1 - 1/(1 + r*tab[n])   resp.   sqrt(r) * tab[n]
// so that a posting with tf < R costs one table read and one multiply-add: score = c0 * T_tf[norm].
// R = kTableRows / (slots rounded up to a power of two): 16 rows with one slot (the usual
// case: one (norm_const, norm_length) per query), 8 with two, 4 with three or four.
constexpr uint32_t kTableRows = 16;   // 256-entry rows of floats in LDS: 16 KB
__host__ __device__ __forceinline__ uint32_t table_rows(uint32_t n_caches) {
  return n_caches <= 1u ? 16u : (n_caches <= 2u ? 8u : 4u);
}
__host__ __device__ __forceinline__ bool table_kind(int32_t kind) {
  return kind == kBM25Tiny || kind == kBM25One || kind == kBM15 || kind == kTfidf ||
         kind == kTfidfTiny;
}
__host__ __device__ __forceinline__ bool sqrt_kind(int32_t kind) {
  return kind == kTfidf || kind == kTfidfTiny;
}
__device__ __forceinline__ float table_value(int32_t kind, float nc, float nl, uint32_t n) {
  switch (kind) {
    case kBM25Tiny: return n ? 1.f / (nc + nl * static_cast<float>(n)) : 0.f;
    case kBM25One: return 1.f / (nc + nl * 1.f);
    case kBM15: return 1.f / nc;
    case kTfidf: return 1.f;
    default: return n ? 1.f / sqrtf(static_cast<float>(n)) : 0.f;  // kTfidfTiny
  }
}

// entry [r][n] of a scorer's table: row 0 holds the norm factor, row r > 0 the whole expression at
// frequency r (build_tables fills the LDS tables with it; k_join_rescore evaluates it per posting:
// one function, the same float)
__device__ __forceinline__ float table_entry(int32_t kind, float nc, float nl, uint32_t r, uint32_t n) {
  const float t = table_value(kind, nc, nl, n);
  if (r == 0) return t;
  if (sqrt_kind(kind)) return sqrtf(static_cast<float>(r)) * t;
  return 1.f - 1.f / (1.f + static_cast<float>(r) * t);
}

template<typename ACC>
struct TileSmemT {
  ACC* acc;          // [TILE + 64] fixed-point score accumulators (score_buf of
                     // block_disjunction, disjunction.hpp:1087-1092, widened to TILE docs)
                     // + one private dummy slot per lane
  uint32_t* cnt;     // [(TILE + 64)/4] per-doc match counters, 1 byte each (AND only; + dummies)
  uint8_t* lnorm;    // [TILE] Norm2 bytes of the tile
  float* caches;     // [kTableRows][256] table rows (BM25Stats::norm_cache and friends)
  DevQTerm* qts;     // [kMaxTerms] the query's term scorers (generic path)
  uint32_t* slow;    // [4] what else the generic path needs: fx_mul (float bits), num_docs,
                     // rows per table slot
};

// Byte offsets of the tile arrays inside the workgroup's LDS block (== carve() below);
// the hot path addresses them absolutely (wave::lds_*).
template<typename ACC, int TILE, bool AND>
struct TileOff {
  static constexpr uint32_t acc = 0;
  static constexpr uint32_t cnt = uint32_t(sizeof(ACC)) * (TILE + 64);
  static constexpr uint32_t lnorm = cnt + (AND ? TILE + 64 : 0);
  static constexpr uint32_t caches = lnorm + TILE;
  static constexpr uint32_t qts = caches + 4u * 256u * kTableRows;
  static constexpr uint32_t slow = qts + uint32_t(sizeof(DevQTerm)) * kMaxTerms;
  static constexpr uint32_t end = slow + 16u;
};

template<typename ACC, int TILE, bool AND>
__device__ __forceinline__ TileSmemT<ACC> carve(unsigned char* smem, unsigned char** rest) {
  using Off = TileOff<ACC, TILE, AND>;
  TileSmemT<ACC> sm;
  sm.acc = reinterpret_cast<ACC*>(smem + Off::acc);
  sm.cnt = reinterpret_cast<uint32_t*>(smem + Off::cnt);
  sm.lnorm = smem + Off::lnorm;
  sm.caches = reinterpret_cast<float*>(smem + Off::caches);
  sm.qts = reinterpret_cast<DevQTerm*>(smem + Off::qts);
  sm.slow = reinterpret_cast<uint32_t*>(smem + Off::slow);
  *rest = smem + Off::end;
  return sm;
}

// One thread per table entry: the rows of slot c come from the first term using it.
template<typename SM>
__device__ __forceinline__ void build_tables(const SM& sm, uint32_t n_caches, uint32_t n_terms) {
  const uint32_t rows = table_rows(n_caches);
  for (uint32_t e = threadIdx.x; e < n_caches * rows * 256u; e += blockDim.x) {
    const uint32_t c = (e >> 8) / rows, r = (e >> 8) % rows, n = e & 255u;
    float v = 0.f;
    for (uint32_t j = 0; j < n_terms; ++j) {
      if (sm.qts[j].cache_id == c) {
        v = table_entry(sm.qts[j].kind, sm.qts[j].norm_const, sm.qts[j].norm_length, r, n);
        break;
      }
    }
    sm.caches[e] = v;
  }
}

// A score already multiplied by DevQuery::fx_mul (a power of two) -> fixed point.
// 64-bit: x < 2^29 is the HIGH word, the fraction becomes the low word (2^E units,
// E = 61 - ceil(log2 U)).  32-bit: x < 2^30 truncated (2^(30-e) units).  `| 1`
// keeps every posting's contribution non-zero: "accumulator != 0" == "matched".
template<typename ACC>
__device__ __forceinline__ ACC fixed_from_scaled(float x);
template<>
__device__ __forceinline__ unsigned long long fixed_from_scaled<unsigned long long>(float x) {
  const uint32_t hi = static_cast<uint32_t>(x);   // truncates
  const float rem = x - static_cast<float>(hi);   // exact
  const uint32_t lo = static_cast<uint32_t>(rem * 4294967296.f);
  return ((static_cast<unsigned long long>(hi) << 32) | lo) | 1ull;
}
template<>
__device__ __forceinline__ uint32_t fixed_from_scaled<uint32_t>(float x) {
  return static_cast<uint32_t>(x) | 1u;
}
template<typename ACC>
__device__ __forceinline__ float from_fixed(ACC a, float fx_inv) {
  return static_cast<float>(a) * fx_inv;
}

// DevQuery::op: bits 0..7 how the unit runs (0 doc tiles, 1 doc tiles with match counters,
// 2 block-driven conjunction), 8..15 the matches a doc needs, 16..17 the boolean filter's
// ScoreMergeType (scorer.hpp:224-236: 0 sum, 1 max, 2 min), bit 18: a kMin disjunction of two
// sub-iterators — a doc only one of them holds scores 0 (disjunction.hpp:338-351).
enum : uint32_t { kScoreSum = 0, kScoreMax = 1, kScoreMin = 2 };
__host__ __device__ __forceinline__ uint32_t query_merge(int32_t op) { return (uint32_t(op) >> 16) & 3u; }
__host__ __device__ __forceinline__ uint32_t query_need(int32_t op) { return (uint32_t(op) >> 8) & 0xFFu; }
__host__ __device__ __forceinline__ bool query_min_both(int32_t op) { return (uint32_t(op) >> 18) & 1u; }
// Max/Min merged accumulators hold max(fixed) resp. max(~fixed) (0 = untouched either way);
// back to the fixed-point score:
template<typename ACC>
__device__ __forceinline__ ACC merged_fixed(ACC a, uint32_t merge) {
  return merge == kScoreMin ? ACC(~a) : a;
}

// Score of one posting — the reference's float expressions, evaluated in the
// same order with no FMA contraction (bm25.cpp:313, 353, 359; tfidf.cpp:185-187, 251-253).
template<typename SM>
__device__ __forceinline__ float score_posting(const DevSegment& seg, const DevQTerm& qt,
                                               float inv_one, const SM& sm, uint32_t freq,
                                               uint32_t doc, uint32_t idx) {
  const float tf = static_cast<float>(freq);
  switch (qt.kind) {
    case kBM1:
      return qt.c0;
    case kBM15:
      return qt.c0 - qt.c0 / (1.f + tf / qt.norm_const);
    case kBM25Tiny: {
      const uint32_t n = sm.lnorm[idx];
      float inv;
      if (qt.cache_id < kMaxCaches) {
        inv = sm.caches[qt.cache_id * sm.slow[2] * 256u + n];
      } else {
        inv = n ? 1.f / (qt.norm_const + qt.norm_length * static_cast<float>(n)) : 0.f;
      }
      return qt.c0 - qt.c0 / (1.f + tf * inv);
    }
    case kBM25One:
      return qt.c0 - qt.c0 / (1.f + tf * inv_one);
    case kBM25Wide: {
      const float c1 = qt.norm_const +
                       qt.norm_length * static_cast<float>(norm_global(seg, doc));
      return qt.c0 - qt.c0 * c1 / (c1 + tf);
    }
    case kTfidf:
      return sqrtf(tf) * qt.c0;
    case kTfidfTiny: {
      const uint32_t n = sm.lnorm[idx];
      const float r = n ? 1.f / sqrtf(static_cast<float>(n)) : 0.f;
      return sqrtf(tf) * qt.c0 * r;
    }
    case kTfidfWide: {
      const uint32_t n = norm_global(seg, doc);
      const float r = n ? 1.f / sqrtf(static_cast<float>(n)) : 0.f;
      return sqrtf(tf) * qt.c0 * r;
    }
    case kBM25Legacy: {   // BM25NormAdapter<kNorm>: 1/stored; tf = kSQRT(freq)  bm25.cpp:242-249, 333-337
      const float c1 = qt.norm_const + qt.norm_length * (1.f / norm_legacy(seg, doc));
      return qt.c0 - qt.c0 * c1 / (c1 + sqrtf(tf));
    }
    default:              // kTfidfLegacy: the stored value as it is  tfidf.cpp:214-219
      return sqrtf(tf) * qt.c0 * norm_legacy(seg, doc);
  }
}

// generic scorer (every kind), used off the hot path; idx = doc - lo (wraps for doc < lo).
// Scorers of the table family use the ARITHMETIC OF THE STRAIGHT-LINE PATH (tile_post /
// tile_post_table: table row or v_rcp / v_sqrt form, chosen per TERM by its largest
// frequency, DevQTerm::pad1), so that a posting contributes the same fixed-point value
// whichever path decodes it — a block of the packed image, an odd framing, the vint tail — and
// the same as on the joined posting streams of join.h: results are bit-identical across them.
template<typename ACC, int TILE, bool AND>
__device__ __forceinline__ void tile_apply(const DevSegment& seg, const TileSmemT<ACC>& sm,
                                           const DevQTerm& qt, float inv_one, uint32_t idx,
                                           uint32_t freq, uint32_t lo, uint32_t span,
                                           float fx_mul) {
  if (idx < span) {
    ACC fx;
    if (table_kind(qt.kind) && qt.cache_id < kMaxCaches) {
      const uint32_t rows = sm.slow[2];
      const float* slot = sm.caches + qt.cache_id * rows * 256u;
      const uint32_t n = sm.lnorm[idx];
      const float cs = qt.c0 * fx_mul;
      if (qt.pad1 < rows) {
        const float t = slot[freq * 256u + n];
        if (sizeof(ACC) == 4) fx = static_cast<ACC>(static_cast<uint32_t>(wave::fma(cs, t, 1.f)));
        else fx = fixed_from_scaled<ACC>(cs * t);
      } else {
        const float inv = slot[n];
        const float tf = static_cast<float>(freq);
        const float scaled = sqrt_kind(qt.kind)
                                 ? wave::fast_sqrt(tf) * cs * inv
                                 : wave::fma(-cs, wave::fast_rcp(wave::fma(tf, inv, 1.f)), cs);
        fx = fixed_from_scaled<ACC>(scaled);
      }
    } else {
      const float s = score_posting(seg, qt, inv_one, sm, freq, lo + idx, idx);
      fx = fixed_from_scaled<ACC>(s * fx_mul);
    }
    const uint32_t merge = sm.slow[3];   // MaxMerger / MinMerger, scorer.hpp:399-423
    if (merge == kScoreSum) atomicAdd(&sm.acc[idx], fx);
    else atomicMax(&sm.acc[idx], merge == kScoreMin ? ACC(~fx) : fx);
    if (AND) atomicAdd(&sm.cnt[idx >> 2], 1u << (8u * (idx & 3u)));
  }
}

// Scoring of N postings at once on the hot path: any scorer of the table family
// (`tab[k]` is the LDS byte offset of posting k's table row, raw[k] its tile-relative doc).
// Reciprocal form, e.g. BM25 over 1-byte norms — bm25.cpp:348-353:
// c0 - c0/(1 + tf*norm_cache[norm]): the division is one v_rcp_f32 and the two
// multiply-adds are fused; square-root form (TF-IDF): one v_sqrt_f32 and two
// multiplies.  Either is within 2 ulp of the reference expression, far inside the
// 1e-5 parity tolerance; `cs` is c0 pre-multiplied by fx_mul so the result is
// already in fixed-point units.
// Staged so that the N norm-byte reads, then the N table reads, then the N LDS
// atomics are issued back to back: one LDS latency per stage instead of one per
// posting.  wave::keep*() pins each stage (the compiler would otherwise sink the
// whole computation behind a per-posting branch).
// Postings outside the tile are not branched around: `doc - lo` wraps to a huge
// value for doc < lo, and one v_min clamps every out-of-tile index to the lane's
// private dummy accumulator acc[TILE + lane]; whatever byte sits at
// lnorm[TILE + lane] (the next LDS array) yields some finite garbage that is
// added to that dummy slot, which nothing ever reads.  (The last tile of a
// segment needs no extra test: docs >= lo + span do not exist.)
template<typename ACC, int TILE, bool AND, int N>
__device__ __forceinline__ void tile_post(const TileSmemT<ACC>& sm, const float (&cs)[N],
                                          const uint32_t (&tab)[N], const uint32_t (&raw)[N],
                                          const uint32_t (&freq)[N], unsigned lane,
                                          bool sqrt_form) {
  using Off = TileOff<ACC, TILE, AND>;
  const unsigned char* base = reinterpret_cast<const unsigned char*>(sm.acc);  // LDS offset 0
  uint32_t idx[N], nb[N];
  float inv[N];
  ACC fx[N];
  const uint32_t dummy = uint32_t(TILE) + lane;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    idx[k] = raw[k] < dummy ? raw[k] : dummy;
    nb[k] = wave::lds_u8(base, Off::lnorm + idx[k]);
  }
  wave::keep_all(nb);   // one asm statement over all N values: one s_waitcnt
#pragma unroll
  for (int k = 0; k < N; ++k) inv[k] = wave::lds_f32(base, tab[k] + nb[k] * 4u);
  wave::keep_all_f(inv);
  if (sqrt_form) {   // wave-uniform
#pragma unroll
    for (int k = 0; k < N; ++k) {
      float scaled = wave::fast_sqrt(static_cast<float>(freq[k])) * cs[k] * inv[k];
      wave::keep_f(scaled);
      fx[k] = fixed_from_scaled<ACC>(scaled);
    }
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float r = wave::fast_rcp(wave::fma(static_cast<float>(freq[k]), inv[k], 1.f));
      float scaled = wave::fma(-cs[k], r, cs[k]);
      wave::keep_f(scaled);
      fx[k] = fixed_from_scaled<ACC>(scaled);
    }
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    wave::lds_add(base, Off::acc + idx[k] * uint32_t(sizeof(ACC)), fx[k]);
    // (out-of-tile postings bump a dummy counter byte, like their dummy accumulator)
    if (AND) wave::lds_add(base, Off::cnt + (idx[k] & ~3u), 1u << (8u * (idx[k] & 3u)));
  }
}

// The two payload words lane `lane` needs for values 2*lane, 2*lane+1 of a part packed with
// `bits` bits that starts `skip16` 16-byte units behind `addr` (0: the doc part; dbits: the
// freq part right behind it).  saddr global loads: `addr` is wave-uniform.  No branch on the
// bit width; reads at most 24 bytes past the part.
// The same for postings whose frequencies all have a table row (kItemTable): score =
// cs * T_tf[norm] — `tab[k]` is the offset of row 0 of the term's slot (all-equal freq
// blocks: of the row of their one frequency, and the extracted freq is 0).  The `+ 1.0` of
// the multiply-add keeps every contribution non-zero ("accumulator != 0" == "matched")
// without a separate OR; it costs at most the one unit per posting the truncation may lose
// anyway.
template<typename ACC, int TILE, bool AND, int N>
__device__ __forceinline__ void tile_post_table(const TileSmemT<ACC>& sm, const float (&cs)[N],
                                                const uint32_t (&tab)[N],
                                                const uint32_t (&raw)[N],
                                                const uint32_t (&freq)[N], unsigned lane) {
  using Off = TileOff<ACC, TILE, AND>;
  const unsigned char* base = reinterpret_cast<const unsigned char*>(sm.acc);  // LDS offset 0
  uint32_t idx[N], nb[N];
  float t[N];
  ACC fx[N];
  const uint32_t dummy = uint32_t(TILE) + lane;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    idx[k] = raw[k] < dummy ? raw[k] : dummy;
    nb[k] = wave::lds_u8(base, Off::lnorm + idx[k]);
  }
  // row of each posting's frequency: computed while the norm bytes are on their way
  uint32_t row[N];
#pragma unroll
  for (int k = 0; k < N; ++k) row[k] = (freq[k] << 10) + tab[k];   // v_lshl_add_u32
  wave::keep_all(row);
  wave::keep_all(nb);
#pragma unroll
  for (int k = 0; k < N; ++k) t[k] = wave::lds_f32(base, (nb[k] << 2) + row[k]);   // v_lshl_add_u32
  wave::keep_all_f(t);
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (sizeof(ACC) == 4) {
      float scaled = wave::fma(cs[k], t[k], 1.f);
      wave::keep_f(scaled);
      fx[k] = static_cast<ACC>(static_cast<uint32_t>(scaled));
    } else {
      float scaled = cs[k] * t[k];
      wave::keep_f(scaled);
      fx[k] = fixed_from_scaled<ACC>(scaled);
    }
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    wave::lds_add(base, Off::acc + idx[k] * uint32_t(sizeof(ACC)), fx[k]);
    if (AND) wave::lds_add(base, Off::cnt + (idx[k] & ~3u), 1u << (8u * (idx[k] & 3u)));
  }
}

// `m` receives the bit offset of the lane's first value in its stream (simd4: lane stream
// l = 2*(lane & 1), value row lane >> 1; scalar: value 2*lane): the extraction shifts by its
// low 5 bits, so it rides along with the payload words instead of being multiplied again.
template<int LAYOUT>
__device__ __forceinline__ void payload_load(uint64_t addr, uint32_t bits, uint32_t skip16,
                                             unsigned lane, uint64_t& a, uint64_t& b,
                                             uint32_t& m) {
  if (LAYOUT == kSimd4) {
    // value j = 4r + l: SSE lane l, bit r*bits of that lane's stream; word k of it = u32 4k + l
    m = wave::mul24(lane >> 1, bits);
    const uint32_t voff = (((m >> 5) + skip16) << 4) | ((lane & 1u) << 3);
    a = wave::gload_u64(addr, voff);
    b = wave::gload_u64(addr, voff + 16u);
  } else {
    m = wave::mul24(lane << 1, bits);
    const uint32_t voff = ((m >> 5) << 2) + (skip16 << 4);
    a = wave::gload_u64(addr, voff);
    b = wave::gload_u64(addr, voff + 4u);
  }
}

// Values 2*lane, 2*lane+1 out of the payload words, 1 <= bits <= 31, given the bit offset
// `m` payload_load computed: one funnel shift (v_alignbit_b32, which only looks at the low
// 5 bits of the shift) + one bit-field extract (v_bfe_u32) each.
template<int LAYOUT>
__device__ __forceinline__ void extract_at(uint64_t a, uint64_t b, uint32_t bits, uint32_t m,
                                           uint32_t& v0, uint32_t& v1) {
  if (LAYOUT == kSimd4) {
    v0 = wave::bfe(wave::funnel(uint32_t(b), uint32_t(a), m), bits);
    v1 = wave::bfe(wave::funnel(uint32_t(b >> 32), uint32_t(a >> 32), m), bits);
  } else {
    const uint32_t s = m & 31u;
    const uint32_t w0 = uint32_t(a), w1 = uint32_t(a >> 32), w2 = uint32_t(b >> 32);
    v0 = wave::bfe(wave::funnel(w1, w0, s), bits);
    const uint32_t s1 = s + bits;  // <= 62
    const bool hi = s1 >= 32u;
    v1 = wave::bfe(wave::funnel(hi ? w2 : w1, hi ? w1 : w0, s1), bits);
  }
}

struct PairRegs {   // payload words of one item pair (two per block part) and their bit
  uint64_t ada, adb, afa, afb;   // offsets, in flight or landed
  uint64_t bda, bdb, bfa, bfb;
  uint32_t adm, afm, bdm, bfm;
};
struct ItemPair {
  ItemCalc a, b;
};

template<int LAYOUT>
__device__ __forceinline__ void pair_issue(const ItemAddr& A, const ItemAddr& B, unsigned lane,
                                           PairRegs& p) {
  payload_load<LAYOUT>(A.addr, A.dbits, 0u, lane, p.ada, p.adb, p.adm);
  payload_load<LAYOUT>(A.addr, A.fbits, A.dbits, lane, p.afa, p.afb, p.afm);
  payload_load<LAYOUT>(B.addr, B.dbits, 0u, lane, p.bda, p.bdb, p.bdm);
  payload_load<LAYOUT>(B.addr, B.fbits, B.dbits, lane, p.bfa, p.bfb, p.bfm);
}
// scalar loads: the records of a wavefront's list are consecutive (`at` = address of one)
__device__ __forceinline__ void load_pair(uint64_t at, ItemPair& c) {
  c.a = wave::sload<ItemCalc>(at + 8u);
  c.b = wave::sload<ItemCalc>(at + sizeof(ItemG) + 8u);
}
__device__ __forceinline__ void load_addrs(uint64_t at, ItemAddr& a, ItemAddr& b) {
  a = wave::sload<ItemAddr>(at);
  b = wave::sload<ItemAddr>(at + sizeof(ItemG));
}

// This wavefront's work items of one doc tile.  The tile's n items are dealt round robin
// (item g goes to wavefront g % nw), each wavefront's share stored contiguously:
// wavefront w owns records [start, start + n).
struct WaveList {
  uint64_t p;    // address of the first record (an integer: see wave::sload)
  uint32_t n;
};
__device__ __forceinline__ WaveList wave_list(uint64_t items, uint32_t off0, uint32_t off1,
                                              uint32_t wv, uint32_t nw_log2) {
  const uint32_t n = off1 - off0, nw = 1u << nw_log2;
  const uint32_t a = n >> nw_log2, r = n & (nw - 1u);
  WaveList l;
  l.p = items + uint64_t(off0 + wv * a + (wv < r ? wv : r)) * sizeof(ItemG);
  l.n = a + (wv < r ? 1u : 0u);
  return l;
}

// The pipeline state a wavefront carries into a tile: the payload words of its first two
// item pairs (requested, maybe not landed) and the address parts of the third pair.
struct ItemPipe {
  PairRegs p0, p1;
  ItemAddr la, lb;
};
// Head of a list, step 1: scalar loads of the first two pairs' address parts.  (Reads up to 4
// records from the list's start whatever its length: always readable, see kItemSlack.)
__device__ __forceinline__ void pipe_begin(const WaveList& l, ItemAddr (&h)[4]) {
  if (l.n) {
    load_addrs(l.p, h[0], h[1]);
    load_addrs(l.p + 2 * sizeof(ItemG), h[2], h[3]);
  }
}
// step 2, once those may have landed: payload requests for both pairs, address parts of the third
template<int LAYOUT>
__device__ __forceinline__ void pipe_issue(const WaveList& l, ItemPipe& s, const ItemAddr (&h)[4],
                                           unsigned lane) {
  if (l.n) {
    pair_issue<LAYOUT>(h[0], h[1], lane, s.p0);
    if (l.n > 2) pair_issue<LAYOUT>(h[2], h[3], lane, s.p1);
    load_addrs(l.p + 4 * sizeof(ItemG), s.la, s.lb);
  }
}

// generic item: any block framing, any scorer, the decoded tail; does its own loads
// (`at` = address of the item's record: the payload address is re-read from it)
template<typename ACC, int LAYOUT, int TILE, bool AND>
__device__ __forceinline__ void slow_item(const DevSegment& seg, const TileSmemT<ACC>& sm,
                                          const ItemCalc& I, uint64_t at, uint32_t tile,
                                          unsigned lane) {
  const DevQTerm qt = sm.qts[I.aux & kItemTermMask];
  const float inv_one = 1.f / (qt.norm_const + qt.norm_length * 1.f);
  const float fx_mul = __uint_as_float(sm.slow[0]);
  const uint32_t lo = kDocMin + tile * uint32_t(TILE);
  const uint32_t left = sm.slow[1] + kDocMin - lo;   // docs from lo to the end of the segment
  const uint32_t span = left < uint32_t(TILE) ? left : uint32_t(TILE);
  if (I.aux & kItemTail) {
    // decoded vint tail / single doc (segment open): absolute doc ids
    const uint32_t* tdocs = seg.tail_docs;
    const uint32_t* tfreqs = seg.tail_freqs;
    for (uint32_t i = lane; i < I.base; i += 64)
      tile_apply<ACC, TILE, AND>(seg, sm, qt, inv_one, tdocs[I.tab + i] - lo, tfreqs[I.tab + i],
                                 lo, span, fx_mul);
    return;
  }
  uint32_t d0, d1, f0, f1;
  const uint8_t* blk = reinterpret_cast<const uint8_t*>(wave::sload<uint64_t>(at));
  decode_block<LAYOUT, true>(blk, I.dbits, I.fbits, I.base, lane, d0, d1, f0, f1);
  // (tile-relative docs: I.base is)
  tile_apply<ACC, TILE, AND>(seg, sm, qt, inv_one, d0, f0, lo, span, fx_mul);
  tile_apply<ACC, TILE, AND>(seg, sm, qt, inv_one, d1, f1, lo, span, fx_mul);
}

// hot path, one item: straight-line code
template<typename ACC, int LAYOUT, int TILE, bool AND>
__device__ __forceinline__ void fast_item(const TileSmemT<ACC>& sm, const ItemCalc& I, uint64_t da,
                                          uint64_t db, uint32_t dm, uint64_t fa, uint64_t fb,
                                          uint32_t fm, unsigned lane) {
  uint32_t x0, x1, f0, f1;
  extract_at<LAYOUT>(da, db, I.dbits, dm, x0, x1);
  extract_at<LAYOUT>(fa, fb, I.fbits, fm, f0, f1);
  const uint32_t d1 = I.base + wave::inclusive_scan(x0 + x1);
  const uint32_t fadd = I.aux >> kItemFreqShift;
  const float css[2] = {I.cs, I.cs};
  const uint32_t tabs2[2] = {I.tab, I.tab};
  const uint32_t raw2[2] = {d1 - x1, d1};
  if (I.aux & kItemTable) {
    const uint32_t freqs2[2] = {f0, f1};
    tile_post_table<ACC, TILE, AND, 2>(sm, css, tabs2, raw2, freqs2, lane);
  } else {
    const uint32_t freqs2[2] = {f0 + fadd, f1 + fadd};
    tile_post<ACC, TILE, AND, 2>(sm, css, tabs2, raw2, freqs2, lane, (I.aux & kItemSqrt) != 0u);
  }
}

// hot path, two items fused: 4 postings per lane in flight, two independent DPP scan
// chains, all LDS lookups issued back to back
template<typename ACC, int LAYOUT, int TILE, bool AND>
__device__ __forceinline__ void fast_pair(const TileSmemT<ACC>& sm, const ItemPair& c,
                                          const PairRegs& p, unsigned lane) {
  uint32_t ax0, ax1, af0, af1, bx0, bx1, bf0, bf1;
  extract_at<LAYOUT>(p.ada, p.adb, c.a.dbits, p.adm, ax0, ax1);
  extract_at<LAYOUT>(p.bda, p.bdb, c.b.dbits, p.bdm, bx0, bx1);
  extract_at<LAYOUT>(p.afa, p.afb, c.a.fbits, p.afm, af0, af1);
  extract_at<LAYOUT>(p.bfa, p.bfb, c.b.fbits, p.bfm, bf0, bf1);
  uint32_t sa = ax0 + ax1, sb = bx0 + bx1;
  wave::inclusive_scan2(sa, sb);
  const uint32_t ad1 = c.a.base + sa;
  const uint32_t bd1 = c.b.base + sb;
  const uint32_t fa_add = c.a.aux >> kItemFreqShift, fb_add = c.b.aux >> kItemFreqShift;
  const float css[4] = {c.a.cs, c.a.cs, c.b.cs, c.b.cs};
  const uint32_t tabs4[4] = {c.a.tab, c.a.tab, c.b.tab, c.b.tab};
  const uint32_t raw4[4] = {ad1 - ax1, ad1, bd1 - bx1, bd1};
  if (c.a.aux & kItemTable) {   // (a fused pair is two table items or two general ones)
    const uint32_t freqs4[4] = {af0, af1, bf0, bf1};
    tile_post_table<ACC, TILE, AND, 4>(sm, css, tabs4, raw4, freqs4, lane);
  } else {
    const uint32_t freqs4[4] = {af0 + fa_add, af1 + fa_add, bf0 + fb_add, bf1 + fb_add};
    tile_post<ACC, TILE, AND, 4>(sm, css, tabs4, raw4, freqs4, lane, (c.a.aux & kItemSqrt) != 0u);
  }
}

// One pair of the hot loop.  Generic items are only noted (`slow` collects their flags) and
// left to items_slow(): with their code inlined here the loop would not fit the scalar
// register file.
template<typename ACC, int LAYOUT, int TILE, bool AND>
__device__ __forceinline__ void pair_compute(const TileSmemT<ACC>& sm, const ItemPair& c,
                                             const PairRegs& p, uint32_t& slow, unsigned lane) {
  if (c.a.aux & kItemPair) {   // scalar branches: the records live in SGPRs
    fast_pair<ACC, LAYOUT, TILE, AND>(sm, c, p, lane);
    return;
  }
  slow |= c.a.aux;
  if (!(c.a.aux & kItemSlow))
    fast_item<ACC, LAYOUT, TILE, AND>(sm, c.a, p.ada, p.adb, p.adm, p.afa, p.afb, p.afm, lane);
  if (c.a.aux & kItemSolo) return;
  slow |= c.b.aux;
  if (!(c.b.aux & kItemSlow))
    fast_item<ACC, LAYOUT, TILE, AND>(sm, c.b, p.bda, p.bdb, p.bdm, p.bfa, p.bfb, p.bfm, lane);
}

// All items of this wavefront in one tile: decode + score + accumulate.  In flight at any
// time: the payload words of the next pair (requested one step ago) and the scalar load of the
// address parts of the pair after it.  One step = fetch the pair's records (a scalar-cache
// hit: their cache line came in with the address parts two steps ago), compute the pair out
// of registers, then request the payload of pair i+2 into the registers just freed and the
// address parts of pair i+3.  Those scalar loads go LAST in a step: scalar loads return out
// of order, so the first wait for any LDS or scalar result drains all of them — placed here
// they have the next step's whole decode to land.  Unrolled by two so that the two payload
// register sets simply alternate (no moves).
// In: `s` as left by pipe_begin + pipe_issue for THIS list.  The look-ahead reads run up to
// 7 records past the list's end: those are records of other lists or the slack records
// behind the last one (always readable, addresses always valid); they are never computed.
// Returns the OR of the aux words of the items it did not fuse: kItemSlow set = the list
// holds generic items, items_slow() has to run.
template<typename ACC, int LAYOUT, int TILE, bool AND>
__device__ __forceinline__ uint32_t items_run(const TileSmemT<ACC>& sm, const WaveList& l,
                                              ItemPipe& s, unsigned lane) {
  uint32_t slow = 0;
  // p = address of the pair computed next, left = items from there on (one running pointer:
  // every record address below is p + a constant, i.e. an immediate of the scalar load)
  uint64_t p = l.p;
  ItemPair c;
  for (uint32_t left = l.n; left;) {
    load_pair(p, c);
    pair_compute<ACC, LAYOUT, TILE, AND>(sm, c, s.p0, slow, lane);
    if (left > 4) {
      pair_issue<LAYOUT>(s.la, s.lb, lane, s.p0);
      load_addrs(p + 6 * sizeof(ItemG), s.la, s.lb);
    }
    if (left <= 2) break;
    load_pair(p + 2 * sizeof(ItemG), c);
    pair_compute<ACC, LAYOUT, TILE, AND>(sm, c, s.p1, slow, lane);
    if (left > 6) {
      pair_issue<LAYOUT>(s.la, s.lb, lane, s.p1);
      load_addrs(p + 8 * sizeof(ItemG), s.la, s.lb);
    }
    if (left <= 4) break;
    left -= 4;
    p += 4 * sizeof(ItemG);
  }
  return slow;
}

// Second pass over a list for its generic items (rare: all-equal doc blocks, 32-bit wide
// parts, scorers outside the table family, decoded tails).
template<typename ACC, int LAYOUT, int TILE, bool AND>
__device__ __forceinline__ void items_slow(const DevSegment& seg, const TileSmemT<ACC>& sm,
                                           const WaveList& l, uint32_t tile, unsigned lane) {
  for (uint32_t i = 0; i < l.n; ++i) {
    const uint64_t at = l.p + i * sizeof(ItemG);
    const ItemCalc I = wave::sload<ItemCalc>(at + 8u);
    if (I.aux & kItemSlow) slow_item<ACC, LAYOUT, TILE, AND>(seg, sm, I, at, tile, lane);
  }
}

// The score function a term scorer compiles to, on explicit values — used on (tf, the doc's
// norm) for postings and on (max freq, min norm) for block bounds, exactly as the wanderator
// runs the one ScoreFunction on its WandSource (formats_10.cpp:2498-2503).  The reference's
// float expressions: bm25.cpp:281-282, 313, 348-359; tfidf.cpp:185-187, 251-253.  `norm`: the
// Norm2 value; for the legacy `Norm` kinds the bits of the stored float.
__device__ __forceinline__ float score_value(const DevQTerm& qt, uint32_t freq, uint32_t norm) {
  const float tf = static_cast<float>(freq);
  switch (qt.kind) {
    case kBM1:
      return qt.c0;
    case kBM15:
      return qt.c0 - qt.c0 / (1.f + tf / qt.norm_const);
    case kBM25Tiny: {
      const float inv = norm ? 1.f / (qt.norm_const + qt.norm_length * static_cast<float>(norm)) : 0.f;
      return qt.c0 - qt.c0 / (1.f + tf * inv);
    }
    case kBM25One: {
      const float inv = 1.f / (qt.norm_const + qt.norm_length);
      return qt.c0 - qt.c0 / (1.f + tf * inv);
    }
    case kBM25Wide: {
      const float c1 = qt.norm_const + qt.norm_length * static_cast<float>(norm);
      return qt.c0 - qt.c0 * c1 / (c1 + tf);
    }
    case kTfidf:
      return sqrtf(tf) * qt.c0;
    case kBM25Legacy: {
      const float c1 = qt.norm_const + qt.norm_length * (1.f / __uint_as_float(norm));
      return qt.c0 - qt.c0 * c1 / (c1 + sqrtf(tf));
    }
    case kTfidfLegacy:
      return sqrtf(tf) * qt.c0 * __uint_as_float(norm);
    default: {  // kTfidfTiny, kTfidfWide
      const float r = norm ? 1.f / sqrtf(static_cast<float>(norm)) : 0.f;
      return sqrtf(tf) * qt.c0 * r;
    }
  }
}
__host__ __device__ __forceinline__ bool needs_norm(int32_t kind) {
  return kind == kBM25Tiny || kind == kBM25Wide || kind == kTfidfTiny || kind == kTfidfWide ||
         kind == kBM25Legacy || kind == kTfidfLegacy;
}
// The norm value score_value() wants for `doc` (1 without a column: bm25.cpp:487-489).
__device__ __forceinline__ uint32_t norm_value(const DevSegment& seg, uint32_t doc) {
  if (!seg.norms) return 1u;
  if (seg.norm_legacy) return __float_as_uint(norm_legacy(seg, doc));
  return seg.norm_width == 1 ? seg.norms[doc - seg.norm_min_doc] : norm_global(seg, doc);
}
// Bound of a block from its (max freq, min norm); the legacy kinds have no block-max norms:
// their bound over any norm (stored values are <= 1).
__device__ __forceinline__ float block_bound(const DevQTerm& qt, uint32_t maxf, uint32_t minn) {
  if (qt.kind == kBM25Legacy) return qt.c0;
  if (qt.kind == kTfidfLegacy) return sqrtf(static_cast<float>(maxf)) * qt.c0;
  return score_value(qt, maxf, minn);
}
// what no posting of the term can exceed (BM25 family: the supremum over tf; TF-IDF: at the
// term's largest frequency, DevTerm::tf_bound)
__device__ __forceinline__ float term_bound(const DevQTerm& qt, uint32_t tf_bound) {
  return sqrt_kind(qt.kind) || qt.kind == kTfidfWide || qt.kind == kTfidfLegacy
             ? sqrtf(static_cast<float>(tf_bound)) * qt.c0 : qt.c0;
}

// -------------------------------------------------------- item list build --

// Per (unit, doc tile): how many work items reach the tile — the blocks of every term that
// overlap it (from the plan table) plus one item per term whose decoded tail reaches into it.
// grid = n_units * tb workgroups, tb = ceil(max tiles / kThreads).
__global__ void __launch_bounds__(kThreads)
k_items_count(const DevQuery* queries, uint32_t jt, uint32_t tile_docs, uint32_t tb,
              const uint32_t* first, const DevTail* tails, uint32_t* tile_cnt) {
  const uint32_t unit = blockIdx.x / tb;
  const uint32_t tile = (blockIdx.x % tb) * kThreads + threadIdx.x;
  const DevQuery qd = queries[unit];
  if (tile >= qd.n_tiles || qd.first_off == kNoPlan) return;
  const uint32_t* f0 = first + qd.first_off + uint64_t(tile) * jt;
  const DevTail* tl = tails + uint64_t(unit) * jt;
  const uint32_t lo = kDocMin + tile * tile_docs;
  uint32_t n = 0;
  for (uint32_t j = 0; j < qd.n_terms; ++j) {
    const uint32_t b0 = f0[j];
    uint32_t b1 = f0[jt + j] + 1u;
    b1 = b1 < tl[j].nblk ? b1 : tl[j].nblk;
    n += b1 > b0 ? b1 - b0 : 0u;
    if (tl[j].n && tl[j].first_doc < lo + tile_docs && tl[j].last_doc >= lo) ++n;
  }
  tile_cnt[qd.tile_base + tile] = n;
}

// One wavefront per (unit, doc tile) writes the tile's work items: everything the scoring
// loop would otherwise derive per block — where the payload lives, the bit widths out of the
// block directory, the preceding block's last doc relative to the tile, the term's scaled c0
// and table row — goes into the record once.  grid = n_units * tb, tb = ceil(max tiles / kWaves).
__global__ void __launch_bounds__(kThreads)
k_items_fill(const DevSegment* segs, const DevQuery* queries, const DevQTerm* qterms,
             uint32_t jt, uint32_t tile_docs, uint32_t tb, uint32_t nw_log2,
             uint32_t caches_off, const uint32_t* first, const DevTail* tails,
             const uint32_t* tile_off, uint32_t total_tiles, ItemG* items,
             float* tile_ub /*WAND: upper bound of any doc's score in the tile; else null*/) {
  __shared__ uint32_t s_pre[kWaves][kMaxTerms + 1];  // exclusive prefix sums of the block counts
  __shared__ uint32_t s_b0[kWaves][kMaxTerms];
  __shared__ uint32_t s_ub[kWaves][kMaxTerms];       // WAND: per term, largest block-max score (float bits)
  // per-term values every item of the term needs (read per lane with a lane-varying term
  // slot: from LDS, not as gathers from the query / term records in global memory)
  __shared__ uint64_t s_dir[kWaves][kMaxTerms];      // DevTail::dir_off
  __shared__ float s_cs[kWaves][kMaxTerms];          // c0 * fixed-point scale
  __shared__ uint32_t s_tf[kWaves][kMaxTerms];       // table slot | table_kind << 8 | sqrt_kind << 9
                                                     // | (every tf of the term has a table row) << 10
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t wv = threadIdx.x >> 6;
  const uint32_t unit = blockIdx.x / tb;
  const uint32_t tile = (blockIdx.x % tb) * kWaves + wv;
  const DevQuery qd = queries[unit];
  if (tile >= qd.n_tiles || qd.first_off == kNoPlan) return;   // whole wavefront
  const DevSegment& seg = segs[qd.seg];
  const uint32_t* f0 = first + qd.first_off + uint64_t(tile) * jt;
  const DevTail* tl = tails + uint64_t(unit) * jt;
  const DevQTerm* qts = qterms + qd.first_term;
  const uint32_t lo = kDocMin + tile * tile_docs;
  uint32_t nb = 0, b0 = 0;
  bool tail_here = false;
  if (lane < qd.n_terms) {
    b0 = f0[lane];
    uint32_t b1 = f0[jt + lane] + 1u;
    b1 = b1 < tl[lane].nblk ? b1 : tl[lane].nblk;
    nb = b1 > b0 ? b1 - b0 : 0u;
    tail_here = tl[lane].n && tl[lane].first_doc < lo + tile_docs && tl[lane].last_doc >= lo;
    const DevQTerm qt = qts[lane];
    s_dir[wv][lane] = tl[lane].dir_off;
    s_cs[wv][lane] = qt.c0 * qd.fx_mul;
    s_tf[wv][lane] = (qt.cache_id < kMaxCaches ? qt.cache_id : 0u) |
                     ((table_kind(qt.kind) && qt.cache_id < kMaxCaches) ? 0x100u : 0u) |
                     (sqrt_kind(qt.kind) ? 0x200u : 0u) |
                     (qt.pad1 < table_rows(qd.n_caches) ? 0x400u : 0u);
  }
  const uint32_t incl = wave::inclusive_scan(nb);
  if (lane <= kMaxTerms) s_pre[wv][lane] = incl - nb;   // lanes >= n_terms hold the total
  if (lane < kMaxTerms) s_b0[wv][lane] = b0;
  if (lane < kMaxTerms) s_ub[wv][lane] = 0u;
  const uint64_t tail_mask = wave::ballot(tail_here);
  wave::sync();
  const uint32_t n_blocks = s_pre[wv][kMaxTerms];
  const uint32_t n = n_blocks + uint32_t(__builtin_popcountll(tail_mask));
  const uint32_t ut = qd.tile_base + tile;
  const uint32_t off0 = tile_off[ut];
  const uint32_t nw = 1u << nw_log2;
  const uint32_t a = n >> nw_log2, r = n & (nw - 1u);
  const uint64_t pk = reinterpret_cast<uint64_t>(seg.pk);

  // term slot and directory row of block item g
  auto locate = [&](uint32_t g, uint32_t& j, uint64_t& e) {
    j = 0;
    for (uint32_t t = 1; t < qd.n_terms; ++t) j += s_pre[wv][t] <= g ? 1u : 0u;
    e = s_dir[wv][j] + s_b0[wv][j] + (g - s_pre[wv][j]);
  };
  // the value of an ALL-EQUAL freq block: vint behind the doc part and the 0 header byte
  auto freq_const = [&](uint32_t j, const BlkDir& d) {
    uint32_t len;
    return vint_from(wave::load_u64(seg.doc + tl[j].doc_start + d.off + 2u + 16u * (d.bits & 0xFFu)),
                     &len);
  };
  // straight-line path: a scorer of the table family, the block in the packed image, and an
  // all-equal frequency that fits the record's 16 bits.  cls: 0 = generic, 1 = straight-line
  // with general frequencies, 2 = every frequency of the TERM has a table row (per term, not
  // per block: the generic path and join.h make the same choice, see tile_apply); bit 2: the
  // square-root form (only tells general items apart)
  const uint32_t rows = table_rows(qd.n_caches);
  auto classify = [&](uint32_t j, const BlkDir& d, uint32_t& fconst) {
    fconst = 0;
    const uint32_t dbits = d.bits & 0xFFu, fbits = d.bits >> 8;
    const uint32_t tf = s_tf[wv][j];
    if (!((tf & 0x100u) && pk_units(dbits, fbits) != 0u)) return 0u;
    if (query_merge(qd.op)) return 0u;   // Max/Min merged scores: the generic path's atomic max
    const uint32_t cls = (tf & 0x400u) ? 2u : ((tf & 0x200u) ? 5u : 1u);
    if (fbits == 0u) {
      fconst = freq_const(j, d);
      if (fconst > 0xFFFFu) return 0u;
    }
    return cls;
  };
  for (uint32_t g0 = 0; g0 < n; g0 += 64) {   // (whole wavefront: shuffles inside)
    const uint32_t g = g0 + lane;
    ItemG I{};
    uint32_t cls = 0;
    if (g < n_blocks) {
      uint32_t j, fconst;
      uint64_t e;
      locate(g, j, e);
      const BlkDir d = seg.blk_dir[e];
      if (tile_ub)   // positive floats order like their bit patterns
        atomicMax(&s_ub[wv][j], __float_as_uint(block_bound(qts[j], seg.blk_maxf[e], seg.blk_minn[e])));
      cls = classify(j, d, fconst);
      const bool fast = cls != 0u;
      const uint32_t tf = s_tf[wv][j];
      const uint32_t slot = tf & 0xFFu;
      I.addr = fast ? pk + (uint64_t(d.aoff) << 4)
                    : reinterpret_cast<uint64_t>(seg.doc) + tl[j].doc_start + d.off;
      I.dbits = d.bits & 0xFFu;
      I.fbits = d.bits >> 8;
      I.base = d.prev_last - lo;
      I.cs = s_cs[wv][j];
      // table items: an all-equal frequency selects its row right here
      I.tab = caches_off + (slot * rows + (cls == 2u ? fconst : 0u)) * 1024u;
      I.aux = j | (fast ? 0u : kItemSlow) | ((tf & 0x200u) ? kItemSqrt : 0u) |
              (cls == 2u ? kItemTable : fconst << kItemFreqShift);
    } else if (g < n) {
      // the (g - n_blocks)-th term whose tail reaches into the tile
      uint64_t m = tail_mask;
      for (uint32_t s = g - n_blocks; s; --s) m &= m - 1;
      const uint32_t j = uint32_t(__builtin_ctzll(m));
      I.addr = pk;   // readable; never used
      I.dbits = 1;
      I.fbits = 1;
      I.base = tl[j].n;
      I.cs = 0.f;
      I.tab = tl[j].tail_row;
      I.aux = j | kItemSlow | kItemTail;
    }
    // class of the wavefront's next item, g + nw: computed by lane + nw, or — for the last
    // nw lanes — looked up directly
    uint32_t next_cls = __shfl_down(cls, nw, 64);
    if (lane + nw >= 64u) {
      next_cls = 0;
      if (g + nw < n_blocks) {
        uint32_t j2, fconst2;
        uint64_t e2;
        locate(g + nw, j2, e2);
        next_cls = classify(j2, seg.blk_dir[e2], fconst2);
      }
    }
    if (g < n) {
      const uint32_t w = g & (nw - 1u), i = g >> nw_log2;
      const uint32_t n_w = a + (w < r ? 1u : 0u);
      if (!(i & 1u)) {
        if (i + 1u == n_w) I.aux |= kItemSolo;
        else if (cls && g + nw < n_blocks && next_cls == cls) I.aux |= kItemPair;
      }
      items[off0 + w * a + (w < r ? w : r) + i] = I;
    }
  }
  if (tile_ub) {
    // WAND: bound of the tile = sum over the terms of their largest block-max score in it
    // (the min lambda of block_disjunction sums the sub-iterators' bounds,
    // disjunction.hpp:1133-1167); a decoded tail has no block-max entry: the term's global
    // bound, counted on top of its blocks (which only loosens the bound)
    wave::sync();
    float ub = 0.f;
    if (lane < qd.n_terms) {
      ub = __uint_as_float(s_ub[wv][lane]);
      if (tail_here) ub += term_bound(qts[lane], seg.terms[tl[lane].term].tf_bound);
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) ub += __shfl_xor(ub, sft, 64);
    if (lane == 0) tile_ub[ut] = ub * (1.f + 1e-6f);
  }
  if (ut + 1u == total_tiles && lane < kItemSlack) {   // readable slack behind the last list
    ItemG I;
    I.addr = pk;
    I.dbits = 1;
    I.fbits = 1;
    I.base = 0;
    I.cs = 0.f;
    I.tab = caches_off;
    I.aux = kItemSlow | kItemTail | kItemSolo;
    items[off0 + n + lane] = I;
  }
}

// ----------------------------------------------------------------- shared --

__device__ __forceinline__ uint32_t score_bin(float v, float scale) {
  const float x = fminf(v * scale, float(kBins - 1));
  return uint32_t(x);
}

// A conservative fixed-point image of the lower edge of score bin `bs` (the
// exact float bin test follows for the few accumulators that pass it).
template<typename ACC>
__device__ __forceinline__ ACC bin_threshold(uint32_t bs, const DevQuery& qd) {
  if (!bs) return ACC(1);
  const double edge = double(bs) / double(qd.bin_scale);
  return static_cast<ACC>(edge / double(qd.fx_inv) * (1.0 - 1e-6));
}

// kNormPieces 8-byte pieces of a tile's norm bytes per thread (the host sizes workgroups to
// at least TILE / (8 * kNormPieces) threads)
template<int TILE>
struct NormStage {
  static constexpr int kPieces = (TILE + 4095) / 4096;
  uint64_t w[kPieces];
  // norms1 = address of the 1-byte column's value for doc kDocMin (0: no such column),
  // count = values from there on
  __device__ __forceinline__ void load(uint64_t norms1, uint64_t count, uint32_t tile) {
#pragma unroll
    for (int e = 0; e < kPieces; ++e) w[e] = 0;
    if (norms1) {
      const uint64_t first = uint64_t(tile) * TILE;
      const uint32_t i0 = threadIdx.x * (8u * kPieces);
      // (the column is staged with kPadBytes of slack: whole pieces stay in bounds)
      if (i0 < uint32_t(TILE) && first + i0 < count) {
        const uint64_t base = norms1 + first;
#pragma unroll
        for (int e = 0; e < kPieces; ++e) w[e] = wave::gload_u64(base, i0 + 8u * e);
      }
    }
  }
  __device__ __forceinline__ void store(uint8_t* lnorm) const {
    const uint32_t i0 = threadIdx.x * (8u * kPieces);
    if (i0 < uint32_t(TILE)) {
      uint64_t* d = reinterpret_cast<uint64_t*>(lnorm + i0);
#pragma unroll
      for (int e = 0; e < kPieces; ++e) d[e] = w[e];
    }
  }
};

// ----------------------------------------------------------------- pilot --

// One workgroup per unit scores the tiles {phase, phase+P, ...}, histograms their
// scores into kBins linear bins over [0, U] and picks a bin b*; k_score drops
// everything below b*.
//   sound (margin == 0): the largest bin with at least k sampled docs at or above
//     it.  Those docs exist, so the final k-th score is >= the lower edge of b*.
//     The full set then holds about k*P candidates.
//   estimated (margin > 0): the largest bin with at least margin*k*(sampled
//     tiles)/(all tiles) sampled docs at or above it (never more than k, never less
//     than kPilotMinSample), i.e. an expected margin*k candidates.  Not a proof:
//     k_select checks "fewer than k candidates although more docs matched"
//     (kStatusUnderflow) and the host then re-runs the batch in sound mode.
constexpr uint32_t kPilotMinSample = 48;

template<typename ACC, int LAYOUT, int TILE, bool AND>
__global__ void __launch_bounds__(kTileThreadsMax)
k_pilot(const uint32_t* units, const DevSegment* segs, const DevQuery* queries,
        const DevQTerm* qterms, uint32_t stride, uint32_t nw_log2, const uint32_t* tile_off,
        uint64_t items /*address of the ItemG records*/, uint32_t* bstar, uint32_t margin,
        const uint32_t* min_bin /*[unit] bin of the caller's score::Min; null: none*/) {
  RT_DYN_SMEM(smem);
  if (!wave::lds_is_at_zero(smem)) __builtin_trap();  // the tile arrays are addressed absolutely
  unsigned char* rest;
  const TileSmemT<ACC> sm = carve<ACC, TILE, AND>(smem, &rest);
  uint32_t* hist = reinterpret_cast<uint32_t*>(rest);  // [kBins]
  const uint32_t tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const uint32_t wv = wave::uniform(tid >> 6);
  const uint32_t q = units[blockIdx.x];   // the units executed as doc tiles
  const DevQuery qd = queries[q];
  const DevSegment& seg = segs[qd.seg];   // (read field by field: the generic path is cold)
  const bool tiny = seg.norms && seg.norm_width == 1;
  const uint64_t norms1 = tiny ? reinterpret_cast<uint64_t>(seg.norms) + (kDocMin - seg.norm_min_doc) : 0;
  const uint64_t norm_count = tiny ? seg.norm_count - (kDocMin - seg.norm_min_doc) : 0;
  const uint32_t n_tiles = qd.n_tiles;
  const uint32_t* dead = seg.dead;
  for (uint32_t i = tid; i < kBins; i += blockDim.x) hist[i] = 0u;
  for (uint32_t i = tid; i < uint32_t(TILE) + 64u; i += blockDim.x) sm.acc[i] = ACC(0);
  if (AND) {
    for (uint32_t i = tid; i < (uint32_t(TILE) + 64u) / 4u; i += blockDim.x) sm.cnt[i] = 0u;
  }
  if (tid < qd.n_terms) sm.qts[tid] = qterms[qd.first_term + tid];
  if (tid == 0) {
    sm.slow[0] = __float_as_uint(qd.fx_mul);
    sm.slow[1] = seg.num_docs;
    sm.slow[2] = table_rows(qd.n_caches);
    sm.slow[3] = query_merge(qd.op);
  }
  __syncthreads();
  build_tables(sm, qd.n_caches, qd.n_terms);
  for (uint32_t tile = (q * 7u) % stride; tile < n_tiles; tile += stride) {
    NormStage<TILE> nrm;
    nrm.load(norms1, norm_count, tile);
    const uint32_t ut = qd.tile_base + tile;
    const WaveList l = wave_list(items, wave::uniform(tile_off[ut]),
                                 wave::uniform(tile_off[ut + 1]), wv, nw_log2);
    ItemPipe s;
    ItemAddr head[4];
    pipe_begin(l, head);
    pipe_issue<LAYOUT>(l, s, head, lane);
    nrm.store(sm.lnorm);
    __syncthreads();   // norms (and, first time round, tables and cleared accumulators) are in place
    if (items_run<ACC, LAYOUT, TILE, AND>(sm, l, s, lane) & kItemSlow)
      items_slow<ACC, LAYOUT, TILE, AND>(seg, sm, l, tile, lane);
    __syncthreads();
    for (uint32_t i = tid; i < uint32_t(TILE); i += blockDim.x) {
      const ACC a = sm.acc[i];
      sm.acc[i] = ACC(0);
      bool m = a != ACC(0);
      const uint32_t c = AND ? (sm.cnt[i >> 2] >> (8u * (i & 3u))) & 0xFFu : 0u;
      if (AND && (qd.op & 0xFF) == 1)  // AND / min-match: op = 1 | required matches << 8
        m = c >= query_need(qd.op);
      // (a deleted doc never leaves the iterator: SegmentReaderImpl::mask)
      if (dead && m) m = !doc_dead(dead, kDocMin + tile * uint32_t(TILE) + i);
      if (m) {
        ACC f = merged_fixed<ACC>(a, query_merge(qd.op));
        if (AND && query_min_both(qd.op) && c < 2u) f = ACC(0);
        atomicAdd(&hist[score_bin(from_fixed<ACC>(f, qd.fx_inv), qd.bin_scale)], 1u);
      }
    }
    __syncthreads();
    if (AND) {
      for (uint32_t i = tid; i < uint32_t(TILE) / 4u; i += blockDim.x) sm.cnt[i] = 0u;
    }
  }
  __syncthreads();
  // docs the sample must show at or above b*
  uint32_t need = qd.k;
  if (margin) {
    const uint32_t phase = (q * 7u) % stride;
    const uint32_t sampled = phase < n_tiles ? (n_tiles - phase + stride - 1) / stride : 0u;
    const uint64_t est = (uint64_t(margin) * qd.k * sampled + n_tiles - 1) / n_tiles;
    const uint32_t lo = est < kPilotMinSample ? kPilotMinSample : uint32_t(est < 0xFFFFFFFFull ? est : 0xFFFFFFFFull);
    need = lo < qd.k ? lo : qd.k;
  }
  // suffix search: lane L of wave 0 owns the 8 bins of chunk 63-L
  if (tid < 64) {
    const uint32_t chunk = 63u - lane;
    uint32_t s = 0;
    for (uint32_t i = 0; i < kBins / 64; ++i) s += hist[chunk * (kBins / 64) + i];
    const uint32_t incl = wave::inclusive_scan(s);  // docs in chunks >= chunk
    const uint64_t reach = wave::ballot(incl >= need);
    uint32_t result = 0;
    if (reach) {
      const int src = __builtin_ctzll(reach);  // highest chunk reaching `need`
      const uint32_t above = wave::bcast(incl - s, src);
      const uint32_t c = 63u - uint32_t(src);
      uint32_t cum = above;
      for (int i = int(kBins / 64) - 1; i >= 0; --i) {
        cum += hist[c * (kBins / 64) + uint32_t(i)];
        if (cum >= need) { result = c * (kBins / 64) + uint32_t(i); break; }
      }
    }
    // (+ the caller's own lower bound — irs::score::Min — whichever is higher)
    if (lane == 0) bstar[q] = (min_bin && min_bin[q] > result) ? min_bin[q] : result;
  }
}

// ----------------------------------------------------------------- score --
//
// Persistent workgroups.  The grid is sized to fill the chip once; every workgroup
// pulls CHUNKS of kChunkTiles consecutive doc tiles of one unit from a global counter.
// Per tile and wavefront:
//   items_run      decode + score + accumulate this wavefront's items of tile u
//   (scalar loads for the head of its list of tile u+1 go out)
//   barrier B1     every accumulation of tile u has landed
//   norms of tile u+1 (in registers since tile u-1) -> LDS; requests for tile u+2;
//   payload requests for the first two item pairs of tile u+1
//   epilogue       read + clear the accumulators of tile u, count hits, stage candidates
//                  (all of the above is in flight behind it)
//   barrier B2     accumulators are clear again
// The returning atomic that reserves candidate slots for tile u-1 and the dequeue of the
// next chunk are in flight the same way.

constexpr uint32_t kChunkTiles = 16;
constexpr uint32_t kScoreCands = 128;   // per-tile candidate staging slots (x2 buffers)

enum : uint32_t {  // indices into the workgroup's LDS scratch words
  kVChunk = 0,     // current chunk id
  kVBase = 2,      // global candidate base of the previous tile
  kVBaseLast = 3,  // ... of the chunk's last tile (own word: slow threads may still read kVBase)
  kVNc0 = 4,       // kVNc0 + (u % 3): candidate count of tile u
  kVWords = 8,
};

template<typename ACC, int TILE, bool AND>
constexpr uint32_t tile_smem_bytes() {
  return TileOff<ACC, TILE, AND>::end;
}
template<typename ACC, int TILE, bool AND>
constexpr uint32_t score_smem_bytes() {
  return tile_smem_bytes<ACC, TILE, AND>()
         + 4u * (kChunkTiles + 2)                          // item offsets of the chunk's tiles
         + 4u * kChunkTiles                                // WAND: which of them are skipped
         + 8u * 2u * kScoreCands                           // candidate staging x2
         + 4u * kVWords;
}

// k_score's parameters live in device memory and are read where they are used (scalar
// loads): held in SGPRs for the whole kernel they would crowd the work-item records out of
// the scalar register file (the compiler then parks records in VGPRs and pays v_readfirstlane).
struct ScoreArgs {
  const DevSegment* segs;
  const DevQuery* queries;
  const DevQTerm* qterms;
  const uint32_t* tile_off;
  uint64_t items;               // address of the ItemG records
  const uint32_t* bstar;
  uint64_t* cands;
  uint32_t* cand_count;
  unsigned long long* hits;
  uint32_t* work_counter;
  const float* tile_ub;         // WAND: per-tile score bounds (k_items_fill), else null
  uint32_t* pruned;             // [unit] set when a tile was skipped (k_select's underflow check)
  uint32_t cpq;                 // chunk ids per unit
  uint32_t n_units;
  uint32_t nw_log2;
  uint32_t cand_cap;
};
#define IRS_ARG(field) \
  (wave::sload<decltype(ScoreArgs::field)>(wave::opaque64(args) + offsetof(ScoreArgs, field)))

template<typename ACC, int LAYOUT, int TILE, bool AND>
__global__ void __launch_bounds__(kTileThreadsMax)
k_score(uint64_t args /*address of a ScoreArgs*/) {
  RT_DYN_SMEM(smem);
  if (!wave::lds_is_at_zero(smem)) __builtin_trap();  // the tile arrays are addressed absolutely
  unsigned char* rest;
  const TileSmemT<ACC> sm = carve<ACC, TILE, AND>(smem, &rest);
  uint32_t* toff = reinterpret_cast<uint32_t*>(rest);       // [kChunkTiles + 2]
  rest += 4u * (kChunkTiles + 2);
  uint32_t* tdead = reinterpret_cast<uint32_t*>(rest);      // [kChunkTiles]
  rest += 4u * kChunkTiles;
  uint64_t* lcand = reinterpret_cast<uint64_t*>(rest);      // [2][kScoreCands]
  rest += 8u * 2u * kScoreCands;
  uint32_t* vars = reinterpret_cast<uint32_t*>(rest);

  const uint32_t tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const uint32_t wv = wave::uniform(tid >> 6);
  // every (segment, query) unit owns `cpq` chunk ids (sized for the segment with the most
  // tiles; ids past a shorter segment's last tile are empty chunks)
  const uint32_t n_units = IRS_ARG(n_units);
  const uint32_t total_chunks = n_units * IRS_ARG(cpq);

  for (uint32_t i = tid; i < uint32_t(TILE) + 64u; i += blockDim.x) sm.acc[i] = ACC(0);
  if (AND) {
    for (uint32_t i = tid; i < (uint32_t(TILE) + 64u) / 4u; i += blockDim.x) sm.cnt[i] = 0u;
  }
  if (tid < kVWords) vars[tid] = 0u;
  if (tid == 0) vars[kVChunk] = atomicAdd(IRS_ARG(work_counter), 1u);
  __syncthreads();
  uint32_t chunk = wave::uniform(vars[kVChunk]);
  __syncthreads();  // (an empty chunk has no barrier before thread 0 publishes the next id)

  while (chunk < total_chunks) {
    // dequeue of the NEXT chunk: issued now, consumed after this chunk
    uint32_t next_chunk = 0;
    if (tid == 0) next_chunk = atomicAdd(IRS_ARG(work_counter), 1u);

    // chunk-major ids: every unit's first chunk, then every unit's second, ... so the short
    // last chunks of the units are handed out at the very end (smaller tail)
    const DevQuery* queries = IRS_ARG(queries);
    const uint32_t q = wave::uniform(queries[chunk % n_units].run_unit);
    const uint32_t tile0 = (chunk / n_units) * kChunkTiles;
    const DevQuery qd = queries[q];
    // (the segment record is read field by field — only a few on the hot path, the generic
    // path is cold: a private copy of all of it would crowd the work items out of the SGPRs)
    const DevSegment& seg = IRS_ARG(segs)[qd.seg];
    const bool tiny = seg.norms && seg.norm_width == 1;
    const uint64_t norms1 = tiny ? reinterpret_cast<uint64_t>(seg.norms) + (kDocMin - seg.norm_min_doc) : 0;
    const uint64_t norm_count = tiny ? seg.norm_count - (kDocMin - seg.norm_min_doc) : 0;
    const uint32_t n_tiles = qd.n_tiles;
    const uint32_t* dead = seg.dead;   // the segment's deleted docs (null: none)
    const uint32_t ntile = tile0 >= n_tiles ? 0u
                           : ((n_tiles - tile0) < kChunkTiles ? (n_tiles - tile0) : kChunkTiles);
    const uint32_t bs = IRS_ARG(bstar)[q];
    uint32_t my_hits = 0;     // matching docs this lane saw in the chunk's epilogues
    uint32_t pend_base = 0;   // thread 0: reserved candidate base of the previous tile (in flight)
    if (ntile) {   // (an empty chunk id of a shorter segment only runs the hand-over below)
    // ---- chunk prologue: everything that is per query / per chunk ----------
    if (tid < qd.n_terms) sm.qts[tid] = IRS_ARG(qterms)[qd.first_term + tid];
    if (tid <= ntile) toff[tid] = IRS_ARG(tile_off)[qd.tile_base + tile0 + tid];
    if (tid < ntile) {
      // WAND: no doc of the tile can reach the threshold bin -> the tile is skipped (its work
      // items are not even read)
      const float* ub = IRS_ARG(tile_ub);
      const bool dead = ub && bs && score_bin(ub[qd.tile_base + tile0 + tid], qd.bin_scale) < bs;
      tdead[tid] = dead ? 1u : 0u;
      if (dead) IRS_ARG(pruned)[q] = 1u;
    }
    if (tid == 0) {
      sm.slow[0] = __float_as_uint(qd.fx_mul);
      sm.slow[1] = seg.num_docs;
      sm.slow[2] = table_rows(qd.n_caches);
      sm.slow[3] = query_merge(qd.op);
    }
    NormStage<TILE> nrm;
    nrm.load(norms1, norm_count, tile0);
    __syncthreads();
    build_tables(sm, qd.n_caches, qd.n_terms);
    const ACC thr = bin_threshold<ACC>(bs, qd);
    const uint32_t nw_log2 = IRS_ARG(nw_log2);
    // ---- prime the pipeline: this wavefront's items of tile 0, norms of tiles 0 and 1
    WaveList l = wave_list(IRS_ARG(items), wave::uniform(toff[0]), wave::uniform(toff[1]), wv,
                           nw_log2);
    if (wave::uniform(tdead[0])) l.n = 0;
    ItemPipe s;
    ItemAddr head[4];
    pipe_begin(l, head);
    nrm.store(sm.lnorm);
    pipe_issue<LAYOUT>(l, s, head, lane);
    if (1 < ntile) nrm.load(norms1, norm_count, tile0 + 1);
    __syncthreads();

    for (uint32_t u = 0; u < ntile; ++u) {
      const uint32_t tile = tile0 + u;
      const bool has_next = u + 1 < ntile;
      // compute of tile u: decode + score + accumulate
      if (items_run<ACC, LAYOUT, TILE, AND>(sm, l, s, lane) & kItemSlow)
        items_slow<ACC, LAYOUT, TILE, AND>(seg, sm, l, tile, lane);
      if (has_next) {   // records of the head of this wavefront's list of tile u+1
        l = wave_list(IRS_ARG(items), wave::uniform(toff[u + 1]), wave::uniform(toff[u + 2]), wv,
                      nw_log2);
        if (wave::uniform(tdead[u + 1])) l.n = 0;
        pipe_begin(l, head);
      }
      __syncthreads();  // B1: every accumulation of tile u has landed

      if (has_next) {
        nrm.store(sm.lnorm);  // norms of tile u+1 (tile u no longer reads them)
        if (u + 2 < ntile) nrm.load(norms1, norm_count, tile + 2u);
        pipe_issue<LAYOUT>(l, s, head, lane);   // payload of the first two pairs of tile u+1
      }

      // epilogue of tile u: read + clear the accumulators, count hits, stage candidates
      uint64_t* lc = lcand + (u & 1u) * kScoreCands;
      uint32_t* ncand = vars + kVNc0 + (u % 3u);
      auto candidate = [&](uint32_t i, ACC a) {   // rare
        // (a sum of at most kMaxTerms units is what postings of zero-boost terms leave: score 0)
        const ACC f = merged_fixed<ACC>(a, query_merge(qd.op));
        const float v = f <= ACC(kMaxTerms) ? 0.f : from_fixed<ACC>(f, qd.fx_inv);
        if (score_bin(v, qd.bin_scale) >= bs) {
          const uint64_t key = make_key(v, kDocMin + tile * uint32_t(TILE) + i);
          const uint32_t slot = atomicAdd(ncand, 1u);
          if (slot < kScoreCands) {
            lc[slot] = key;
          } else {  // rarer: more candidates in one tile than staging slots
            const uint32_t cap = IRS_ARG(cand_cap);
            const uint32_t g = atomicAdd(&IRS_ARG(cand_count)[q], 1u);
            if (g < cap) IRS_ARG(cands)[uint64_t(q) * cap + g] = key;
          }
        }
      };
      if (!wave::uniform(tdead[u])) {   // (a skipped tile accumulated nothing)
        const bool is_and = AND && (qd.op & 0xFF) == 1;  // op = 1 | required matches << 8
        const uint32_t need = query_need(qd.op);
        const bool min_both = AND && query_min_both(qd.op);
        // eight accumulators per lane per step: two 4-wide LDS reads in flight, two wide clears.
        // `two`: the step's second group of four exists (always, when the tile is a whole
        // number of double steps — the usual geometry: the test then costs nothing).
        const uint32_t step = blockDim.x * 4u;
        auto eight = [&](uint32_t i, uint32_t i2, bool two) {
          ACC a[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = sm.acc[i + e];
#pragma unroll
          for (int e = 0; e < 4; ++e) a[4 + e] = sm.acc[i2 + e];
#pragma unroll
          for (int e = 0; e < 8; ++e) wave::keep_acc(a[e]);
          uint32_t cw0 = 0, cw1 = 0;   // match counters of docs i..i+3 / i2..i2+3, a byte each
          if (AND) {
            cw0 = sm.cnt[i >> 2];
            cw1 = sm.cnt[i2 >> 2];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) sm.acc[i + e] = ACC(0);
#pragma unroll
          for (int e = 0; e < 4; ++e) sm.acc[i2 + e] = ACC(0);
          if (AND) {   // this thread is the only reader of those counter words: clear them here
            sm.cnt[i >> 2] = 0u;
            sm.cnt[i2 >> 2] = 0u;
          }
          if (!two) {
#pragma unroll
            for (int e = 4; e < 8; ++e) a[e] = ACC(0);
          }
          if (dead) {   // (wave-uniform) deleted docs look untouched: SegmentReaderImpl::mask.
            // Bit (doc - kDocMin) = tile * TILE + i: four docs of a group are one aligned nibble
            const uint32_t j = tile * uint32_t(TILE) + i, j2 = tile * uint32_t(TILE) + i2;
            const uint32_t g0 = (dead[j >> 5] >> (j & 31u)) & 0xFu, g1 = (dead[j2 >> 5] >> (j2 & 31u)) & 0xFu;
#pragma unroll
            for (int e = 0; e < 8; ++e)
              a[e] = (((e < 4 ? g0 : g1) >> (uint32_t(e) & 3u)) & 1u) ? ACC(0) : a[e];
          }
          if (AND && is_and) {
            // AND / min-match: a doc counts only with >= `need` matching terms; the others
            // are made to look untouched
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const uint32_t c = ((e < 4 ? cw0 : cw1) >> (8u * (uint32_t(e) & 3u))) & 0xFFu;
              a[e] = c >= need ? a[e] : ACC(0);
            }
          }
          if (AND && min_both) {
            // a kMin disjunction of two: a doc only one of them holds scores 0 (as the
            // complement of the smallest fixed-point value: still "matched")
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const uint32_t c = ((e < 4 ? cw0 : cw1) >> (8u * (uint32_t(e) & 3u))) & 0xFFu;
              a[e] = (a[e] != ACC(0) && c < 2u) ? ACC(~ACC(1)) : a[e];
            }
          }
          ACC top = a[0];
#pragma unroll
          for (int e = 0; e < 8; ++e) top = a[e] > top ? a[e] : top;
          // matching docs, per lane (summed once per chunk)
          wave::count_nonzero4(my_hits, a[0], a[1], a[2], a[3]);
          wave::count_nonzero4(my_hits, a[4], a[5], a[6], a[7]);
          if (top >= thr) {  // rare: one copy of the candidate code, per-lane loop
            uint32_t cm = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) cm |= a[e] >= thr ? (1u << e) : 0u;
            while (cm) {
              const uint32_t e = uint32_t(__builtin_ctz(cm));
              cm &= cm - 1u;
              ACC x = a[0];
#pragma unroll
              for (int f = 1; f < 8; ++f) x = e == uint32_t(f) ? a[f] : x;
              candidate((e < 4u ? i : i2 - 4u) + e, x);
            }
          }
        };
        if (uint32_t(TILE) % (2u * step) == 0u) {   // (wave-uniform)
          for (uint32_t i = tid * 4u; i < uint32_t(TILE); i += 2u * step) eight(i, i + step, true);
        } else {
          for (uint32_t i = tid * 4u; i < uint32_t(TILE); i += 2u * step) {
            const bool two = i + step < uint32_t(TILE);  // same for the whole workgroup
            eight(i, two ? i + step : i, two);
          }
        }
      }
      if (tid == 0) {
        vars[kVBase] = pend_base;                 // tile u-1's reservation has arrived by now
        vars[kVNc0 + ((u + 1u) % 3u)] = 0u;       // counter of tile u+1 (last used by tile u-2)
      }
      __syncthreads();  // B2: accumulators are clear again
      // flush tile u-1's staged candidates to its reserved global range
      if (u > 0) {
        const uint32_t pn_raw = vars[kVNc0 + ((u - 1u) % 3u)];
        const uint32_t pn = pn_raw < kScoreCands ? pn_raw : kScoreCands;
        if (pn) {
          const uint32_t gbase = vars[kVBase];
          const uint64_t* pl = lcand + ((u - 1u) & 1u) * kScoreCands;
          const uint32_t cap = IRS_ARG(cand_cap);
          uint64_t* out = IRS_ARG(cands) + uint64_t(q) * cap;
          for (uint32_t i = tid; i < pn; i += blockDim.x) {
            const uint32_t g = gbase + i;
            if (g < cap) out[g] = pl[i];
          }
        }
      }
      // reserve global slots for tile u (returning atomic; consumed one tile later)
      if (tid == 0) {
        const uint32_t cn_raw = *ncand;
        const uint32_t cn = cn_raw < kScoreCands ? cn_raw : kScoreCands;
        pend_base = cn ? atomicAdd(&IRS_ARG(cand_count)[q], cn) : 0u;
      }
    }
    }
    // ---- chunk epilogue: flush the last tile, publish hits, pick up the next chunk
    if (tid == 0) {
      vars[kVBaseLast] = pend_base;
      vars[kVChunk] = next_chunk;
    }
    my_hits = wave::reduce_add(my_hits);
    if (lane == 0 && my_hits)
      atomicAdd(&IRS_ARG(hits)[q], static_cast<unsigned long long>(my_hits));
    __syncthreads();
    {
      const uint32_t lu = ntile ? ntile - 1u : 0u;
      const uint32_t pn_raw = vars[kVNc0 + (lu % 3u)];
      const uint32_t pn = pn_raw < kScoreCands ? pn_raw : kScoreCands;
      if (pn) {
        const uint32_t gbase = vars[kVBaseLast];
        const uint64_t* pl = lcand + (lu & 1u) * kScoreCands;
        const uint32_t cap = IRS_ARG(cand_cap);
        uint64_t* out = IRS_ARG(cands) + uint64_t(q) * cap;
        for (uint32_t i = tid; i < pn; i += blockDim.x) {
          const uint32_t g = gbase + i;
          if (g < cap) out[g] = pl[i];
        }
      }
    }
    chunk = wave::uniform(vars[kVChunk]);
    __syncthreads();  // everyone has read the chunk id and the staging buffers
    if (tid == 0) vars[kVNc0] = vars[kVNc0 + 1] = vars[kVNc0 + 2] = 0u;
    __syncthreads();
  }
}
#undef IRS_ARG

}  // namespace irs_hip
