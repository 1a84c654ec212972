// fast.h — plain disjunctions over joined posting streams in TWO passes:
//
//   k_join_fast     every (query, doc tile): accumulate an APPROXIMATE score per doc in packed
//                   16-bit accumulators (two docs per LDS word), count the matching docs exactly,
//                   emit the docs whose approximate score can still reach the threshold bin
//   k_join_rescore  per query: keep the candidates that can still belong to the top k, look their
//                   postings up in the exact entry streams and re-score them EXACTLY — the same
//                   fixed-point arithmetic as join.h / score.h, so the hits are bit-identical
//
// Replaces, like join.h, block_disjunction::refill (disjunction.hpp:1240-1351) + the harness heap
// (index-search.cpp:740-787) for Or filters; what the reference does per posting — a score
// function call and a `+=` into score_buf — is split into a cheap bound and an exact evaluation
// of the few docs that matter, which is what its own WAND mode does with block-max scores
// (disjunction.hpp:1133-1167), only per doc.
//
// Why: PMC of round 3's k_join_score: more than half of its instructions are per
// (query, tile) costs — the accumulator scan of the epilogue and the work split — and both LDS
// accesses of a posting (table gather + accumulator add) conflict on the banks.  A tile of LDS
// holds 12288 exact accumulators but 24576 packed ones: half the tiles, half the epilogue bytes
// per doc, and a posting costs ONE LDS operation because its query-independent factor
// T(tf, norm) was evaluated once, by k_join, for all queries of the batch:
//
//   fast entry  [ T / Tn * 2^16 : 16 | (doc - first doc of its 12288-doc half tile) * 4 : 16 ]
//   per posting c = mul_hi(entry, csq) + 1        csq = c0 * Tn * S16 of the (query, term)
//               acc16[doc] += c                   one v_mul_hi, one v_lshl_add, one v_and, one ds_add
//
// Sums stay below 2^15 (S16 is the query's scale), so the two halves of a word never carry into
// each other and SWAR / v_pk_*_u16 tests work on both at once.  Error: |c - x| < 3 units per posting
// against the real-valued contribution x (u16 floor, the entry's low bits riding along, csq floor,
// mul_hi floor, + 1), i.e. |A - X| < 3 n for a doc of an n-term query.
#pragma once
#include "join.h"

namespace irs_hip {

constexpr uint32_t kFastTile = 2u * kJoinTile;   // docs per packed accumulator tile
constexpr uint32_t kFastChunkTiles = 16;         // tiles per work-queue item (= a join.h chunk's doc range)
constexpr uint32_t kFastErr = 3;                 // |c - x| per posting, in 16-bit units
constexpr uint32_t kFastKeep = 6;                // x n: slack under the k-th approximate score (see k_join_rescore)
constexpr uint32_t kFastMaxSum = 32767;          // a doc's accumulator stays below 2^15
constexpr uint32_t kFastBins = 1024;             // coarse histogram of approximate scores (32 units per bin)
constexpr uint32_t kFastRound = 4096;            // candidates re-scored per round of k_join_rescore
constexpr uint32_t kFastPre = 512;               // entries per piece requested behind the previous tile's epilogue

// LDS layout of k_join_fast (byte offsets; the accumulators sit at LDS address 0: an entry's low
// 16 bits ARE the byte address of its doc's word)
struct FastOff {
  static constexpr uint32_t acc = 0;                                      // [kJoinTile] u32 = 2 x u16
  static constexpr uint32_t dummy = 4u * kJoinTile;                       // [64] u32: a word per lane nobody reads
  static constexpr uint32_t rng = dummy + 256u;                           // [2 * chunk tiles + 1][kMaxTerms] u32
  static constexpr uint32_t jts = rng + 4u * (2u * kFastChunkTiles + 1u) * kMaxTerms;   // JoinTerm[kMaxTerms]
  static constexpr uint32_t share = jts + uint32_t(sizeof(JoinTerm)) * kMaxTerms;       // [16 waves][4] u32
  static constexpr uint32_t cand = share + 16u * 16u;                     // [2][kJoinCands] u64
  static constexpr uint32_t vars = cand + 8u * 2u * kJoinCands;           // [16] u32
  static constexpr uint32_t end = vars + 64u;
};
static_assert(FastOff::cand % 8u == 0u, "candidate keys are 8-byte aligned");
static_assert(4u * kJoinTile + 256u <= 65536u, "an entry's 16 address bits reach every word and the dummies");

struct FastArgs {
  const DevQuery* queries;
  const JoinTerm* jterms;      // entries / bounds of the EXACT streams; pad[0] = csq
  int64_t fast_delta;          // fast entries of a stream = its exact entries + this many bytes
  const uint32_t* bstar;
  uint64_t* cands;             // [unit][cap] (approximate score << 32) | doc
  uint32_t* cand_count;
  unsigned long long* hits;
  const uint32_t* order;       // as JoinArgs: one work queue per XCD
  uint32_t* work_counter;
  uint32_t base[kJoinQueues + 1];
  uint32_t first[kJoinQueues + 1];
  uint32_t cpq, n_units, nw_log2, cand_cap, chunk_tiles;
};

// what a wavefront takes of a chunk's entries: from fraction f_lo of term j0 to fraction f_hi of
// term j1 (32-bit fixed point; 0xFFFFFFFF = the whole term), every term in between whole
struct FastShare {
  uint32_t j0, j1, f_lo, f_hi;
};
__device__ __forceinline__ uint32_t fast_cut(uint32_t n, uint32_t f) {
  return f == 0xFFFFFFFFu ? n : wave::mul_hi(n, f);
}

// `count` consecutive fast entries from `base` (wave-uniform), all of one half tile: the
// contribution goes to the low (shift 0) or the high (16) half of the doc's word
__device__ __forceinline__ void fast_post4(const unsigned char* lds, const uint32_t (&e)[4], uint32_t csq,
                                           uint32_t shift, uint32_t one) {
  uint32_t c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) c[k] = (wave::mul_hi(e[k], csq) << shift) + one;
#pragma unroll
  for (int k = 0; k < 4; ++k) wave::lds_add(lds, FastOff::acc + (e[k] & 0xFFFCu), c[k]);
}
// 256 consecutive entries at `base`, lane l holding entries 4l .. 4l+3 (one 16-byte load per
// lane), of which the first n (1 .. 256) are wanted: lanes behind the end do not load ...
__device__ __forceinline__ void fast_load(uint64_t base, uint32_t n, unsigned lane, uint32_t (&e)[4]) {
  // (no initialisation of the other lanes' registers: fast_take replaces what they hold, and a
  // write here would have to wait for whatever load is still in flight into these registers)
  wave::undef4(e);
  if (4u * lane < n) wave::gload_u32x4(base, 16u * lane, e);
}
// ... and elements at or behind n add to the lane's dummy word
__device__ __forceinline__ void fast_take(const unsigned char* lds, uint32_t (&e)[4], uint32_t n,
                                          uint32_t csq, uint32_t shift, unsigned lane) {
  if (n < 256u) {   // (wave-uniform)
    const uint32_t dummy = FastOff::dummy + 4u * lane;   // (score 0: adds `one` to a word nobody reads)
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = 4u * lane + uint32_t(k) < n ? e[k] : dummy;
  }
  fast_post4(lds, e, csq, shift, 1u << shift);
}
// `count` consecutive entries from `base`: 512 at a time, both loads of a step in flight together
__device__ __forceinline__ void fast_run(const unsigned char* lds, uint64_t base, uint32_t count,
                                         uint32_t csq, uint32_t shift, unsigned lane) {
  while (count) {
    const uint32_t n = count < 512u ? count : 512u;
    uint32_t r0[4], r1[4];
    fast_load(base, n < 256u ? n : 256u, lane, r0);
    if (n > 256u) fast_load(base + 1024u, n - 256u, lane, r1);
    fast_take(lds, r0, n < 256u ? n : 256u, csq, shift, lane);
    if (n > 256u) fast_take(lds, r1, n - 256u, csq, shift, lane);
    base += 2048u;
    count -= n;
  }
}

// What the tile loop needs of a query term: where its fast entries are and its multiplier
struct alignas(16) FastTerm {
  uint64_t fent;
  uint32_t csq, pad;
};
enum : uint32_t {   // more LDS scratch words (behind join.h's kJ*)
  kFUnit = 8,       // the chunk's unit
  kFThr = 9,        // its threshold in 16-bit units
  kFDoc0 = 10,      // first doc of the chunk's first tile
  kFParity = 11,    // which candidate staging buffer the chunk fills
};

// The tiles of one chunk.  Everything it needs lives in LDS (tile boundaries, term records, the
// wavefronts' shares, the chunk's scalars) — a function of its own so that the persistent
// kernel's bookkeeping does not compete with the tile loop for registers.
// Returns the lane's match count: low half | high half.
__device__ __attribute__((noinline)) uint32_t fast_tiles(unsigned char* smem, const FastArgs* args,
                                                         uint32_t ntile_arg) {
  const uint32_t* rng = reinterpret_cast<const uint32_t*>(smem + FastOff::rng);
  const FastTerm* fts = reinterpret_cast<const FastTerm*>(smem + FastOff::jts);
  const uint32_t* vars = reinterpret_cast<const uint32_t*>(smem + FastOff::vars);
  const uint32_t tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const uint32_t wv = wave::uniform(tid >> 6);
  const uint32_t ntile = wave::uniform(ntile_arg);   // (arguments arrive in vector registers)
  // (read ONCE: inside this function every use of blockDim is a load from the dispatch packet,
  // and a vector-memory wait in the epilogue loop would also wait for the prefetched entries)
  const uint32_t n_threads = wave::uniform(blockDim.x);
  const FastShare sh = reinterpret_cast<const FastShare*>(smem + FastOff::share)[wv];
  const uint32_t j0 = wave::uniform(sh.j0), j1 = wave::uniform(sh.j1);
  const uint32_t f_lo = wave::uniform(sh.f_lo), f_hi = wave::uniform(sh.f_hi);
  const uint32_t thr = wave::uniform(vars[kFThr]);
  const uint32_t thr_k = (0x8000u - (thr < 0x8000u ? thr : 0x8000u)) * 0x00010001u;   // half + K: bit 15 set <=> half >= thr
  uint32_t hit_pk = 0;
  // A wavefront's entries of a tile are up to 2 (j1 - j0 + 1) PIECES: per term its low half
  // tile's part (piece 2 (j - j0)) and its high half tile's (+ 1); piece_of() works one out
  // (count 0: empty).
  struct Piece {
    uint64_t base;
    uint32_t cnt, par;   // par: csq | (high half) << 16
  };
  const uint32_t n_pieces = j1 >= j0 ? 2u * (j1 - j0 + 1u) : 0u;
  auto piece_of = [&](uint32_t u, uint32_t p) -> Piece {   // (u < ntile, p < n_pieces)
    const uint32_t j = j0 + (p >> 1);
    const uint32_t a0 = rng[(2u * u) * kMaxTerms + j];
    const uint32_t b1 = rng[(2u * u + 1u) * kMaxTerms + j];
    const uint32_t a2 = rng[(2u * u + 2u) * kMaxTerms + j];
    const uint32_t n = a2 - a0;
    const uint32_t lo = a0 + (j == j0 ? fast_cut(n, f_lo) : 0u);
    const uint32_t hi = a0 + (j == j1 ? fast_cut(n, f_hi) : n);
    const uint32_t mid = b1 < lo ? lo : (b1 > hi ? hi : b1);
    const uint32_t from = (p & 1u) ? mid : lo, to = (p & 1u) ? hi : mid;
    Piece P;
    P.base = fts[j].fent + 4ull * from;
    P.cnt = to > from ? to - from : 0u;
    P.par = fts[j].csq | ((p & 1u) << 16);
    return P;
  };
  // The first four pieces of every tile of the chunk, worked out ONCE, lane-parallel: lane
  // 4 u + s holds piece s of tile u — the tile loop reads them with v_readlane (no LDS round
  // trip, no arithmetic between a barrier and the next tile's loads).
  static_assert(kFastChunkTiles * 4u == 64u, "one lane per (tile, piece slot)");
  uint32_t d_lo = 0, d_hi = 0, d_cnt = 0, d_par = 0;
  {
    const uint32_t u = lane >> 2, sl = lane & 3u;
    if (u < ntile && sl < n_pieces) {
      const Piece P = piece_of(u, sl);
      d_lo = uint32_t(P.base);
      d_hi = uint32_t(P.base >> 32);
      d_cnt = P.cnt;
      d_par = P.par;
    }
  }
  auto slot = [&](uint32_t u, uint32_t sl) -> Piece {   // (wave-uniform arguments)
    const uint32_t L = 4u * u + sl;
    Piece P;
    P.cnt = u < ntile ? wave::read_lane(d_cnt, L & 63u) : 0u;
    P.base = (uint64_t(wave::read_lane(d_hi, L & 63u)) << 32) | wave::read_lane(d_lo, L & 63u);
    P.par = wave::read_lane(d_par, L & 63u);
    return P;
  };
  // Pieces 0 and 1 of the NEXT tile — up to kFastPre entries each — are requested before this
  // tile's barrier: the loads fly behind the epilogue, and a wavefront whose share lies inside
  // one term (the rule) finds its whole tile waiting.
  Piece A{0, 0, 0}, B{0, 0, 0};
  uint32_t ra[4], rb[4], sa[4], sb[4];   // entries 0 .. 255 / 256 .. 511 of the two pieces
  auto request = [&](uint32_t u) {
    A = slot(u, 0u);
    B = slot(u, 1u);
    const uint32_t na = A.cnt < kFastPre ? A.cnt : kFastPre, nb = B.cnt < kFastPre ? B.cnt : kFastPre;
    // all addresses first, then the loads back to back: whatever the address arithmetic has to
    // wait for (the registers' previous loads) is waited for BEFORE the first new load is issued
    uint64_t pa = A.base + 16u * lane, pb = B.base + 16u * lane;
    wave::keep64(pa);
    wave::keep64(pb);
    wave::undef4(ra);
    wave::undef4(rb);
    wave::undef4(sa);
    wave::undef4(sb);
    if (4u * lane < na) wave::gload_u32x4_at(pa, ra);
    if (4u * lane < nb) wave::gload_u32x4_at(pb, rb);
    if (4u * lane + 256u < na) wave::gload_u32x4_at(pa + 1024u, sa);
    if (4u * lane + 256u < nb) wave::gload_u32x4_at(pb + 1024u, sb);
  };
  auto consume = [&](const Piece& P, uint32_t (&r)[4], uint32_t (&s2)[4]) {
    const uint32_t n = P.cnt < kFastPre ? P.cnt : kFastPre;
    const uint32_t csq = P.par & 0xFFFFu, shift = (P.par >> 16) << 4;
    fast_take(smem, r, n < 256u ? n : 256u, csq, shift, lane);
    if (n > 256u) fast_take(smem, s2, n - 256u, csq, shift, lane);
    if (P.cnt > n) fast_run(smem, P.base + 4ull * n, P.cnt - n, csq, shift, lane);
  };
  // A share that SPANS terms (one wavefront of 16 as a rule: the query's rare terms together) is
  // many small pieces.  Its tile is laid out in SLABS of up to 64 entries of one piece, lane
  // k < 16 holding slab k's address, count and multiplier; all 16 loads are issued before the
  // previous tile's barrier, like the two big pieces of a one-term share.  (Loaded piece by
  // piece behind the barrier, every one of them costs a memory round trip while the other 15
  // wavefronts of the workgroup wait.)
  const bool multi = n_pieces > 2u;
  uint32_t g_lo = 0, g_hi = 0, g_cnt = 0, g_par = 0;
  bool g_over = false;   // the tile has more than 16 slabs
  // piece p = lane (p < n_pieces <= 32): the piece, its slabs, its first slab's index
  auto layout = [&](uint32_t u, Piece& P, uint32_t& first, uint32_t& ns) {
    P = Piece{0, 0, 0};
    if (u < ntile && lane < n_pieces) P = piece_of(u, lane);
    ns = (P.cnt + 63u) >> 6;
    first = wave::inclusive_scan(ns) - ns;
  };
  auto request_multi = [&](uint32_t u) {
    Piece P;
    uint32_t first, ns;
    layout(u, P, first, ns);
    g_cnt = 0;
    for (uint32_t i = 0; i < n_pieces; ++i) {   // (wave-uniform) slab k = lane belongs to piece i?
      const uint32_t n = wave::read_lane(ns, i);
      if (!n) continue;
      const uint32_t f = wave::read_lane(first, i);
      if (f >= 16u) break;
      const uint32_t cnt = wave::read_lane(P.cnt, i), par = wave::read_lane(P.par, i);
      const uint64_t base = (uint64_t(wave::read_lane(uint32_t(P.base >> 32), i)) << 32) |
                            wave::read_lane(uint32_t(P.base), i);
      if (lane >= f && lane < f + n && lane < 16u) {
        const uint32_t o = lane - f;
        const uint64_t at = base + 256ull * o;
        g_lo = uint32_t(at);
        g_hi = uint32_t(at >> 32);
        g_cnt = cnt - 64u * o < 64u ? cnt - 64u * o : 64u;
        g_par = par;
      }
    }
    g_over = wave::read_lane(first + ns, 31u) > 16u;
    // the 16 loads back to back (saddr form: no address arithmetic in vector registers)
    uint32_t* const r[16] = {&ra[0], &ra[1], &ra[2], &ra[3], &rb[0], &rb[1], &rb[2], &rb[3],
                             &sa[0], &sa[1], &sa[2], &sa[3], &sb[0], &sb[1], &sb[2], &sb[3]};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const uint32_t c = wave::read_lane(g_cnt, uint32_t(k));
      if (c) {
        const uint64_t base = (uint64_t(wave::read_lane(g_hi, uint32_t(k))) << 32) | wave::read_lane(g_lo, uint32_t(k));
        if (lane < c) *r[k] = wave::gload_u32(base, 4u * lane);
      }
    }
  };
  auto consume_multi = [&](uint32_t u) {
    uint32_t* const r[16] = {&ra[0], &ra[1], &ra[2], &ra[3], &rb[0], &rb[1], &rb[2], &rb[3],
                             &sa[0], &sa[1], &sa[2], &sa[3], &sb[0], &sb[1], &sb[2], &sb[3]};
    const uint32_t dummy = FastOff::dummy + 4u * lane;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const uint32_t c = wave::read_lane(g_cnt, uint32_t(k));
      if (c) {
        const uint32_t par = wave::read_lane(g_par, uint32_t(k));
        const uint32_t shift = (par >> 16) << 4;
        const uint32_t e = lane < c ? *r[k] : dummy;
        wave::lds_add(smem, FastOff::acc + (e & 0xFFFCu), (wave::mul_hi(e, par & 0xFFFFu) << shift) + (1u << shift));
      }
    }
    if (g_over) {   // (rare: the slabs behind the 16th, piece by piece)
      Piece P;
      uint32_t first, ns;
      layout(u, P, first, ns);
      for (uint32_t i = 0; i < n_pieces; ++i) {
        const uint32_t n = wave::read_lane(ns, i), f = wave::read_lane(first, i);
        if (!n || f + n <= 16u) continue;
        const uint32_t skip = f < 16u ? 64u * (16u - f) : 0u;   // entries of the piece in slabs < 16
        const uint32_t cnt = wave::read_lane(P.cnt, i), par = wave::read_lane(P.par, i);
        const uint64_t base = (uint64_t(wave::read_lane(uint32_t(P.base >> 32), i)) << 32) |
                              wave::read_lane(uint32_t(P.base), i);
        fast_run(smem, base + 4ull * skip, cnt - skip, par & 0xFFFFu, (par >> 16) << 4, lane);
      }
    }
  };
  if (multi) request_multi(0); else request(0);
  for (uint32_t u = 0; u < ntile; ++u) {
    // ---- accumulate this wavefront's entries of tile u
    if (multi) {
      consume_multi(u);
      request_multi(u + 1u);
    } else {
      if (A.cnt) consume(A, ra, sa);
      if (B.cnt) consume(B, rb, sb);
      request(u + 1u);
    }
    __syncthreads();   // B1: every accumulation of tile u has landed
    // ---- epilogue: 12 words (24 docs) per lane, read AND cleared by LDS exchanges
    auto candidate = [&](uint32_t word, uint32_t w) {   // rare: everything it needs comes from LDS
      uint32_t* v = reinterpret_cast<uint32_t*>(smem + FastOff::vars);
      const uint32_t parity = v[kFParity], q = v[kFUnit];
      const uint32_t doc0 = v[kFDoc0] + u * kFastTile;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t a = h ? (w >> 16) : (w & 0xFFFFu);
        if (a >= thr) {
          const uint64_t key = (uint64_t(a) << 32) | (doc0 + word + (h ? kJoinTile : 0u));
          const uint32_t slot = atomicAdd(&v[kJNc + parity], 1u);
          if (slot < kJoinCands) {
            reinterpret_cast<uint64_t*>(smem + FastOff::cand)[parity * kJoinCands + slot] = key;
          } else {   // rarer: more candidates in one chunk than staging slots
            const uint32_t cap = args->cand_cap;
            const uint32_t g = atomicAdd(&args->cand_count[q], 1u);
            if (g < cap) args->cands[uint64_t(q) * cap + g] = key;
          }
        }
      }
    };
    auto four = [&](uint32_t i, const uint32_t (&v)[4]) {
      uint32_t top = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hit_pk = wave::pk_add_u16(hit_pk, wave::pk_min_u16(v[k], 0x00010001u));
        top = wave::pk_max_u16(top, v[k]);
      }
      if ((wave::pk_add_u16(top, thr_k) & 0x80008000u) != 0u) {
        for (uint32_t k = 0; k < 4u; ++k) candidate(i + k, v[k]);
      }
    };
    // (one exchange at a time: the prefetched entries of the next tile occupy the registers a
    // second one in flight would need)
    for (uint32_t i = tid * 4u; i < kJoinTile; i += n_threads * 4u) {
      uint32_t v0[4];
      wave::lds_take4(smem, FastOff::acc + i * 4u, v0);
      four(i, v0);
    }
    __syncthreads();   // B2: accumulators are clear again
  }
  return hit_pk;
}

// Persistent workgroups pulling chunks of kFastChunkTiles consecutive tiles of one unit from one
// work queue per XCD (join.h join_pull).  Per chunk the entries of the query's terms are dealt to
// the wavefronts ONCE, as fractions of every term's entries in the chunk (FastShare): per tile a
// wavefront then only scales its fractions to the tile's counts — scalar arithmetic on one or two
// terms — instead of splitting every tile's entries anew.  Per tile: accumulate, barrier, every
// thread reads + clears its 12 words (24 docs), barrier.
__global__ void __launch_bounds__(kTileThreadsMax) IRS_WAVES_PER_SIMD(8)
k_join_fast(const FastArgs* __restrict__ args) {
  RT_DYN_SMEM(smem);
  if (!wave::lds_is_at_zero(smem)) __builtin_trap();
  uint32_t* acc = reinterpret_cast<uint32_t*>(smem + FastOff::acc);
  uint32_t* rng = reinterpret_cast<uint32_t*>(smem + FastOff::rng);
  FastTerm* fts = reinterpret_cast<FastTerm*>(smem + FastOff::jts);
  FastShare* share = reinterpret_cast<FastShare*>(smem + FastOff::share);
  uint64_t* lcand = reinterpret_cast<uint64_t*>(smem + FastOff::cand);
  uint32_t* vars = reinterpret_cast<uint32_t*>(smem + FastOff::vars);
  const uint32_t tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const uint32_t nw = blockDim.x >> 6;
  const uint32_t total_chunks = args->base[kJoinQueues];

  for (uint32_t i = tid; i < kJoinTile; i += blockDim.x) acc[i] = 0u;
  if (tid < 16u) vars[tid] = 0u;
  __syncthreads();
  auto pull = [&](uint32_t g0, uint32_t first_raw, uint32_t& g_out) -> uint32_t {
    g_out = g0;
    if (first_raw < args->base[g0 + 1u]) return first_raw;
    for (uint32_t k = 1; k < kJoinQueues; ++k) {
      const uint32_t g = (g0 + k) % kJoinQueues;
      if (args->base[g + 1u] == args->base[g]) continue;
      const uint32_t raw = atomicAdd(&args->work_counter[g], 1u);
      if (raw < args->base[g + 1u]) {
        g_out = g;
        return raw;
      }
    }
    return args->base[kJoinQueues];
  };
  if (tid == 0) {
    const uint32_t g0 = blockIdx.x % kJoinQueues;
    uint32_t g = g0;
    vars[kJChunk] = pull(g0, atomicAdd(&args->work_counter[g0], 1u), g);
    vars[kJGroup] = g;
  }
  __syncthreads();
  uint32_t chunk = wave::uniform(vars[kJChunk]);
  uint32_t group = wave::uniform(vars[kJGroup]);
  uint32_t parity = 0;
  uint32_t pend_q = 0, pend_n = 0, pend_base = 0;   // thread 0: the previous chunk's reservation
  __syncthreads();

  while (chunk < total_chunks) {
    uint32_t next_raw = 0;   // (a returning atomic in flight until the hand-over)
    if (tid == 0) next_raw = atomicAdd(&args->work_counter[group], 1u);
    const uint32_t n_units = args->first[group + 1u] - args->first[group];
    const uint32_t local = chunk - args->base[group];
    const uint32_t q = wave::uniform(args->order[args->first[group] + local % n_units]);
    const uint32_t per_chunk = args->chunk_tiles;
    const uint32_t tile0 = (local / n_units) * per_chunk;   // (fast tiles)
    const DevQuery qd = args->queries[q];
    const uint32_t n_half = qd.n_tiles;                      // the unit's kJoinTile tiles
    const uint32_t n_tiles = (n_half + 1u) >> 1;
    const uint32_t ntile = tile0 >= n_tiles ? 0u
                           : ((n_tiles - tile0) < per_chunk ? (n_tiles - tile0) : per_chunk);
    const uint32_t n_terms = qd.n_terms;
    uint32_t hit_pk = 0;   // matches seen by this lane: low half | high half
    if (ntile) {
      // the terms' fast entries and multipliers; half-tile boundaries of the chunk, every term:
      // rng[i][j] = bounds_j[2 * tile0 + i] (a segment with an odd number of half tiles: the
      // last boundary once more)
      if (tid < kMaxTerms) {
        FastTerm ft{0, 0, 0};
        if (tid < n_terms) {
          const JoinTerm jt = args->jterms[qd.first_term + tid];
          ft.fent = jt.entries + uint64_t(args->fast_delta);
          ft.csq = jt.pad[0];
        }
        fts[tid] = ft;
      }
      for (uint32_t e = tid; e < (2u * ntile + 1u) * kMaxTerms; e += blockDim.x) {
        const uint32_t i = e / kMaxTerms, j = e % kMaxTerms;
        uint32_t v = 0;
        if (j < n_terms) {
          uint32_t h = 2u * tile0 + i;
          h = h < n_half ? h : n_half;
          v = reinterpret_cast<const uint32_t*>(args->jterms[qd.first_term + j].bounds)[h];
        }
        rng[e] = v;
      }
      if (tid == 0) {
        // the threshold in 16-bit units: docs whose exact score reaches bin `bs` have A >= thr
        const uint32_t bs = args->bstar[q];
        uint32_t thr = 1u;
        if (bs) {
          const double edge = double(bs) / double(qd.bin_scale) * double(qd.s16) * (1.0 - 1e-6);
          const double t = edge - double(kFastErr * n_terms);
          thr = t > 1.0 ? uint32_t(t) : 1u;
        }
        vars[kFThr] = thr;
        vars[kFUnit] = q;
        vars[kFDoc0] = kDocMin + tile0 * kFastTile;
        vars[kFParity] = parity;
      }
      __syncthreads();
      // the wavefronts' shares of the chunk (one thread each)
      if (tid < nw) {
        // (32-bit sums: a chunk holds fewer than 2^32 entries of a query's terms; the fractions
        // only have to be the SAME value wherever two wavefronts meet — float division will do)
        uint32_t W = 0;
        for (uint32_t j = 0; j < n_terms; ++j)
          W += rng[2u * ntile * kMaxTerms + j] - rng[j];
        const uint32_t lo = uint32_t(uint64_t(W) * tid / nw), hi = uint32_t(uint64_t(W) * (tid + 1u) / nw);
        auto frac = [](uint32_t x, uint32_t N) -> uint32_t {   // x < N
          const float f = (float(x) / float(N)) * 4294967296.f;
          return f >= 4294967040.f ? 0xFFFFFF00u : uint32_t(f);
        };
        FastShare s{1u, 0u, 0u, 0u};   // (j0 > j1: nothing)
        if (hi > lo) {
          uint32_t P = 0;
          bool open = false;
          for (uint32_t j = 0; j < n_terms; ++j) {
            const uint32_t N = rng[2u * ntile * kMaxTerms + j] - rng[j];
            if (!N) continue;
            if (!open && P + N > lo) {
              open = true;
              s.j0 = j;
              s.f_lo = frac(lo - P, N);
            }
            if (open && P < hi) {
              s.j1 = j;
              s.f_hi = (hi - P >= N) ? 0xFFFFFFFFu : frac(hi - P, N);
            }
            P += N;
          }
        }
        share[tid] = s;
      }
      __syncthreads();
      hit_pk = fast_tiles(smem, args, ntile);
    }
    // ---- chunk hand-over (as k_join_score): flush the PREVIOUS chunk's staged candidates,
    // reserve slots for this chunk's, publish the matches
    const uint32_t cap = args->cand_cap;
    if (tid == 0) {
      vars[kJPendQ] = pend_q;
      vars[kJPendBase] = pend_base;
      vars[kJPendN] = pend_n;
      uint32_t g = group;
      vars[kJChunk] = pull(group, next_raw, g);
      vars[kJGroup] = g;
    }
    uint32_t my_hits = wave::reduce_add((hit_pk & 0xFFFFu) + (hit_pk >> 16));
    if (lane == 0 && my_hits)
      atomicAdd(&args->hits[q], static_cast<unsigned long long>(my_hits));
    __syncthreads();
    {
      const uint32_t pn = vars[kJPendN];
      if (pn) {
        const uint32_t pq = vars[kJPendQ], gbase = vars[kJPendBase];
        const uint64_t* pl = lcand + (parity ^ 1u) * kJoinCands;
        uint64_t* out = args->cands + uint64_t(pq) * cap;
        for (uint32_t i = tid; i < pn; i += blockDim.x) {
          const uint32_t g = gbase + i;
          if (g < cap) out[g] = pl[i];
        }
      }
    }
    chunk = wave::uniform(vars[kJChunk]);
    group = wave::uniform(vars[kJGroup]);
    if (tid == 0) {
      const uint32_t raw = vars[kJNc + parity];
      pend_n = raw < kJoinCands ? raw : kJoinCands;
      pend_q = q;
      pend_base = pend_n ? atomicAdd(&args->cand_count[q], pend_n) : 0u;
    }
    __syncthreads();   // everyone has read the hand-over words and the previous staging buffer
    parity ^= 1u;
    if (tid == 0) vars[kJNc + parity] = 0u;
  }
  if (tid == 0) {
    vars[kJPendQ] = pend_q;
    vars[kJPendBase] = pend_base;
    vars[kJPendN] = pend_n;
  }
  __syncthreads();
  {
    const uint32_t cap = args->cand_cap;
    const uint32_t pn = vars[kJPendN];
    if (pn) {
      const uint32_t pq = vars[kJPendQ], gbase = vars[kJPendBase];
      const uint64_t* pl = lcand + (parity ^ 1u) * kJoinCands;
      uint64_t* out = args->cands + uint64_t(pq) * cap;
      for (uint32_t i = tid; i < pn; i += blockDim.x) {
        const uint32_t g = gbase + i;
        if (g < cap) out[g] = pl[i];
      }
    }
  }
}

// ---------------------------------------------------------------- rescore --

// The fixed-point contribution of one exact entry to its doc's score: what join_post adds
// (table row where every frequency of the term has one, else the general expression).
__device__ __forceinline__ uint32_t join_fixed(const float* caches, uint32_t e, float cs, uint32_t mode) {
  const uint32_t tabofs = mode & kJoinTabMask;
  const int form = join_form(mode);
  if (form == kJTable) {
    const float t = caches[((e & 0xFFFFu) | tabofs) >> 2];
    return static_cast<uint32_t>(wave::fma(cs, t, 1.f));
  }
  const float t = caches[((e & 0x3FCu) | tabofs) >> 2];
  const float tf = static_cast<float>((e >> 10) & kJoinTfMax);
  const float scaled = (form == kJSqrt) ? wave::fast_sqrt(tf) * cs * t
                                        : wave::fma(-cs, wave::fast_rcp(wave::fma(tf, t, 1.f)), cs);
  return static_cast<uint32_t>(scaled) | 1u;
}

struct RescoreArgs {
  const uint32_t* units;       // the launch's units
  const DevQuery* queries;
  const DevQTerm* qterms;
  const JoinTerm* jterms;
  const uint32_t* bstar;
  const uint64_t* acands;      // [unit][cap] k_join_fast's candidates
  const uint32_t* acand_count;
  uint64_t* cands;             // [unit][cap] exact keys for k_select
  uint32_t* cand_count;
  uint32_t cand_cap;
};

// One workgroup per unit.  (1) A coarse histogram of the candidates' approximate scores finds a
// lower bound L of the k-th largest; a doc of the exact top k has A >= L - kFastKeep * n (any k
// docs with A > A_d + 5 n have exact scores above doc d's: |A - X| < 3 n from below, 2 n from
// above; one more n for the fixed-point sums' own rounding) — everything else is dropped unseen.
// (2) The kept docs' postings are looked up in the exact entry streams (binary search inside the
// doc's tile, one thread per (doc, term)) and summed in fixed point exactly as join.h does; docs
// whose exact score reaches the threshold bin become candidates of k_select — the same set, bit
// for bit, the one-pass kernel emits among those that can still matter.
__global__ void __launch_bounds__(kTileThreadsMax)
k_join_rescore(const RescoreArgs* __restrict__ args) {
  RT_DYN_SMEM(smem);
  // LDS: qts | caches (build_tables) | hist / acc | docs
  DevQTerm* qts = reinterpret_cast<DevQTerm*>(smem);
  float* caches = reinterpret_cast<float*>(smem + sizeof(DevQTerm) * kMaxTerms);
  uint32_t* racc = reinterpret_cast<uint32_t*>(caches + 256u * kTableRows);   // [kFastRound] (first: hist[kFastBins])
  uint32_t* rdoc = racc + kFastRound;                                          // [kFastRound]
  __shared__ uint32_t s_n, s_cut, s_out;
  const uint32_t tid = threadIdx.x;
  const uint32_t q = args->units[blockIdx.x];
  const DevQuery qd = args->queries[q];
  const uint32_t cap = args->cand_cap;
  const uint32_t n_raw = args->acand_count[q];
  if (n_raw > cap) {   // the approximate candidates did not fit: k_select reports the overflow
    if (tid == 0) args->cand_count[q] = n_raw;   // (what the host grows the buffers to)
    return;
  }
  const uint32_t n = n_raw;
  const uint64_t* src = args->acands + uint64_t(q) * cap;
  const uint32_t n_terms = qd.n_terms;
  if (tid < n_terms) qts[tid] = args->qterms[qd.first_term + tid];
  for (uint32_t i = tid; i < kFastBins; i += blockDim.x) racc[i] = 0u;
  if (tid == 0) { s_cut = 0u; s_out = 0u; }
  __syncthreads();
  {
    JoinSm sm;
    sm.qts = qts;
    sm.caches = caches;
    build_tables(sm, qd.n_caches, n_terms);
  }
  if (n > qd.k) {
    for (uint32_t i = tid; i < n; i += blockDim.x)
      atomicAdd(&racc[uint32_t(src[i] >> 32) >> 5], 1u);
    __syncthreads();
    if (tid < 64u) {   // suffix search over 1024 bins: lane L owns the 16 bins of chunk 63 - L
      const uint32_t c = 63u - tid;
      uint32_t s = 0;
      for (uint32_t i = 0; i < kFastBins / 64u; ++i) s += racc[c * (kFastBins / 64u) + i];
      const uint32_t incl = wave::inclusive_scan(s);
      const uint64_t reach = wave::ballot(incl >= qd.k);
      if (reach) {
        const int srcl = __builtin_ctzll(reach);
        const uint32_t above = wave::bcast(incl - s, srcl);
        const uint32_t cc = 63u - uint32_t(srcl);
        if (tid == 0) {
          uint32_t cum = above, bin = cc * (kFastBins / 64u);
          for (int i = int(kFastBins / 64u) - 1; i >= 0; --i) {
            cum += racc[cc * (kFastBins / 64u) + uint32_t(i)];
            if (cum >= qd.k) { bin = cc * (kFastBins / 64u) + uint32_t(i); break; }
          }
          const uint32_t edge = bin << 5, slack = kFastKeep * n_terms;
          s_cut = edge > slack ? edge - slack : 0u;
        }
      }
    }
  }
  __syncthreads();
  const uint32_t cut = s_cut;
  const uint32_t bs = args->bstar[q];
  uint64_t* out = args->cands + uint64_t(q) * cap;
  // rounds of kFastRound candidates (of which the kept ones are re-scored)
  for (uint32_t r0 = 0; r0 < n; r0 += kFastRound) {
    __syncthreads();
    if (tid == 0) s_n = 0u;
    __syncthreads();
    const uint32_t r1 = r0 + kFastRound < n ? r0 + kFastRound : n;
    for (uint32_t i = r0 + tid; i < r1; i += blockDim.x) {
      const uint64_t key = src[i];
      if (uint32_t(key >> 32) >= cut) {
        const uint32_t slot = atomicAdd(&s_n, 1u);
        rdoc[slot] = uint32_t(key);
        racc[slot] = 0u;
      }
    }
    __syncthreads();
    const uint32_t kept = s_n;
    // one thread per (kept doc, term)
    for (uint32_t w = tid; w < kept * n_terms; w += blockDim.x) {
      const uint32_t c = w / n_terms, j = w % n_terms;
      const uint32_t doc = rdoc[c];
      const JoinTerm jt = args->jterms[qd.first_term + j];
      const uint32_t tile = (doc - kDocMin) / kJoinTile, idx = (doc - kDocMin) % kJoinTile;
      const uint32_t* bnd = reinterpret_cast<const uint32_t*>(jt.bounds);
      const uint32_t* ent = reinterpret_cast<const uint32_t*>(jt.entries);
      uint32_t a = bnd[tile], b = bnd[tile + 1u];
      while (a < b) {   // lower_bound of idx among the tile's entries (sorted by doc)
        const uint32_t m = (a + b) >> 1;
        if ((ent[m] >> 18) < idx) a = m + 1u; else b = m;
      }
      if (a < bnd[tile + 1u]) {
        const uint32_t e = ent[a];
        if ((e >> 18) == idx) atomicAdd(&racc[c], join_fixed(caches, e, jt.cs, jt.mode));
      }
    }
    __syncthreads();
    for (uint32_t c = tid; c < kept; c += blockDim.x) {
      const uint32_t f = racc[c];
      const float v = f <= kMaxTerms ? 0.f : from_fixed<uint32_t>(f, qd.fx_inv);
      if (f && score_bin(v, qd.bin_scale) >= bs) {
        const uint32_t slot = atomicAdd(&s_out, 1u);
        out[slot] = make_key(v, rdoc[c]);   // (slot < n <= cap)
      }
    }
  }
  __syncthreads();
  if (tid == 0) args->cand_count[q] = s_out;
}

constexpr size_t kRescoreSmem = sizeof(DevQTerm) * kMaxTerms + 4u * 256u * kTableRows + 8u * kFastRound;

}  // namespace irs_hip
