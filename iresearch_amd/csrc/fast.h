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
// The fast entries are a copy of the exact ones (k_fast_pack: same postings, scored once for all
// queries of the batch) with every (term, half tile) padded to a MULTIPLE OF FOUR entries — a lane
// takes four consecutive entries with one 16-byte load and never has to tell its entries apart.
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
constexpr uint32_t kFastSlots = 4;               // 256-entry loads a wavefront has in flight per tile
constexpr uint32_t kFastPackTiles = 16;          // half tiles per k_fast_pack workgroup

// LDS layout of k_join_fast (byte offsets; the accumulators sit at LDS address 0: an entry's low
// 16 bits ARE the byte address of its doc's word)
struct alignas(16) FastSlot {   // one 256-entry load of a wavefront's tile
  uint32_t lo, hi;   // address of its first entry
  uint32_t cnt;      // entries from there to the end of its piece (a multiple of 4; 0: none)
  uint32_t csq;      // the term's multiplier
};
struct FastOff {
  static constexpr uint32_t acc = 0;                                      // [kJoinTile] u32 = 2 x u16
  static constexpr uint32_t dummy = 4u * kJoinTile;                       // [64] u32: a word per lane nobody reads
  static constexpr uint32_t rng = dummy + 256u;                           // [2 * chunk tiles + 1][kMaxTerms] u32
  static constexpr uint32_t fts = rng + 4u * (2u * kFastChunkTiles + 1u) * kMaxTerms;   // FastTerm[kMaxTerms]
  static constexpr uint32_t share = fts + 32u * kMaxTerms;                // [16 waves][4] u32
  static constexpr uint32_t ord = share + 16u * 16u;                      // [kMaxTerms] u32: the terms by decreasing size
  static constexpr uint32_t cand = ord + 4u * kMaxTerms;                  // [2][kJoinCands] u64
  static constexpr uint32_t vars = cand + 8u * 2u * kJoinCands;           // [16] u32
  static constexpr uint32_t desc = vars + 64u;                            // [16 waves][chunk tiles][kFastSlots] FastSlot
  static constexpr uint32_t end = desc + 16u * kFastChunkTiles * kFastSlots * uint32_t(sizeof(FastSlot));
};
static_assert(FastOff::cand % 8u == 0u && FastOff::desc % 16u == 0u, "alignment of the staging keys / slot records");
static_assert(4u * kJoinTile + 256u <= 65536u, "an entry's 16 address bits reach every word and the dummies");
static_assert(2u * FastOff::end <= 160u * 1024u, "two workgroups per CU");

// Per (unit, term): where the term's fast entries and their tile boundaries are, its multiplier
struct alignas(16) FastTerm {
  uint64_t fent;
  uint64_t fbounds;
  uint32_t csq, pad[3];
};
static_assert(sizeof(FastTerm) == 32, "FastTerm");

struct FastArgs {
  const DevQuery* queries;
  const FastTerm* fterms;      // parallel to DevQTerm / JoinTerm
  uint64_t dummies;            // address of 64 x 4 entries that add nothing: lane l's at + 16 l
  const uint32_t* bstar;
  uint64_t* cands;             // [unit][cap] (approximate score << 32) | doc
  uint32_t* cand_count;
  unsigned long long* hits;
  const uint32_t* order;       // as JoinArgs: one work queue per XCD
  uint32_t* work_counter;
  uint32_t base[kJoinQueues + 1];
  uint32_t first[kJoinQueues + 1];
  uint32_t cpq, n_units, nw_log2, cand_cap, chunk_tiles;
  const struct FastChunk* chunks;   // [chunk id] k_fast_shares
};

// ---------------------------------------------------------------- pack --

// Where the half tiles of a stream begin among its FAST entries: every half tile's entries
// rounded up to a multiple of four.  One wavefront per stream.
__global__ void __launch_bounds__(64)
k_fast_layout(const StreamRec* streams, uint32_t n_streams) {
  if (blockIdx.x >= n_streams) return;
  const StreamRec S = streams[blockIdx.x];
  if (!S.fent) return;
  const unsigned lane = threadIdx.x;
  const uint32_t* bnd = reinterpret_cast<const uint32_t*>(S.bounds);
  uint32_t* fb = reinterpret_cast<uint32_t*>(S.fbounds);
  uint32_t carry = 0;
  for (uint32_t t0 = 0; t0 <= S.n_tiles; t0 += 64u) {
    const uint32_t t = t0 + lane;
    const uint32_t c = t < S.n_tiles ? ((bnd[t + 1u] - bnd[t] + 3u) & ~3u) : 0u;
    const uint32_t incl = wave::inclusive_scan(c);
    if (t <= S.n_tiles) fb[t] = carry + incl - c;
    carry += wave::bcast(incl, 63);
  }
}

// The fast entries themselves: one workgroup per kFastPackTiles half tiles of a stream; an exact
// entry (idx : 14 | tf : 6 | norm : 8 | 00) becomes (T / Tn * 2^16 : 16 | idx * 4 : 16) at its
// padded place, the padding adds nothing to a word nobody reads.
__global__ void __launch_bounds__(kThreads)
k_fast_pack(const StreamRec* streams, const uint32_t* first_wg /*[stream + 1]*/, uint32_t n_streams) {
  // which stream: binary search of the workgroup in the streams' first workgroups
  uint32_t a = 0, b = n_streams;
  while (b - a > 1u) {
    const uint32_t m = (a + b) >> 1;
    if (first_wg[m] <= blockIdx.x) a = m; else b = m;
  }
  const StreamRec S = streams[a];
  if (!S.fent) return;
  const uint32_t* bnd = reinterpret_cast<const uint32_t*>(S.bounds);
  const uint32_t* fb = reinterpret_cast<const uint32_t*>(S.fbounds);
  const uint32_t* ent = reinterpret_cast<const uint32_t*>(S.entries);
  uint32_t* fent = reinterpret_cast<uint32_t*>(S.fent);
  const uint32_t t0 = (blockIdx.x - first_wg[a]) * kFastPackTiles;
  const uint32_t t1 = t0 + kFastPackTiles < S.n_tiles ? t0 + kFastPackTiles : S.n_tiles;
  // the group's entries are one contiguous run of the exact stream; its (at most 17) boundaries
  // go to LDS, a thread finds an entry's half tile by bisection there
  __shared__ uint32_t s_b[kFastPackTiles + 1u], s_fb[kFastPackTiles + 1u];
  const uint32_t nt = t1 - t0;
  if (threadIdx.x <= nt) {
    s_b[threadIdx.x] = bnd[t0 + threadIdx.x];
    s_fb[threadIdx.x] = fb[t0 + threadIdx.x];
  }
  __syncthreads();
  const uint32_t p0 = s_b[0], p1 = s_b[nt];
  const int32_t kind = S.kind;
  const float nc = S.nc, nl = S.nl;
  for (uint32_t p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
    uint32_t lo = 0, hi = nt;   // the tile i with s_b[i] <= p < s_b[i + 1]
    while (hi - lo > 1u) {
      const uint32_t m = (lo + hi) >> 1;
      if (s_b[m] <= p) lo = m; else hi = m;
    }
    const uint32_t e = ent[p];
    fent[s_fb[lo] + (p - s_b[lo])] = fast_entry(e >> 18, fast_unit(kind, nc, nl, join_tf(e), (e >> 2) & 0xFFu));
  }
  if (threadIdx.x < 4u * nt) {   // <= 3 pad entries per half tile
    const uint32_t i = threadIdx.x >> 2, k = threadIdx.x & 3u;
    const uint32_t c = s_b[i + 1u] - s_b[i];
    if (c + k < ((c + 3u) & ~3u)) fent[s_fb[i] + c + k] = FastOff::dummy + 4u * k;
  }
}

// ---------------------------------------------------------------- shares --

// what a wavefront takes of a chunk's entries: from fraction f_lo of term j0 to fraction f_hi of
// term j1 (32-bit fixed point; 0xFFFFFFFF = the whole term), every term in between whole; the
// terms are named by their place in the chunk's order (FastChunk::ord)
struct FastShare {
  uint32_t j0, j1, f_lo, f_hi;
};
// (cuts fall on multiples of four entries: so do the half tiles' boundaries, k_fast_layout)
__device__ __forceinline__ uint32_t fast_cut(uint32_t n, uint32_t f) {
  return f == 0xFFFFFFFFu ? n : (wave::mul_hi(n, f) & ~3u);
}
struct FastChunk {
  FastShare share[16];   // per wavefront of the workgroup
  uint32_t ord[16];      // the query's terms with entries in the chunk, largest first
};

// The wavefronts' shares of EVERY chunk of the launch, one thread per chunk id: the terms in
// order of decreasing size, the sequence of their entries cut into at most nw consecutive shares
// of at most T entries that touch at most TWO terms each — the smallest such T (bisection).
// (Equal shares put the query's rare terms — several of them, a few entries each — into ONE
// wavefront, which then walks ten pieces per tile while fifteen others wait at the barrier.
// A kernel of its own: the bisection is serial work, a few microseconds per chunk — spread over
// 26 000 threads it costs nothing, inside the persistent workgroups it would cost them all.)
__global__ void __launch_bounds__(kThreads)
k_fast_shares(const FastArgs* __restrict__ args, FastChunk* out) {
  const uint32_t chunk = blockIdx.x * blockDim.x + threadIdx.x;
  if (chunk >= args->base[kJoinQueues]) return;
  uint32_t group = 0;
  while (group + 1u < kJoinQueues && chunk >= args->base[group + 1u]) ++group;
  const uint32_t n_units = args->first[group + 1u] - args->first[group];
  const uint32_t local = chunk - args->base[group];
  const uint32_t q = args->order[args->first[group] + local % n_units];
  const uint32_t tile0 = (local / n_units) * args->chunk_tiles;
  const DevQuery qd = args->queries[q];
  const uint32_t n_half = qd.n_tiles, n_tiles = (n_half + 1u) >> 1;
  const uint32_t ntile = tile0 >= n_tiles ? 0u
                         : ((n_tiles - tile0) < args->chunk_tiles ? (n_tiles - tile0) : args->chunk_tiles);
  const uint32_t nw = 1u << args->nw_log2;
  FastChunk rec;
  for (uint32_t w = 0; w < 16u; ++w) {
    rec.share[w] = FastShare{1u, 0u, 0u, 0u};   // (j0 > j1: nothing)
    rec.ord[w] = 0u;
  }
  uint32_t N[kMaxTerms];
  uint32_t W = 0, nt = 0;
  if (ntile) {
    const uint32_t h0 = 2u * tile0, h1 = 2u * (tile0 + ntile) < n_half ? 2u * (tile0 + ntile) : n_half;
    for (uint32_t j = 0; j < qd.n_terms; ++j) {
      const uint32_t* fb = reinterpret_cast<const uint32_t*>(args->fterms[qd.first_term + j].fbounds);
      const uint32_t n = fb[h1] - fb[h0];
      if (!n) continue;
      uint32_t at = nt++;
      for (; at > 0 && N[at - 1] < n; --at) {   // insertion sort, descending
        N[at] = N[at - 1];
        rec.ord[at] = rec.ord[at - 1];
      }
      N[at] = n;
      rec.ord[at] = j;
      W += n;
    }
  }
  // (fewer wavefronts than half the terms: as many terms per share as it takes)
  const uint32_t max_terms = (nt + nw - 1u) / nw > 2u ? (nt + nw - 1u) / nw : 2u;
  auto shares = [&](uint32_t T, FastShare* dst) -> uint32_t {   // shares needed with at most T entries each
    auto frac = [](uint32_t x, uint32_t n) -> uint32_t {   // x < n
      const float f = (float(x) / float(n)) * 4294967296.f;
      return f >= 4294967040.f ? 0xFFFFFF00u : uint32_t(f);
    };
    uint32_t count = 0, t = 0, o = 0;
    while (t < nt) {
      FastShare sh{t, t, frac(o, N[t]), 0u};
      uint32_t left = T, touched = 0;
      while (t < nt && left > 0u && touched < max_terms) {
        const uint32_t take = N[t] - o < left ? N[t] - o : left;
        o += take;
        left -= take;
        ++touched;
        sh.j1 = t;
        if (o == N[t]) {
          sh.f_hi = 0xFFFFFFFFu;
          ++t;
          o = 0;
        } else {
          sh.f_hi = frac(o, N[t]);
        }
      }
      if (dst && count < nw) dst[count] = sh;
      ++count;
    }
    return count;
  };
  if (W) {
    uint32_t lo = (W + nw - 1u) / nw, hi = W;   // (T = W: a share per max_terms terms: <= nw of them)
    for (int it = 0; it < 12 && lo < hi; ++it) {
      const uint32_t mid = lo + (hi - lo) / 2u;
      if (shares(mid, nullptr) <= nw) hi = mid; else lo = mid + 1u;
    }
    shares(hi, rec.share);
  }
  out[chunk] = rec;
}

// ---------------------------------------------------------------- tiles --

// four entries per lane: the contribution goes to the low (shift 0) or the high (16) half of
// every doc's word
template<uint32_t SHIFT>
__device__ __forceinline__ void fast_post4(const unsigned char* lds, const uint32_t (&e)[4], uint32_t csq) {
  uint32_t c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) c[k] = (wave::mul_hi(e[k], csq) << SHIFT) + (1u << SHIFT);
#pragma unroll
  for (int k = 0; k < 4; ++k) wave::lds_add(lds, FastOff::acc + (e[k] & 0xFFFCu), c[k]);
}
// `count` (a multiple of 4) consecutive entries from `base` (wave-uniform), 256 per step — the
// rare tail of a piece behind the slots of its tile
template<uint32_t SHIFT>
__device__ __forceinline__ void fast_run(const unsigned char* lds, uint64_t base, uint64_t dummies, uint32_t count,
                                         uint32_t csq, unsigned lane) {
  while (count) {
    uint32_t e[4];
    const uint64_t at = (4u * lane < count ? base : dummies) + 16u * lane;
    wave::gload_u32x4_at(at, e);
    fast_post4<SHIFT>(lds, e, csq);
    base += 1024u;
    count = count > 256u ? count - 256u : 0u;
  }
}

enum : uint32_t {   // more LDS scratch words (behind join.h's kJ*)
  kFUnit = 8,       // the chunk's unit
  kFThr = 9,        // its threshold in 16-bit units
  kFDoc0 = 10,      // first doc of the chunk's first tile
  kFParity = 11,    // which candidate staging buffer the chunk fills
};

// The tiles of one chunk.  Everything it needs lives in LDS (tile boundaries, term records, the
// wavefronts' shares, the chunk's scalars) — a function of its own so that the persistent
// kernel's bookkeeping does not compete with the tile loop for registers.
//
// A wavefront's entries of a tile are PIECES: per term of its share the part in the tile's low
// half (contributions to the low halves of the words) and the part in its high half.  They are
// dealt to kFastSlots = 4 SLOTS of up to 256 entries — slots 0, 1 low half, slots 2, 3 high half:
// a share inside ONE term (the rule) puts the first 512 entries of each half there, a share over
// two terms the first 256 of each term and half — whose records (address, count, multiplier) are
// worked out once per chunk, lane-parallel, into LDS.  The tile loop is then straight-line code:
// four records, four 16-byte loads per lane (lanes past a slot's end read entries that add nothing:
// an address select, no mask), requested BEFORE the previous tile's barrier so that they fly
// behind its epilogue; sixteen multiply-adds into LDS; no scalar arithmetic, no branch.  What
// does not fit the slots (a piece longer than its slots, a share over three terms and more) is
// flagged per tile and handled by a loop behind the slots — rare by construction (k_fast_shares).
// Returns the lane's match count: low half | high half.
__device__ __attribute__((noinline)) uint32_t fast_tiles(unsigned char* smem, const FastArgs* args,
                                                         uint32_t ntile_arg) {
  const uint32_t* rng = reinterpret_cast<const uint32_t*>(smem + FastOff::rng);
  const FastTerm* fts = reinterpret_cast<const FastTerm*>(smem + FastOff::fts);
  const uint32_t* ord = reinterpret_cast<const uint32_t*>(smem + FastOff::ord);
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
  const uint64_t dummies = args->dummies;
  const uint32_t n_share = j1 >= j0 ? j1 - j0 + 1u : 0u;   // terms of the share
  // entries [from, to) of the share's term number jp (0 ..) in half `h` of tile u
  auto piece = [&](uint32_t u, uint32_t jp, uint32_t h, uint32_t& j, uint32_t& from, uint32_t& to) {
    j = ord[j0 + jp];
    const uint32_t a0 = rng[(2u * u) * kMaxTerms + j];
    const uint32_t b1 = rng[(2u * u + 1u) * kMaxTerms + j];
    const uint32_t a2 = rng[(2u * u + 2u) * kMaxTerms + j];
    const uint32_t n = a2 - a0;
    const uint32_t lo = a0 + (jp == 0u ? fast_cut(n, f_lo) : 0u);
    const uint32_t hi = a0 + (jp + 1u == n_share ? fast_cut(n, f_hi) : n);
    const uint32_t mid = b1 < lo ? lo : (b1 > hi ? hi : b1);
    from = h ? mid : lo;
    to = h ? hi : mid;
    if (to < from) to = from;
  };
  // ---- the chunk's slot records: lane 4 u + s works out slot s of tile u
  FastSlot* desc = reinterpret_cast<FastSlot*>(smem + FastOff::desc) + wv * (kFastChunkTiles * kFastSlots);
  bool over = false;   // (per lane) the slot's piece is longer than the slots hold
  {
    const uint32_t u = lane >> 2, sl = lane & 3u;
    FastSlot D{uint32_t(dummies), uint32_t(dummies >> 32), 0u, 0u};
    if (u < ntile && n_share) {
      // one term: slots (0, 1) = its low half's first and second 256 entries, (2, 3) its high
      // half's; two terms and more: slot (0, 2) = the first term's halves, (1, 3) the second's
      const uint32_t h = sl >> 1;
      const uint32_t jp = n_share == 1u ? 0u : (sl & 1u);
      const uint32_t skip = n_share == 1u ? 256u * (sl & 1u) : 0u;
      if (jp < n_share) {
        uint32_t j, from, to;
        piece(u, jp, h, j, from, to);
        if (to > from + skip) {
          const uint64_t at = fts[j].fent + 4ull * (from + skip);
          D.lo = uint32_t(at);
          D.hi = uint32_t(at >> 32);
          D.cnt = to - from - skip;
          D.csq = fts[j].csq;
          over = D.cnt > (n_share == 1u && !(sl & 1u) ? 512u : 256u);
        }
      }
    }
    desc[lane] = D;
  }
  // tiles with something behind their slots: a long piece, or a share over three terms and more
  const uint64_t slow_tiles = n_share > 2u ? ~0ull : wave::ballot(over);
  wave::sync();   // (the records are read by every lane of the wavefront)
  uint32_t hit_pk = 0;
  const uint32_t lane16 = 16u * lane, lane4 = 4u * lane;
  uint32_t r0[4], r1[4], r2[4], r3[4];
  uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;   // the slots' multipliers
  // the four loads of tile u: records first, then the addresses, then the loads back to back
  auto request = [&](uint32_t u) {
    const FastSlot* d = desc + kFastSlots * (u < ntile ? u : 0u);
    const FastSlot d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];
    const bool live = u < ntile;
    uint64_t a0 = ((live && lane4 < d0.cnt) ? ((uint64_t(d0.hi) << 32) | d0.lo) : dummies) + lane16;
    uint64_t a1 = ((live && lane4 < d1.cnt) ? ((uint64_t(d1.hi) << 32) | d1.lo) : dummies) + lane16;
    uint64_t a2 = ((live && lane4 < d2.cnt) ? ((uint64_t(d2.hi) << 32) | d2.lo) : dummies) + lane16;
    uint64_t a3 = ((live && lane4 < d3.cnt) ? ((uint64_t(d3.hi) << 32) | d3.lo) : dummies) + lane16;
    wave::keep64(a0);
    wave::keep64(a1);
    wave::keep64(a2);
    wave::keep64(a3);
    q0 = d0.csq; q1 = d1.csq; q2 = d2.csq; q3 = d3.csq;
    wave::gload_u32x4_at(a0, r0);
    wave::gload_u32x4_at(a1, r1);
    wave::gload_u32x4_at(a2, r2);
    wave::gload_u32x4_at(a3, r3);
  };
  request(0);
  for (uint32_t u = 0; u < ntile; ++u) {
    // ---- accumulate this wavefront's entries of tile u
    fast_post4<0u>(smem, r0, q0);
    fast_post4<0u>(smem, r1, q1);
    fast_post4<16u>(smem, r2, q2);
    fast_post4<16u>(smem, r3, q3);
    if ((slow_tiles >> (4u * u)) & 0xFull) {   // (rare) what the slots do not hold
      for (uint32_t jp = 0; jp < n_share; ++jp) {
        for (uint32_t h = 0; h < 2u; ++h) {
          uint32_t j, from, to;
          piece(u, jp, h, j, from, to);
          j = wave::uniform(j);
          from = wave::uniform(from);
          to = wave::uniform(to);
          const uint32_t held = n_share == 1u ? 512u : (jp < 2u ? 256u : 0u);
          if (to > from + held) {
            const uint64_t base = wave::uniform64(fts[j].fent) + 4ull * (from + held);
            const uint32_t csq = wave::uniform(fts[j].csq);
            if (h) fast_run<16u>(smem, base, dummies, to - from - held, csq, lane);
            else fast_run<0u>(smem, base, dummies, to - from - held, csq, lane);
          }
        }
      }
    }
    request(u + 1u);
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
    // (one exchange at a time: the requested entries of the next tile occupy the registers a
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
  FastTerm* fts = reinterpret_cast<FastTerm*>(smem + FastOff::fts);
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
        FastTerm ft{0, 0, 0, {0, 0, 0}};
        if (tid < n_terms) ft = args->fterms[qd.first_term + tid];
        fts[tid] = ft;
      }
      for (uint32_t e = tid; e < (2u * ntile + 1u) * kMaxTerms; e += blockDim.x) {
        const uint32_t i = e / kMaxTerms, j = e % kMaxTerms;
        uint32_t v = 0;
        if (j < n_terms) {
          uint32_t h = 2u * tile0 + i;
          h = h < n_half ? h : n_half;
          v = reinterpret_cast<const uint32_t*>(args->fterms[qd.first_term + j].fbounds)[h];
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
      // the wavefronts' shares of the chunk and the order of its terms: worked out for every
      // chunk of the launch by k_fast_shares
      if (tid < 16u) {
        const FastChunk* rec = args->chunks + chunk;
        share[tid] = rec->share[tid];
        reinterpret_cast<uint32_t*>(smem + FastOff::ord)[tid] = rec->ord[tid];
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
  const float tf = static_cast<float>(join_tf(e));
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
