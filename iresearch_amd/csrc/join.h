// join.h — disjunctions over JOINED POSTING STREAMS: every distinct (segment, term) of a
// batch is decoded ONCE per run, whatever the number of queries that use it.
//
//   k_join        decode + norm join: one 4-byte entry per posting
//                    [ (doc - tile's first doc) * 4 : 16 | tf & 63 : 6 | norm : 8 | tf >> 6 : 2 ]
//                 in posting order, plus the entry index at which every doc tile starts
//   k_join_pilot  scores every P-th doc tile, derives a per-query score-bin threshold
//   k_join_score  accumulates the streams of a query's terms tile by tile in LDS, emits the
//                 candidates above the threshold, counts hits
//
// Replaces, like score.h, block_disjunction::refill (disjunction.hpp:1240-1351) and
// basic_disjunction (:204-358) — and it is literally the pipeline BASELINE.json's north_star
// names: a decode kernel (bit-exact doc ids and frequencies, wavefront prefix sum) followed by
// the candidate / score-accumulation kernel.  Why a second organisation next to score.h's work
// items: PMC showed k_score bound by instruction issue with ~70 wave-instructions per 64
// postings, 24 of them the decode itself and the rest per-(query, tile, block) bookkeeping,
// while the queries of a batch share their frequent terms (log-uniform ranks: a top-octave
// term is used by ~60 of 1000 queries).  An entry is query independent — the score of a
// posting is c0 * T[tf][norm] with a table that only depends on the scorer — so a query's
// posting costs: one coalesced 4-byte load, one table read, one multiply-add, one LDS add.
//
// The score arithmetic is score.h's (same tables, same fixed point, the same classification of a
// posting into table row / general expression): the hits of plain disjunctions are BIT-IDENTICAL
// to the work-item path's (tests: case_paths_agree).  Units that count matches in their
// accumulators' low bits (conjunctions, min-match: COUNT) round every contribution to a multiple
// of 16 units — within the parity tolerance where they are allowed to join (count_precise).
//
// Eligibility (anything else runs on score.h's / conj.h's kernels): sum merge, scorers of the
// table family over 1-byte norms or none, 32-bit accumulators, every term's frequencies below
// 256 (the entry's low 16 bits ARE the byte offset of T[tf][norm] for the terms whose frequencies
// all have table rows — tf < 16, low bits 0; a long document's tf >= 64 continues in the two low
// bits, which only the general expression reads).  A batch with ExecutionContext::wand set runs
// its joined units exhaustively (nothing is pruned here).  Conjunctions and
// min-match disjunctions of at most 15 terms join too where the rounding of their match-counting accumulators (join_post COUNT) stays below
// the parity tolerance and — conjunctions — where walking every entry of every term beats
// decoding only the blocks the rarest term's docs fall into (irs_hip.hip unit_joinable).
#pragma once
#include "score.h"

namespace irs_hip {

constexpr uint32_t kJoinTile = 12288;     // docs per accumulator tile (48 KB of u32 in LDS)
constexpr uint32_t kJoinTfMax = 255;      // entry layout: 6 + 2 bits of tf
constexpr uint32_t kJoinBlocks = 64;      // blocks per k_join workgroup
constexpr uint32_t kJoinChunkTiles = 64;  // consecutive tiles of one unit per work-queue item, at most
constexpr uint32_t kJoinChunkPlain = 32;  // ... of the 32-bit kernels (paired tiles take up to 64: 32 visits)
constexpr uint32_t kJoinCands = 256;      // candidate staging slots per chunk (x2 buffers)
constexpr uint32_t kJoinSlack = 1024;     // readable entries behind the last stream
constexpr uint32_t kJoinQueues = 8;       // work queues of k_join_score: one per XCD
// units that need per-doc match counts (conjunctions, min-match): the low 4 bits of an
// accumulator count matches (so at most 15 terms), contributions are rounded to multiples of 16
// ablation hooks of k_join_score (tools/build_variant.sh rewrites them; the product builds with 0):
// entry counts shifted right (what leaving out a share of the postings would buy), no hit
// counting, an epilogue that only clears / does nothing, accumulation without its LDS adds /
// table reads, empty shares that still wait for their requested entries, no barriers
constexpr uint32_t kAblShift = 0;
constexpr uint32_t kAblCount = 0;
constexpr uint32_t kAblEpi = 0;
constexpr uint32_t kAblNoAdd = 0;
constexpr uint32_t kAblNoTab = 0;
constexpr uint32_t kAblWait = 0;
constexpr uint32_t kAblNoBar = 0;
constexpr uint32_t kJoinCountMask = 15u;
constexpr uint32_t kJoinCountTerms = 15u;
constexpr float kJoinCountRound = 8.f;

enum : uint32_t {
  kJoinGeneral = 1u << 30,   // JoinTerm::mode: some tf of the term has no table row
  kJoinSqrt = 1u << 31,      //   square-root score form (TF-IDF family)
  kJoinTabMask = 0xFFFFu,    //   LDS byte offset of the term's table slot inside the tables
  kJoinHiHalf = 1u << 29,    // (paired tiles, lane view only) the pair's second tile: high 16 bits
};
// the shift of a paired-tile lane's contributions: 0 or 16
__host__ __device__ __forceinline__ uint32_t join_half_shift(uint32_t mode) { return (mode >> 25) & 16u; }

// One distinct (segment, term) of a batch: where its entries and tile boundaries live.
struct alignas(16) StreamRec {
  uint64_t entries;   // device address of the first entry (u32 each, posting order)
  uint64_t bounds;    // device address of bounds[0 .. n_tiles]: entry index of the first
                      // posting with doc >= tile's first doc; bounds[n_tiles] = n
  uint64_t pad64[2];
  uint32_t seg, term;
  uint32_t n;         // postings
  uint32_t n_tiles;   // doc tiles of the segment
  int32_t kind;       // the scorer signature the stream belongs to: Kind, norm_const,
  float nc, nl;       //   norm_length (one signature per stream: build_streams)
  uint32_t pad;
};
static_assert(sizeof(StreamRec) == 64, "StreamRec");
// k_join work: kJoinBlocks blocks (the tail counts as one) of a stream, and EVERYTHING the
// workgroup needs to know about it in one record, read by a scalar load: the kernel used to walk
// work item -> stream -> segment -> term -> directory (five dependent loads, ~4 us) before its
// first payload byte, for 16 blocks of work — it was bound by that chain, not by bytes.
struct alignas(16) JoinWg {
  uint64_t entries, bounds;   // of the stream (StreamRec)
  uint64_t doc;               // the term's first posting byte: DevSegment::doc + DevTerm::doc_start
  uint64_t dir;               // its first directory record: DevSegment::blk_dir + DevTerm::dir_off
  uint64_t pnorm;             // the norm byte of every posting of the term's full blocks, posting
                              // order, 128 per block (k_posting_norms; 0: no 1-byte Norm2 column)
  uint64_t tail_norms;        // ... and of its decoded tail
  uint64_t tail_docs, tail_freqs;   // the decoded tail / single doc (+ DevTerm::tail_row)
  uint32_t first;             // the workgroup's first block
  uint32_t nblk;              // full blocks of the list
  uint32_t tail_n;            // postings of the decoded tail (1 for a single-doc term)
  uint32_t tail_base;         // last doc in front of the tail (0: the list has no full block)
  uint32_t last_doc;          // of the list
  uint32_t n_tiles;           // doc tiles of the segment
  uint32_t n;                 // postings of the list
  uint32_t dead_lo, dead_hi;  // DevSegment::dead (the deleted-docs bitmap; 0: none)
  uint32_t pad;
  uint64_t pk;                // DevSegment::pk: the packed-payload image (16-byte aligned payloads)
  uint64_t pad2;
};
static_assert(sizeof(JoinWg) == 128, "JoinWg");
// Per (unit, term slot), parallel to DevQTerm: all k_join_score needs in one 32-byte record.
struct alignas(16) JoinTerm {
  uint64_t entries;
  uint64_t bounds;
  float cs;          // c0 * the unit's fixed-point scale
  uint32_t mode;     // table slot offset | kJoinGeneral | kJoinSqrt
  uint32_t pad[2];
};
static_assert(sizeof(JoinTerm) == 32, "JoinTerm");
constexpr uint32_t kJoinTermQuads = uint32_t(sizeof(JoinTerm)) / 16u;

__host__ __device__ __forceinline__ uint32_t join_entry(uint32_t idx, uint32_t tf, uint32_t norm) {
  return (idx << 18) | ((tf & 63u) << 10) | ((norm & 0xFFu) << 2) | ((tf >> 6) & 3u);
}
__host__ __device__ __forceinline__ uint32_t join_tf(uint32_t e) {
  return ((e >> 10) & 63u) | ((e & 3u) << 6);
}

// ------------------------------------------------------------------ join --

// One workgroup = kJoinBlocks consecutive blocks of one stream, in rounds of kWaves x
// kJoinPerWave: a wavefront owns every 4th block of a round and keeps its blocks of the round IN
// FLIGHT TOGETHER — the directory records, then every block's payload words, then every
// block's norm bytes: three memory round trips per round instead of three per block (a block is
// 100 VALU instructions behind 3 dependent loads).  Decode as decode.h (bit-exact doc ids +
// frequencies); entries are written coalesced (posting i is entry i).
// (Two blocks per wavefront, not four: 50 VGPRs instead of 93 let 8 wavefronts share a SIMD
// instead of 5 — 0.80 against 0.85 ms on the headline batch, bit-identical entries; one block
// 0.85, three 0.81, four with the registers capped by spilling 1.08 - 1.36: DESIGN §3.12.)
constexpr uint32_t kJoinPerWave = 2;
constexpr uint32_t kJoinRounds = kJoinBlocks / (kWaves * kJoinPerWave);
static_assert(kJoinRounds * kWaves * kJoinPerWave == kJoinBlocks, "k_join rounds");

// entries + tile boundaries of the two postings a lane holds of one block
__device__ __forceinline__ void join_emit(uint32_t* ent, uint32_t* bnd, uint32_t b, unsigned lane,
                                          uint32_t d0, uint32_t d1, uint32_t f0, uint32_t f1,
                                          uint32_t n0, uint32_t n1, bool v0, bool v1,
                                          uint32_t prev /*0: the list's first posting*/,
                                          const uint32_t* dead /*deleted docs, or null*/) {
  const uint32_t p0 = kBlock * b + 2u * lane;
  const uint32_t t0 = v0 ? (d0 - kDocMin) / kJoinTile : 0u;
  const uint32_t t1 = v1 ? (d1 - kDocMin) / kJoinTile : t0;
  uint32_t e0 = join_entry((d0 - kDocMin) - t0 * kJoinTile, f0, n0);
  uint32_t e1 = join_entry((d1 - kDocMin) - t1 * kJoinTile, f1, n1);
  if (dead) {   // (wave-uniform) a deleted doc's posting keeps its place in the stream and adds to
                // a dummy accumulator behind the tile's (JoinOff::dummy): no query ever sees it
    const uint32_t nowhere = (4u * kJoinTile + 4u * lane) << 16;
    if (v0 && doc_dead(dead, d0)) e0 = (e0 & 0xFFFFu) | nowhere;
    if (v1 && doc_dead(dead, d1)) e1 = (e1 & 0xFFFFu) | nowhere;
  }
  if (v1) {
    uint64_t both = (uint64_t(e1) << 32) | e0;
    __builtin_memcpy(ent + p0, &both, 8);
  } else if (v0) {
    ent[p0] = e0;
  }
  // tile boundaries: posting p opens every tile in (tile of posting p - 1, tile of p]
  const uint32_t up = __shfl_up(t1, 1, 64);
  const int32_t tp = lane ? int32_t(up) : (prev ? int32_t((prev - kDocMin) / kJoinTile) : -1);
  if (v0) {
    for (int32_t u = tp + 1; u <= int32_t(t0); ++u) bnd[u] = p0;
  }
  if (v1) {
    for (uint32_t u = t0 + 1u; u <= t1; ++u) bnd[u] = p0 + 1u;
  }
}

template<int LAYOUT>
__global__ void __launch_bounds__(kThreads)
k_join(const JoinWg* wgs) {
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t wv = wave::uniform(threadIdx.x >> 6);
  const JoinWg W = wave::sload<JoinWg>(reinterpret_cast<uint64_t>(wgs) + uint64_t(blockIdx.x) * sizeof(JoinWg));
  uint32_t* ent = reinterpret_cast<uint32_t*>(W.entries);
  uint32_t* bnd = reinterpret_cast<uint32_t*>(W.bounds);
  const uint8_t* doc = reinterpret_cast<const uint8_t*>(W.doc);
  const bool tiny = W.pnorm != 0;
  const uint32_t* dead = reinterpret_cast<const uint32_t*>((uint64_t(W.dead_hi) << 32) | W.dead_lo);
  const uint32_t nb = W.nblk + (W.tail_n ? 1u : 0u);
  uint32_t end = W.first + kJoinBlocks;
  if (end > nb) end = nb;
  // this wavefront's blocks of a round: b_i = r0 + wv + kWaves * i  (all wave-uniform); their
  // directory records by scalar loads — those of round r+1 are REQUESTED as soon as round r's
  // payload loads are out and arrive while round r decodes: one exposed round trip per round
  // (the payloads) instead of two (records, then payloads)
  auto records = [&](uint32_t r0, BlkDir (&out)[kJoinPerWave]) {
#pragma unroll
    for (uint32_t i = 0; i < kJoinPerWave; ++i) {
      const uint32_t b = r0 + wv + kWaves * i;
      out[i] = BlkDir{0u, 0u, 0u, 0u};
      if (b < end && b < W.nblk) out[i] = wave::sload<BlkDir>(W.dir + uint64_t(b) * sizeof(BlkDir));
    }
  };
  BlkDir dir[kJoinPerWave];
  records(W.first, dir);
  for (uint32_t r0 = W.first; r0 < end; r0 += kWaves * kJoinPerWave) {
    RawPair rd[kJoinPerWave], rf[kJoinPerWave];
    bool full[kJoinPerWave], plain[kJoinPerWave], packed[kJoinPerWave];
    uint32_t pn[kJoinPerWave];
#pragma unroll
    for (uint32_t i = 0; i < kJoinPerWave; ++i) {
      const uint32_t b = r0 + wv + kWaves * i;
      full[i] = b < end && b < W.nblk;
    }
#pragma unroll
    for (uint32_t i = 0; i < kJoinPerWave; ++i) {
      const uint32_t dbits = dir[i].bits & 0xFFu, fbits = dir[i].bits >> 8;
      // (an all-equal doc part is a vint of unknown length: that block decodes by itself below)
      plain[i] = full[i] && dbits != 0u;
      // both parts 1..31-bit packed (nearly every block of a frequent term): the 16-byte aligned
      // copy in the packed image through saddr loads, one funnel shift + bit-field extract per
      // value — the decoder the block-driven kernels use; `.doc` keeps serving the other framings
      // (round 6: k_join 0.93 -> 0.83 ms for the headline batch's 290 M postings, bit-identical)
      packed[i] = plain[i] && pk_both(dbits, fbits);
      if (packed[i]) {
        const uint64_t pl = W.pk + (uint64_t(dir[i].aoff) << 4);
        raw_load_packed_g<LAYOUT>(pl, dbits, lane, rd[i].a, rd[i].b);
        raw_load_packed_g<LAYOUT>(pl + 16u * dbits, fbits, lane, rf[i].a, rf[i].b);
      } else if (plain[i]) {
        const uint8_t* blk = doc + dir[i].off;
        rd[i] = raw_load<LAYOUT>(blk + 1, dbits, lane);
        rf[i] = raw_load<LAYOUT>(blk + 1u + 16u * dbits + 1u, fbits, lane);
      }
      pn[i] = 0u;
      if (full[i] && tiny) {
        const uint32_t b = r0 + wv + kWaves * i;
        pn[i] = *reinterpret_cast<const uint16_t*>(W.pnorm + uint64_t(b) * kBlock + 2u * lane);
      }
    }
    BlkDir next[kJoinPerWave];
    records(r0 + kWaves * kJoinPerWave, next);   // (past the end: zeros, nothing is read)
    uint32_t d0[kJoinPerWave], d1[kJoinPerWave], f0[kJoinPerWave], f1[kJoinPerWave];
#pragma unroll
    for (uint32_t i = 0; i < kJoinPerWave; ++i) {
      d0[i] = d1[i] = kDocMin;
      f0[i] = f1[i] = 0;
      if (packed[i]) {
        const uint32_t dbits = dir[i].bits & 0xFFu, fbits = dir[i].bits >> 8;
        uint32_t x0, x1;
        extract_fast<LAYOUT>(rd[i].a, rd[i].b, dbits, lane, x0, x1);
        d1[i] = dir[i].prev_last + wave::inclusive_scan(x0 + x1);
        d0[i] = d1[i] - x1;
        extract_fast<LAYOUT>(rf[i].a, rf[i].b, fbits, lane, f0[i], f1[i]);
      } else if (plain[i]) {
        const uint32_t dbits = dir[i].bits & 0xFFu, fbits = dir[i].bits >> 8;
        uint32_t x0, x1;
        raw_extract<LAYOUT>(rd[i], dbits, lane, x0, x1);
        d1[i] = dir[i].prev_last + wave::inclusive_scan(x0 + x1);
        d0[i] = d1[i] - x1;
        if (fbits == 0u) {   // all-equal frequencies: the vint behind the header byte
          uint32_t len;
          f0[i] = f1[i] = vint_from(rf[i].a, &len);
        } else {
          raw_extract<LAYOUT>(rf[i], fbits, lane, f0[i], f1[i]);
        }
      } else if (full[i]) {
        decode_block<LAYOUT, true>(doc + dir[i].off, 0u, dir[i].bits >> 8, dir[i].prev_last, lane,
                                   d0[i], d1[i], f0[i], f1[i]);
      }
    }
    // the two postings' norm bytes: posting order, one 2-byte load per lane (requested with
    // the payloads: they do not depend on the decoded doc ids)
    uint32_t n0[kJoinPerWave], n1[kJoinPerWave];
#pragma unroll
    for (uint32_t i = 0; i < kJoinPerWave; ++i) {
      n0[i] = pn[i] & 0xFFu;
      n1[i] = pn[i] >> 8;
    }
#pragma unroll
    for (uint32_t i = 0; i < kJoinPerWave; ++i) {
      const uint32_t b = r0 + wv + kWaves * i;
      if (full[i])
        join_emit(ent, bnd, b, lane, d0[i], d1[i], f0[i], f1[i], n0[i], n1[i], true, true,
                  b ? dir[i].prev_last : 0u, dead);
    }
#pragma unroll
    for (uint32_t i = 0; i < kJoinPerWave; ++i) dir[i] = next[i];
  }
  // the vint tail / single doc, decoded when the segment was opened: the list's last "block"
  const uint32_t bt = W.nblk;
  if (W.tail_n && bt >= W.first && bt < end && ((bt - W.first) % kWaves) == wv) {
    const uint32_t* tdocs = reinterpret_cast<const uint32_t*>(W.tail_docs);
    const uint32_t* tfreqs = reinterpret_cast<const uint32_t*>(W.tail_freqs);
    const uint32_t i0 = 2u * lane;
    const bool v0 = i0 < W.tail_n, v1 = i0 + 1u < W.tail_n;
    uint32_t td0 = kDocMin, td1 = kDocMin, tf0 = 0, tf1 = 0;
    if (v0) { td0 = tdocs[i0]; tf0 = tfreqs[i0]; }
    if (v1) { td1 = tdocs[i0 + 1u]; tf1 = tfreqs[i0 + 1u]; }
    const uint8_t* tnorms = reinterpret_cast<const uint8_t*>(W.tail_norms);
    const uint32_t tn0 = (v0 && tiny) ? tnorms[i0] : 0u;
    const uint32_t tn1 = (v1 && tiny) ? tnorms[i0 + 1u] : 0u;
    join_emit(ent, bnd, bt, lane, td0, td1, tf0, tf1, tn0, tn1, v0, v1, W.tail_base, dead);
  }
  // behind the list's last posting every remaining tile is empty: whoever holds the last block
  if (nb && nb - 1u >= W.first && nb - 1u < end && ((nb - 1u - W.first) % kWaves) == wv) {
    const uint32_t last_tile = (W.last_doc - kDocMin) / kJoinTile;
    for (uint32_t u = last_tile + 1u + lane; u <= W.n_tiles; u += 64u) bnd[u] = W.n;
  }
}

// The norm byte of every posting, in posting order (1-byte Norm2 columns): built once per segment,
// on its first joined batch — k_join then reads a block's 128 norm bytes with one coalesced
// 2-byte load per lane instead of gathering them by doc id for every batch (the gathers were
// 0.4 of its 1.3 ms).  128 bytes per directory row (= full block); work split by row, as
// k_pack_payloads.
template<int LAYOUT>
__global__ void __launch_bounds__(kThreads)
k_posting_norms(DevSegment seg, uint64_t rows, uint8_t* pnorm) {
  const unsigned lane = threadIdx.x & 63u;
  const uint8_t* norms = seg.norms - seg.norm_min_doc;   // (indexed by doc id)
  IRS_FOR_ROWS(e, rows) {
    const uint32_t base = seg.blk_dir[e].prev_last;
    uint32_t d0, d1, f0, f1;
    decode_block<LAYOUT, false>(row_block(seg, e), seg.blk_bits[e] & 0xFFu,
                                0, base, lane, d0, d1, f0, f1);
    const uint16_t both = uint16_t(uint32_t(norms[d0]) | (uint32_t(norms[d1]) << 8));
    *reinterpret_cast<uint16_t*>(pnorm + e * kBlock + 2u * lane) = both;
  }
}
// ... and of the decoded tails / single docs: one thread per entry of tail_docs
__global__ void __launch_bounds__(kThreads)
k_tail_norms(DevSegment seg, uint64_t n, uint8_t* tail_norms) {
  const uint64_t i = uint64_t(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  const uint32_t d = seg.tail_docs[i];
  // (rows of terms without a tail hold no doc)
  tail_norms[i] = (d >= kDocMin && d <= seg.num_docs) ? (seg.norms - seg.norm_min_doc)[d] : 0u;
}

// ----------------------------------------------------------------- score --

// LDS layout of k_join_pilot / k_join_score (byte offsets; the hot path addresses them
// absolutely, wave::lds_*).  The dummy words sit right behind the accumulators so that their
// offsets fit the 16 address bits of an entry: a lane without a posting adds to its own dummy.
struct JoinOff {
  static constexpr uint32_t acc = 0;                                   // [kJoinTile] u32
  static constexpr uint32_t dummy = 4u * kJoinTile;                    // [64] u32
  static constexpr uint32_t qts = dummy + 256u;                        // DevQTerm[kMaxTerms]
  static constexpr uint32_t jts = qts + uint32_t(sizeof(DevQTerm)) * kMaxTerms;   // JoinTerm[kMaxTerms]
  static constexpr uint32_t rng = jts + uint32_t(sizeof(JoinTerm)) * kMaxTerms;   // [chunk tiles + 1][kMaxTerms] u32
  static constexpr uint32_t cum = rng + 4u * (kJoinChunkTiles + 1u) * kMaxTerms;  // [chunk tiles][kMaxTerms] u32
  static constexpr uint32_t sig = cum + 4u * kJoinChunkTiles * kMaxTerms;         // [2][16] u32
  static constexpr uint32_t cand = sig + 4u * 2u * 16u;                // [2][kJoinCands] u64
  static constexpr uint32_t vars = cand + 8u * 2u * kJoinCands;        // [16] u32
  static constexpr uint32_t caches = vars + 64u;                       // [kTableRows][256] f32
  static constexpr uint32_t end = caches + 4u * 256u * kTableRows;
};
static_assert(JoinOff::cand % 8u == 0u, "candidate keys are 8-byte aligned");
static_assert(JoinOff::caches % 16u == 0u && JoinOff::caches <= 65535u, "the table base is a DS immediate");
static_assert(JoinOff::dummy + 256u <= 65536u, "dummy offsets fit an entry's 16 address bits");

struct alignas(16) JoinQuad {   // 16 bytes moved at once
  uint32_t x, y, z, w;
};
struct JoinSm {   // what build_tables() wants to see
  DevQTerm* qts;
  float* caches;
};

// The per-lane view of a query: lane j holds term j (zeros beyond the query's terms).
struct JoinLane {
  uint32_t ent_lo, ent_hi;
  float cs;
  uint32_t mode;
};

enum : int { kJTable = 0, kJRcp = 1, kJSqrt = 2 };
__device__ __forceinline__ int join_form(uint32_t mode) {
  return !(mode & kJoinGeneral) ? kJTable : ((mode & kJoinSqrt) ? kJSqrt : kJRcp);
}
// the entry a lane without a posting carries: its own dummy accumulator, table offset 0
__device__ __forceinline__ uint32_t join_dummy(unsigned lane) {
  return (JoinOff::dummy + 4u * lane) << 16;
}

// N entries per lane (N slabs of 64): table reads back to back, then the multiply-adds, then
// the LDS adds.  FORM kJTable: every frequency of the term has a table row, score =
// cs * T_tf[norm] — the entry's low 16 bits ARE the offset of T_tf[norm] inside the slot; else
// row 0 and the general expression (score.h tile_post: v_rcp / v_sqrt form).
//
// COUNT (conjunctions, min-match): the low kJoinCountBits bits of an accumulator count the
// terms that hold the doc — every contribution is rounded to a multiple of 16 units and carries
// a 1 there (one v_and_or_b32 more per slab: no second LDS access, no counter array); the unit's
// eligibility bounds the rounding error (irs_hip.hip unit_joinable).
//
// HALF (paired tiles, k_join_score<kJKHalf>): the accumulator word of a doc offset is shared by the
// two tiles of a pair — 16 bits each; a contribution is the 16-bit image of the 32-bit one
// (cs arrives scaled by 2^-15), rounded UP by at least one unit (+ 2.0 before the truncation) and
// shifted into its tile's half (`sh`: 0 or 16, wave-uniform): the sums only pick the docs whose
// exact 32-bit sum k_join_rescore then works out.
template<int FORM, int N, bool COUNT, bool HALF = false>
__device__ __forceinline__ void join_post(const unsigned char* lds, const uint32_t (&e)[4],
                                          float cs, uint32_t tabofs, uint32_t sh = 0u) {
  float t[N];
  uint32_t fx[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const uint32_t at = (FORM == kJTable) ? ((e[k] & 0xFFFFu) | tabofs) : ((e[k] & 0x3FCu) | tabofs);
    t[k] = kAblNoTab ? __uint_as_float(at | 0x3F000000u) : wave::lds_f32(lds, JoinOff::caches + at);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) wave::keep_f(t[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (FORM == kJTable) {
      fx[k] = static_cast<uint32_t>(wave::fma(cs, t[k], HALF ? 2.f : (COUNT ? kJoinCountRound : 1.f)));
      if (COUNT) fx[k] = (fx[k] & ~kJoinCountMask) | 1u;
      if (HALF) fx[k] <<= sh;
    } else {
      const float tf = static_cast<float>(join_tf(e[k]));
      float scaled = (FORM == kJSqrt) ? wave::fast_sqrt(tf) * cs * t[k]
                                      : wave::fma(-cs, wave::fast_rcp(wave::fma(tf, t[k], 1.f)), cs);
      wave::keep_f(scaled);
      fx[k] = COUNT ? ((static_cast<uint32_t>(scaled + kJoinCountRound) & ~kJoinCountMask) | 1u)
                    : (HALF ? (static_cast<uint32_t>(scaled + 2.f) << sh)
                            : (static_cast<uint32_t>(scaled) | 1u));
    }
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (kAblNoAdd) wave::keep(fx[k]);
    else wave::lds_add(lds, JoinOff::acc + (e[k] >> 16), fx[k]);
  }
}
// `slabs` (1..4, wave-uniform) of them.  (The compiler merges the tails of the four copies into
// one that indexes e[] dynamically — a 16-byte scratch array per group.  Round 6 wrote the stages
// as fall-through switches instead, every slab naming its own registers: no scratch access left on
// this path, 490 more branches, k_join_score 5.68 -> 6.62 ms; and slab after slab with early
// exits / always four slabs with dummies: profiles/r06_join_models.txt.  This form stays.)
template<int FORM, bool COUNT, bool HALF = false>
__device__ __forceinline__ void join_post_n(const unsigned char* lds, const uint32_t (&e)[4],
                                            uint32_t slabs, float cs, uint32_t tabofs, uint32_t sh = 0u) {
  if (slabs >= 4u) join_post<FORM, 4, COUNT, HALF>(lds, e, cs, tabofs, sh);
  else if (slabs == 3u) join_post<FORM, 3, COUNT, HALF>(lds, e, cs, tabofs, sh);
  else if (slabs == 2u) join_post<FORM, 2, COUNT, HALF>(lds, e, cs, tabofs, sh);
  else join_post<FORM, 1, COUNT, HALF>(lds, e, cs, tabofs, sh);
}
// M: kJSimple | kJCount | kJHalf (template mode bits of the tile loop).  kJHalf: paired tiles —
// lane j < 2 * kMaxTerms is term j % kMaxTerms in tile j / kMaxTerms of the pair; a lane's mode
// word also says which half its tile's sums live in (kJoinHiHalf; JoinLane of join_lane_half)
enum : int { kJSimple = 1, kJCount = 2, kJHalf = 4 };
template<int M>
__device__ __forceinline__ void join_post_any(const unsigned char* lds, const uint32_t (&e)[4],
                                              uint32_t slabs, float cs, uint32_t mode) {
  constexpr bool COUNT = (M & kJCount) != 0;
  if (M & kJHalf) {
    const uint32_t sh = join_half_shift(mode);
    if (M & kJSimple) {
      join_post_n<kJTable, false, true>(lds, e, slabs, cs, 0u, sh);
    } else {
      const uint32_t tabofs = mode & kJoinTabMask;
      const int form = join_form(mode);   // (wave-uniform)
      if (form == kJTable) join_post_n<kJTable, false, true>(lds, e, slabs, cs, tabofs, sh);
      else if (form == kJRcp) join_post_n<kJRcp, false, true>(lds, e, slabs, cs, tabofs, sh);
      else join_post_n<kJSqrt, false, true>(lds, e, slabs, cs, tabofs, sh);
    }
  } else if (M & kJSimple) {
    join_post_n<kJTable, COUNT>(lds, e, slabs, cs, 0u);
  } else {
    const uint32_t tabofs = mode & kJoinTabMask;
    const int form = join_form(mode);   // (wave-uniform)
    if (form == kJTable) join_post_n<kJTable, COUNT>(lds, e, slabs, cs, tabofs);
    else if (form == kJRcp) join_post_n<kJRcp, COUNT>(lds, e, slabs, cs, tabofs);
    else join_post_n<kJSqrt, COUNT>(lds, e, slabs, cs, tabofs);
  }
}

// The first ceil(count / 64) <= 4 slabs of `count` (> 0) entries at `base`: a dword per lane and
// slab (coalesced); lanes past the end keep their dummy entry, so whatever consumes the values
// needs no mask; slabs past the end are not loaded at all.
__device__ __forceinline__ void join_load(uint64_t base, uint32_t count, unsigned lane,
                                          uint32_t (&e)[4]) {
  const uint32_t off = lane * 4u;
  e[0] = e[1] = e[2] = e[3] = join_dummy(lane);
  if (lane < count) e[0] = wave::gload_u32(base, off);
  if (count > 64u) {   // (wave-uniform)
    if (lane + 64u < count) e[1] = wave::gload_u32(base, off + 256u);
    if (count > 128u) {
      if (lane + 128u < count) e[2] = wave::gload_u32(base, off + 512u);
      if (count > 192u && lane + 192u < count) e[3] = wave::gload_u32(base, off + 768u);
    }
  }
}

// `count` consecutive entries from address `base` (wave-uniform), 256 per step.
template<int FORM, bool COUNT, bool HALF = false>
__device__ __forceinline__ void join_run(const unsigned char* lds, uint64_t base, uint32_t count,
                                         float cs, uint32_t tabofs, unsigned lane, uint32_t sh = 0u) {
  const uint32_t off = lane * 4u;
  if (HALF) {
    // paired tiles: a share is several groups long — group g + 1 is requested before group g is
    // accumulated (the wait for g leaves g + 1 in flight)
    if (count >= 256u) {
      uint32_t cur[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) cur[k] = wave::gload_u32(base, off + 256u * uint32_t(k));
      base += 1024u;
      count -= 256u;
      while (count >= 256u) {
        uint32_t nxt[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) nxt[k] = wave::gload_u32(base, off + 256u * uint32_t(k));
        join_post<FORM, 4, COUNT, HALF>(lds, cur, cs, tabofs, sh);
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
        base += 1024u;
        count -= 256u;
      }
      join_post<FORM, 4, COUNT, HALF>(lds, cur, cs, tabofs, sh);
    }
  } else {
    while (count >= 256u) {
      uint32_t e[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) e[k] = wave::gload_u32(base, off + 256u * uint32_t(k));
      wave::keep_all(e);
      join_post<FORM, 4, COUNT, HALF>(lds, e, cs, tabofs, sh);
      base += 1024u;
      count -= 256u;
    }
  }
  if (count) {
    uint32_t e[4];
    join_load(base, count, lane, e);
    join_post_n<FORM, COUNT, HALF>(lds, e, (count + 63u) >> 6, cs, tabofs, sh);
  }
}

// A wavefront's share of one tile.  The entries of the query's terms in the tile, term after
// term, are one sequence of N entries; wavefront w of nw takes [N*w/nw, N*(w+1)/nw) — a
// contiguous piece that touches one or two terms, rarely more.  begin() intersects that piece
// with every term's range (lane j = term j) and REQUESTS the first kJoinPre * 64 entries of the
// first term it touches — the loads then fly behind the previous tile's barrier and epilogue;
// finish() consumes them and streams whatever is left.  SIMPLE: every term of the query scores
// through table slot 0 (one scorer, small frequencies — the usual batch): no per-term form.
constexpr uint32_t kJoinPre = 256;   // entries begin() requests: four loads, always
struct JoinRun {
  uint32_t a_lo, a_hi;   // lane j = term j: address of the first entry this wavefront takes
  uint32_t cnt;          //   and how many (0: none)
  uint64_t mask;         // (uniform) terms with cnt > 0 that are still to do
  uint64_t rest;         // (uniform) what begin() left of its term: address,
  uint32_t left;         //   entries
  uint32_t pre;          // (uniform) entries requested by begin(): 0 .. kJoinPre
  float cs;              // (uniform) of that term
  uint32_t mode;
  uint32_t e[4];         // the requested slabs, RAW: lanes past `pre` hold some other entry
};

// a = first entry of term j in the tile (index into its stream), n = its entries there,
// c = inclusive prefix sum of n over the terms (lanes beyond the query's terms: n = 0).
// Always issues exactly four loads (to `safe`, any readable address, when the share is empty;
// lanes past the end re-read the share's last entry): two runs are in flight at a time — the
// next tile's and the one after — and only a FIXED number of younger loads lets the wait for
// the older run's data leave the younger one's in flight (s_waitcnt vmcnt(4)).
template<int M>
__device__ __forceinline__ void join_begin(JoinRun& r, const JoinLane& T, uint32_t a, uint32_t n,
                                           uint32_t c, uint32_t wv, uint32_t nw_log2,
                                           uint64_t safe, unsigned lane) {
  const uint32_t N = wave::read_lane(c, ((M & kJHalf) ? 2u * kMaxTerms : kMaxTerms) - 1u);
  const uint32_t lo = (N * wv) >> nw_log2, hi = (N * (wv + 1u)) >> nw_log2;
  const uint32_t first = c - n;                      // the term's position in the sequence
  const uint32_t st = lo > first ? lo : first;
  const uint32_t en = hi < c ? hi : c;
  r.cnt = en > st ? en - st : 0u;
  const uint64_t addr = ((uint64_t(T.ent_hi) << 32) | T.ent_lo) + 4ull * (uint64_t(a) + (st - first));
  r.a_lo = uint32_t(addr);
  r.a_hi = uint32_t(addr >> 32);
  uint64_t mask = wave::ballot(r.cnt != 0u);
  uint64_t base = safe;
  uint32_t take = 0;
  r.left = 0;
  r.rest = 0;
  r.cs = 0.f;
  r.mode = 0;
  if (mask) {
    const uint32_t j = uint32_t(__builtin_ctzll(mask));
    mask &= mask - 1ull;
    base = (uint64_t(wave::read_lane(r.a_hi, j)) << 32) | wave::read_lane(r.a_lo, j);
    const uint32_t cnt = wave::read_lane(r.cnt, j);
    r.cs = wave::read_lane_f(T.cs, j);
    // (paired tiles, one table: the mode word is just the half — the term's index says which)
    if ((M & kJSimple) && (M & kJHalf)) r.mode = j >= kMaxTerms ? uint32_t(kJoinHiHalf) : 0u;
    else if (!(M & kJSimple)) r.mode = wave::read_lane(T.mode, j);
    take = cnt < kJoinPre ? cnt : kJoinPre;
    r.left = cnt - take;
    r.rest = base + 4ull * take;
  }
  r.pre = take;
  r.mask = mask;
  const uint32_t last = take ? (take - 1u) * 4u : 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t off = lane * 4u + 256u * uint32_t(k);
    r.e[k] = wave::gload_u32(base, off < last ? off : last);
  }
}

template<int M>
__device__ __forceinline__ void join_some(const unsigned char* lds, uint64_t base, uint32_t cnt,
                                          float cs, uint32_t mode, unsigned lane) {
  constexpr bool COUNT = (M & kJCount) != 0;
  if (M & kJHalf) {
    const uint32_t sh = join_half_shift(mode);
    if (M & kJSimple) {
      join_run<kJTable, false, true>(lds, base, cnt, cs, 0u, lane, sh);
    } else {
      const uint32_t tabofs = mode & kJoinTabMask;
      const int form = join_form(mode);
      if (form == kJTable) join_run<kJTable, false, true>(lds, base, cnt, cs, tabofs, lane, sh);
      else if (form == kJRcp) join_run<kJRcp, false, true>(lds, base, cnt, cs, tabofs, lane, sh);
      else join_run<kJSqrt, false, true>(lds, base, cnt, cs, tabofs, lane, sh);
    }
  } else if (M & kJSimple) {
    join_run<kJTable, COUNT>(lds, base, cnt, cs, 0u, lane);
  } else {
    const uint32_t tabofs = mode & kJoinTabMask;
    const int form = join_form(mode);
    if (form == kJTable) join_run<kJTable, COUNT>(lds, base, cnt, cs, tabofs, lane);
    else if (form == kJRcp) join_run<kJRcp, COUNT>(lds, base, cnt, cs, tabofs, lane);
    else join_run<kJSqrt, COUNT>(lds, base, cnt, cs, tabofs, lane);
  }
}

template<int M>
__device__ __forceinline__ void join_finish(const unsigned char* lds, JoinRun& r, const JoinLane& T,
                                            unsigned lane) {
  // (the run's scalars crossed a barrier and a loop back edge inside a struct: the compiler no
  // longer knows they are wave-uniform and would predicate everything below lane by lane)
  const uint32_t pre = wave::uniform(r.pre);
  if (!pre) {   // (nothing requested: nothing at all)
    if (kAblWait) wave::keep_all(r.e);
    return;
  }
  constexpr bool SIMPLE = (M & kJSimple) != 0 && (M & kJHalf) == 0;   // (no mode word needed)
  const uint32_t mode0 = SIMPLE ? 0u : wave::uniform(r.mode);
  const float cs0 = wave::uniform_f(r.cs);
  {
    const uint32_t dummy = join_dummy(lane);
    uint32_t e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = lane + 64u * uint32_t(k) < pre ? r.e[k] : dummy;
    join_post_any<M>(lds, e, (pre + 63u) >> 6, cs0, mode0);
  }
  const uint32_t left = wave::uniform(r.left);
  if (left) join_some<M>(lds, wave::uniform64(r.rest), left, cs0, mode0, lane);
  uint64_t mask = wave::uniform64(r.mask);
  while (mask) {
    const uint32_t j = uint32_t(__builtin_ctzll(mask));
    mask &= mask - 1ull;
    const uint64_t base = (uint64_t(wave::read_lane(r.a_hi, j)) << 32) | wave::read_lane(r.a_lo, j);
    const uint32_t mode = SIMPLE ? 0u
                          : (((M & kJSimple) && (M & kJHalf)) ? (j >= kMaxTerms ? uint32_t(kJoinHiHalf) : 0u)
                                                              : wave::read_lane(T.mode, j));
    join_some<M>(lds, base, wave::read_lane(r.cnt, j), wave::read_lane_f(T.cs, j), mode, lane);
  }
}

// Chunk / pilot prologue, whole workgroup: the query's term scorers and stream records to
// LDS; the score tables are rebuilt only when the scorer parameters differ from what the
// previous query of this workgroup left there (the queries of a batch normally share them).
// Ends with every thread seeing all of it.
__device__ __forceinline__ void join_prologue(unsigned char* smem, const DevQuery& qd,
                                              const DevQTerm* qterms, const JoinTerm* jterms) {
  DevQTerm* qts = reinterpret_cast<DevQTerm*>(smem + JoinOff::qts);
  JoinQuad* jts = reinterpret_cast<JoinQuad*>(smem + JoinOff::jts);
  uint32_t* sig = reinterpret_cast<uint32_t*>(smem + JoinOff::sig);   // [0]: in force, [16]: wanted
  const uint32_t tid = threadIdx.x;
  if (tid < qd.n_terms) {
    const DevQTerm qt = qterms[qd.first_term + tid];
    qts[tid] = qt;
    if (qt.cache_id < kMaxCaches) {   // (same values from every term of the slot)
      sig[16u + 1u + 3u * qt.cache_id] = uint32_t(qt.kind);
      sig[16u + 2u + 3u * qt.cache_id] = __float_as_uint(qt.norm_const);
      sig[16u + 3u + 3u * qt.cache_id] = __float_as_uint(qt.norm_length);
    }
  }
  if (tid < kJoinTermQuads * kMaxTerms) {   // a JoinTerm = two 16-byte halves
    const uint32_t j = tid / kJoinTermQuads;
    uint32_t x = 0, y = 0, z = 0, w = 0;   // (field by field: an aggregate temporary would live in scratch)
    if (j < qd.n_terms) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(jterms + qd.first_term) + 4u * tid;
      x = src[0]; y = src[1]; z = src[2]; w = src[3];
    }
    jts[tid].x = x; jts[tid].y = y; jts[tid].z = z; jts[tid].w = w;
  }
  if (tid == 0) sig[16u] = qd.n_caches;
  __syncthreads();
  bool same = sig[0] == sig[16u];
  for (uint32_t c = 0; c < 3u * qd.n_caches; ++c) same = same && sig[1u + c] == sig[17u + c];
  if (!same) {   // (the same for every thread)
    JoinSm sm;
    sm.qts = qts;
    sm.caches = reinterpret_cast<float*>(smem + JoinOff::caches);
    build_tables(sm, qd.n_caches, qd.n_terms);
    __syncthreads();   // everyone has compared before the signature changes
    if (tid < 1u + 3u * kMaxCaches) sig[tid] = sig[16u + tid];
    __syncthreads();
  }
}

__device__ __forceinline__ JoinLane join_lane(const unsigned char* smem, unsigned lane) {
  JoinLane T{};
  if (lane < kMaxTerms) {
    const JoinQuad lo = reinterpret_cast<const JoinQuad*>(smem + JoinOff::jts)[kJoinTermQuads * lane];
    const JoinQuad hi = reinterpret_cast<const JoinQuad*>(smem + JoinOff::jts)[kJoinTermQuads * lane + 1u];
    T.ent_lo = lo.x;   // JoinTerm::entries
    T.ent_hi = lo.y;
    T.cs = __uint_as_float(hi.x);
    T.mode = hi.y;
  }
  return T;
}

// ... of a pair of tiles (kJHalf): lane j < 2 * kMaxTerms = term j % kMaxTerms in the pair's tile
// j / kMaxTerms; cs in 16-bit units (2^-15: exact), mode = the half's shift
static_assert(kMaxTerms == 16u, "join_lane_half / join_pairs: lane >> 4 is the tile of the pair");
__device__ __forceinline__ JoinLane join_lane_half(const unsigned char* smem, unsigned lane) {
  JoinLane T{};
  if (lane < 2u * kMaxTerms) {
    const uint32_t j = lane & (kMaxTerms - 1u);
    const JoinQuad lo = reinterpret_cast<const JoinQuad*>(smem + JoinOff::jts)[kJoinTermQuads * j];
    const JoinQuad hi = reinterpret_cast<const JoinQuad*>(smem + JoinOff::jts)[kJoinTermQuads * j + 1u];
    T.ent_lo = lo.x;
    T.ent_hi = lo.y;
    T.cs = __uint_as_float(hi.x) * (1.f / 32768.f);
    T.mode = hi.y | (lane >= kMaxTerms ? uint32_t(kJoinHiHalf) : 0u);
  }
  return T;
}

struct JoinArgs {
  const DevQuery* queries;
  const DevQTerm* qterms;
  const JoinTerm* jterms;
  const uint32_t* bstar;
  uint64_t* cands;
  uint32_t* cand_count;
  unsigned long long* hits;
  const uint32_t* order;  // work-queue order: slot i names the unit that runs i-th in a chunk round
  // One work queue per XCD: the units are dealt to kJoinQueues groups of alike units (sorted by
  // their heaviest term), a workgroup pulls from the queue of the XCD it runs on (blockIdx % 8:
  // how the hardware deals workgroups today — a placement assumption that only speed depends on)
  // and moves on to the other queues when that one is empty.  The workgroups that share an L2
  // then read the same streams at the same time.  Chunk ids: queue g owns [base[g], base[g+1]),
  // chunk-major over its units order[first[g] .. first[g+1]); work_counter[g] starts at base[g].
  uint32_t* work_counter;
  uint32_t base[kJoinQueues + 1];
  uint32_t first[kJoinQueues + 1];
  uint32_t cpq;          // chunk ids per unit
  uint32_t n_units;
  uint32_t nw_log2;
  uint32_t cand_cap;
  uint32_t chunk_tiles;  // tiles per chunk, <= kJoinChunkTiles: the units' tiles cut evenly
};

// One workgroup per unit scores the tiles {phase, phase + P, ...} and picks the threshold bin
// (score.h k_pilot: same histogram, same rule).
__global__ void __launch_bounds__(kTileThreadsMax)
k_join_pilot(const uint32_t* units, const DevQuery* queries, const DevQTerm* qterms,
             const JoinTerm* jterms, uint32_t stride, uint32_t nw_log2, uint32_t* bstar,
             uint32_t margin, const uint32_t* min_bin,
             const uint32_t* group_of /*[unit] group + 1, 0: a threshold of its own; null: all*/,
             uint32_t* group_hist /*[group][kBins + 2]: bins, sampled tiles, tiles*/) {
  RT_DYN_SMEM(smem);
  if (!wave::lds_is_at_zero(smem)) __builtin_trap();
  uint32_t* acc = reinterpret_cast<uint32_t*>(smem + JoinOff::acc);
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem + JoinOff::end);   // [kBins]
  uint32_t* sig = reinterpret_cast<uint32_t*>(smem + JoinOff::sig);
  const uint32_t tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const uint32_t wv = wave::uniform(tid >> 6);
  const uint32_t q = units[blockIdx.x];
  const DevQuery qd = queries[q];
  const uint32_t n_tiles = qd.n_tiles;
  const uint32_t need_matches = query_need(qd.op);   // (> 1: conjunction / min-match)
  for (uint32_t i = tid; i < kBins; i += blockDim.x) hist[i] = 0u;
  for (uint32_t i = tid; i < kJoinTile; i += blockDim.x) acc[i] = 0u;
  if (tid == 0) sig[0] = 0xFFFFFFFFu;
  __syncthreads();
  join_prologue(smem, qd, qterms, jterms);
  const JoinLane T = join_lane(smem, lane);
  // The sampled tiles in passes of kJoinChunkTiles: their boundaries first (one batch of loads
  // into LDS: first entry and entry count per term), then tile after tile with the NEXT tile's
  // first entries requested before this tile's barrier — they fly behind its epilogue, which
  // reads and clears the accumulators with LDS exchanges (as k_join_score's).
  uint32_t* row_a = reinterpret_cast<uint32_t*>(smem + JoinOff::rng);   // [pass tile][kMaxTerms]
  uint32_t* row_n = reinterpret_cast<uint32_t*>(smem + JoinOff::cum);
  const uint32_t first_tile = (q * 7u) % stride;
  const uint32_t sampled_tiles = first_tile < n_tiles ? (n_tiles - first_tile + stride - 1) / stride : 0u;
  const uint64_t safe = reinterpret_cast<uint64_t>(jterms);
  auto histogram = [&](uint32_t f) {
    if (!f) return;
    if (need_matches > 1u) {
      if ((f & kJoinCountMask) >= need_matches)
        atomicAdd(&hist[score_bin(from_fixed<uint32_t>(f & ~kJoinCountMask, qd.fx_inv), qd.bin_scale)], 1u);
    } else {
      const float v = f <= kMaxTerms ? 0.f : from_fixed<uint32_t>(f, qd.fx_inv);
      atomicAdd(&hist[score_bin(v, qd.bin_scale)], 1u);
    }
  };
  for (uint32_t s0 = 0; s0 < sampled_tiles; s0 += kJoinChunkTiles) {
    const uint32_t ns = sampled_tiles - s0 < kJoinChunkTiles ? sampled_tiles - s0 : kJoinChunkTiles;
    __syncthreads();   // (the previous pass is through with the rows)
    for (uint32_t e = tid; e < ns * kMaxTerms; e += blockDim.x) {
      const uint32_t i = e / kMaxTerms, j = e % kMaxTerms;
      uint32_t a = 0, n = 0;
      if (j < qd.n_terms) {
        const uint32_t* bnd = reinterpret_cast<const uint32_t*>(
            reinterpret_cast<const JoinTerm*>(smem + JoinOff::jts)[j].bounds);
        const uint32_t tile = first_tile + (s0 + i) * stride;
        a = bnd[tile];
        n = bnd[tile + 1u] - a;
      }
      row_a[e] = a;
      row_n[e] = n;
    }
    __syncthreads();
    auto begin = [&](uint32_t i, JoinRun& r) {   // (i >= ns: an empty share)
      uint32_t a = 0, n = 0;
      if (lane < kMaxTerms && i < ns) {
        a = row_a[i * kMaxTerms + lane];
        n = row_n[i * kMaxTerms + lane];
      }
      if (need_matches > 1u)
        join_begin<kJCount>(r, T, a, n, wave::inclusive_scan(n), wv, nw_log2, safe, lane);
      else
        join_begin<0>(r, T, a, n, wave::inclusive_scan(n), wv, nw_log2, safe, lane);
    };
    auto finish = [&](JoinRun& r) {
      if (need_matches > 1u) join_finish<kJCount>(smem, r, T, lane);
      else join_finish<0>(smem, r, T, lane);
    };
    auto end_tile = [&]() {
      __syncthreads();
      for (uint32_t i = tid * 4u; i < kJoinTile; i += blockDim.x * 4u) {
        uint32_t v[4];
        wave::lds_take4(smem, JoinOff::acc + i * 4u, v);
        if (v[0] | v[1] | v[2] | v[3]) {
          histogram(v[0]);
          histogram(v[1]);
          histogram(v[2]);
          histogram(v[3]);
        }
      }
      __syncthreads();
    };
    JoinRun r0, r1;
    begin(0, r0);
    for (uint32_t i = 0; i < ns; i += 2u) {
      finish(r0);
      begin(i + 1u, r1);
      end_tile();
      if (i + 1u < ns) {
        finish(r1);
        begin(i + 2u, r0);
        end_tile();
      }
    }
  }
  if (group_of && group_of[q]) {
    // one threshold for the units of a group (the same query on several segments,
    // k_group_threshold): this unit only contributes its sample
    uint32_t* gh = group_hist + uint64_t(group_of[q] - 1u) * (kBins + 2u);
    for (uint32_t i = tid; i < kBins; i += blockDim.x)
      if (hist[i]) atomicAdd(&gh[i], hist[i]);
    if (tid == 0) {
      const uint32_t phase = (q * 7u) % stride;
      atomicAdd(&gh[kBins], phase < n_tiles ? (n_tiles - phase + stride - 1) / stride : 0u);
      atomicAdd(&gh[kBins + 1u], n_tiles);
    }
    return;
  }
  uint32_t need = qd.k;
  if (margin) {
    const uint32_t phase = (q * 7u) % stride;
    const uint32_t sampled = phase < n_tiles ? (n_tiles - phase + stride - 1) / stride : 0u;
    const uint64_t est = (uint64_t(margin) * qd.k * sampled + n_tiles - 1) / (n_tiles ? n_tiles : 1u);
    const uint32_t lo = est < kPilotMinSample ? kPilotMinSample : uint32_t(est < 0xFFFFFFFFull ? est : 0xFFFFFFFFull);
    need = lo < qd.k ? lo : qd.k;
  }
  if (tid < 64) {   // suffix search: lane L owns the 8 bins of chunk 63-L (k_pilot)
    const uint32_t chunk = 63u - lane;
    uint32_t s = 0;
    for (uint32_t i = 0; i < kBins / 64; ++i) s += hist[chunk * (kBins / 64) + i];
    const uint32_t incl = wave::inclusive_scan(s);
    const uint64_t reach = wave::ballot(incl >= need);
    uint32_t result = 0;
    if (reach) {
      const int src = __builtin_ctzll(reach);
      const uint32_t above = wave::bcast(incl - s, src);
      const uint32_t c = 63u - uint32_t(src);
      uint32_t cum = above;
      for (int i = int(kBins / 64) - 1; i >= 0; --i) {
        cum += hist[c * (kBins / 64) + uint32_t(i)];
        if (cum >= need) { result = c * (kBins / 64) + uint32_t(i); break; }
      }
    }
    if (lane == 0) bstar[q] = (min_bin && min_bin[q] > result) ? min_bin[q] : result;
  }
}

// A group = one query on the several segments of a batch whose results the caller merges: ONE
// threshold bin for the group, picked from the summed pilot histograms with the rule of
// k_join_pilot (an expected margin * k candidates in the whole group instead of in every
// segment — the candidates are what a small segment's tiles spend their time on).  One wavefront
// per group; members[g * n_segs + s] = unit or 0xFFFFFFFF.
__global__ void __launch_bounds__(64)
k_group_threshold(const DevQuery* queries, const uint32_t* members, uint32_t n_segs,
                  const uint32_t* group_hist, uint32_t margin, const uint32_t* min_bin,
                  uint32_t* bstar) {
  const unsigned lane = threadIdx.x;
  const uint32_t g = blockIdx.x;
  const uint32_t* hist = group_hist + uint64_t(g) * (kBins + 2u);
  uint32_t first = 0xFFFFFFFFu;
  for (uint32_t s = 0; s < n_segs; ++s) {
    const uint32_t u = members[uint64_t(g) * n_segs + s];
    if (u != 0xFFFFFFFFu && first == 0xFFFFFFFFu) first = u;
  }
  if (first == 0xFFFFFFFFu) return;
  const uint32_t k = queries[first].k;
  const uint32_t sampled = hist[kBins], n_tiles = hist[kBins + 1u];
  uint32_t need = k;
  if (margin) {
    const uint64_t est = (uint64_t(margin) * k * sampled + n_tiles - 1) / (n_tiles ? n_tiles : 1u);
    const uint32_t lo = est < kPilotMinSample ? kPilotMinSample : uint32_t(est < 0xFFFFFFFFull ? est : 0xFFFFFFFFull);
    need = lo < k ? lo : k;
  }
  const uint32_t chunk = 63u - lane;   // suffix search: lane L owns the 8 bins of chunk 63-L
  uint32_t sum = 0;
  for (uint32_t i = 0; i < kBins / 64; ++i) sum += hist[chunk * (kBins / 64) + i];
  const uint32_t incl = wave::inclusive_scan(sum);
  const uint64_t reach = wave::ballot(incl >= need);
  uint32_t result = 0;
  if (reach) {
    const int src = __builtin_ctzll(reach);
    const uint32_t above = wave::bcast(incl - sum, src);
    const uint32_t c = 63u - uint32_t(src);
    uint32_t cum = above;
    for (int i = int(kBins / 64) - 1; i >= 0; --i) {
      cum += hist[c * (kBins / 64) + uint32_t(i)];
      if (cum >= need) { result = c * (kBins / 64) + uint32_t(i); break; }
    }
  }
  if (lane < n_segs) {
    const uint32_t u = members[uint64_t(g) * n_segs + lane];
    if (u != 0xFFFFFFFFu) bstar[u] = (min_bin && min_bin[u] > result) ? min_bin[u] : result;
  }
}

// ... and its soundness check behind k_select: the group's lists together hold min(k, docs that
// matched) entries, or the (estimated) threshold was too high — kStatusUnderflow, the host re-runs
// the batch with the sound threshold (what k_select checks per unit for ungrouped units).  In two
// steps, because the group may span RANKS (irs_hip_batch_set_comm): k_group_sums leaves three
// counters per group — entries listed, docs matched (capped at k per unit: only "more than
// listed" matters), units whose threshold is the caller's own — and the run's status bits so far
// as two more counters; the host layer's all-reduce sums them over the ranks; k_group_verdict
// reads the sums, so EVERY rank reaches the same verdict and takes the same recovery path.
constexpr uint32_t kGroupSumWords = 3;
__global__ void __launch_bounds__(64)
k_group_sums(const DevQuery* queries, const uint32_t* members, uint32_t n_segs, uint32_t n_groups,
             const uint32_t* out_count, const unsigned long long* hits, const uint32_t* bstar,
             const uint32_t* min_bin, const uint32_t* status, uint32_t* sums) {
  const uint32_t g = blockIdx.x * 64u + threadIdx.x;
  if (g == 0) {
    const uint32_t st = *status;
    sums[uint64_t(n_groups) * kGroupSumWords] = (st & kStatusUnderflow) ? 1u : 0u;
    sums[uint64_t(n_groups) * kGroupSumWords + 1u] = (st & kStatusOverflow) ? 1u : 0u;
  }
  if (g >= n_groups) return;
  uint32_t got = 0, matched = 0, callers = 0;
  for (uint32_t s = 0; s < n_segs; ++s) {
    const uint32_t u = members[uint64_t(g) * n_segs + s];
    if (u == 0xFFFFFFFFu) continue;
    const uint32_t k = queries[u].k;
    got += out_count[u];
    matched += hits[u] < k ? uint32_t(hits[u]) : k;
    callers += (min_bin && min_bin[u] != 0u && bstar[u] == min_bin[u]) ? 1u : 0u;
  }
  sums[uint64_t(g) * kGroupSumWords] = got;
  sums[uint64_t(g) * kGroupSumWords + 1u] = matched;
  sums[uint64_t(g) * kGroupSumWords + 2u] = callers;
}
// (queries[g] = the query's unit on the first segment: every unit of it carries the same k)
__global__ void __launch_bounds__(64)
k_group_verdict(const DevQuery* queries, uint32_t n_groups, const uint32_t* sums, uint32_t* status) {
  const uint32_t g = blockIdx.x * 64u + threadIdx.x;
  if (g == 0) {
    uint32_t st = 0;
    if (sums[uint64_t(n_groups) * kGroupSumWords]) st |= kStatusUnderflow;
    if (sums[uint64_t(n_groups) * kGroupSumWords + 1u]) st |= kStatusOverflow;
    if (st) atomicOr(status, st);
  }
  if (g >= n_groups) return;
  const uint32_t got = sums[uint64_t(g) * kGroupSumWords], matched = sums[uint64_t(g) * kGroupSumWords + 1u];
  const bool callers = sums[uint64_t(g) * kGroupSumWords + 2u] != 0u;
  if (got < queries[g].k && matched > got && !callers) atomicOr(status, kStatusUnderflow);
}

// The tiles of one chunk (k_join_score); everything per query / per chunk arrives in `ctx`.
struct JoinTileCtx {
  const JoinArgs* args;
  uint32_t q, bs, thr, cap;
  uint32_t need;       // matches a doc needs (units with match counts)
  float fx_inv, bin_scale;
  uint64_t* lc;        // this chunk's candidate staging buffer
  uint32_t* ncand;     // ... and its fill count
};

template<int M>
__device__ __forceinline__ void join_tiles(unsigned char* smem, const JoinTileCtx& ctx,
                                           const JoinLane& T, uint32_t tile0, uint32_t ntile,
                                           uint32_t wv, uint32_t nw_log2, uint32_t& my_hits) {
  const uint32_t* rng = reinterpret_cast<const uint32_t*>(smem + JoinOff::rng);
  const uint32_t* cum = reinterpret_cast<const uint32_t*>(smem + JoinOff::cum);
  const uint32_t tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const uint64_t safe = reinterpret_cast<uint64_t>(ctx.args->jterms);
  constexpr bool COUNT = (M & kJCount) != 0;
  auto begin = [&](uint32_t u, JoinRun& r) {   // (u >= ntile: an empty share, four loads all the same)
    uint32_t a = 0, n = 0, c = 0;
    if (lane < kMaxTerms && u < ntile) {
      a = rng[u * kMaxTerms + lane];
      n = (rng[(u + 1u) * kMaxTerms + lane] - a) >> kAblShift;
      c = cum[u * kMaxTerms + lane];
    }
    join_begin<M>(r, T, a, n, c, wv, nw_log2, safe, lane);
  };
  // two runs in flight: while tile u is being accumulated, the entries of tiles u+1 AND u+2
  // are on their way (a request then has a whole tile's time to come back from HBM, not just
  // an epilogue's); the loop is unrolled by two so that the two register sets simply alternate
  JoinRun r0, r1;
  begin(0, r0);
  begin(1, r1);
  // barrier B1 + epilogue + barrier B2 of tile u
  auto end_tile = [&](uint32_t u) {
    if (!kAblNoBar) __syncthreads();   // B1: every accumulation of tile u has landed
    const uint32_t doc0 = kDocMin + (tile0 + u) * kJoinTile;
    auto candidate = [&](uint32_t i, uint32_t f) {   // rare
      const float v = COUNT ? from_fixed<uint32_t>(f & ~kJoinCountMask, ctx.fx_inv)
                            : (f <= kMaxTerms ? 0.f : from_fixed<uint32_t>(f, ctx.fx_inv));
      if (score_bin(v, ctx.bin_scale) >= ctx.bs) {
        const uint64_t key = make_key(v, doc0 + i);
        const uint32_t slot = atomicAdd(ctx.ncand, 1u);
        if (slot < kJoinCands) {
          ctx.lc[slot] = key;
        } else {   // rarer: more candidates in one chunk than staging slots
          const uint32_t g = atomicAdd(&ctx.args->cand_count[ctx.q], 1u);
          if (g < ctx.cap) ctx.args->cands[uint64_t(ctx.q) * ctx.cap + g] = key;
        }
      }
    };
    // four accumulators per lane read AND cleared by one LDS exchange; two in flight where the
    // geometry allows
    auto four = [&](uint32_t i, const uint32_t (&w)[4]) {
      uint32_t v[4] = {w[0], w[1], w[2], w[3]};
      if (COUNT) {   // only docs held by enough terms exist
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool m = (v[k] & kJoinCountMask) >= ctx.need;
          my_hits += m ? 1u : 0u;
          v[k] = m ? v[k] : 0u;
        }
      } else if (!kAblCount) {
        wave::count_nonzero4(my_hits, v[0], v[1], v[2], v[3]);
      }
      if (kAblEpi) return;
      uint32_t top = v[0] > v[1] ? v[0] : v[1];
      const uint32_t top2 = v[2] > v[3] ? v[2] : v[3];
      top = top > top2 ? top : top2;
      if (top >= ctx.thr) {   // rare: one copy of the candidate code, per-lane loop
        uint32_t cm = (v[0] >= ctx.thr ? 1u : 0u) | (v[1] >= ctx.thr ? 2u : 0u) |
                      (v[2] >= ctx.thr ? 4u : 0u) | (v[3] >= ctx.thr ? 8u : 0u);
        while (cm) {
          const uint32_t e = uint32_t(__builtin_ctz(cm));
          cm &= cm - 1u;
          const uint32_t x = e == 0u ? v[0] : (e == 1u ? v[1] : (e == 2u ? v[2] : v[3]));
          candidate(i + e, x);
        }
      }
    };
    const uint32_t step = blockDim.x * 4u;
    uint32_t i = kAblEpi >= 2u ? kJoinTile : tid * 4u;
    for (; i + step < kJoinTile; i += 2u * step) {
      uint32_t v0[4], v1[4];
      wave::lds_take4x2(smem, JoinOff::acc + i * 4u, JoinOff::acc + (i + step) * 4u, v0, v1);
      four(i, v0);
      four(i + step, v1);
    }
    if (i < kJoinTile) {
      uint32_t v0[4];
      wave::lds_take4(smem, JoinOff::acc + i * 4u, v0);
      four(i, v0);
    }
    if (!kAblNoBar) __syncthreads();   // B2: accumulators are clear again
  };
  // straight-line pairs of tiles (the compiler's wait-count bookkeeping only sees the fixed
  // distance between a run's loads and its use when no branch separates the two register
  // sets); an odd tile count ends with an empty share, whose begin / finish do nothing
  for (uint32_t u = 0; u < ntile; u += 2u) {
    join_finish<M>(smem, r0, T, lane);
    begin(u + 2u, r0);
    end_tile(u);
    join_finish<M>(smem, r1, T, lane);
    begin(u + 3u, r1);
    if (u + 1u < ntile) end_tile(u + 1u);
  }
}

// PAIRED TILES (k_join_score<kJKHalf>).  The per-(query, tile) visit — two barriers, the share
// bookkeeping, the wait for the first entries, the dense read-and-clear of 48 KB — is what
// k_join_score spends its time on, not the postings (profiles/r05_pruning.txt: 2.27 of 5.7 ms with
// no entries at all, and the first quarter of the entries costs as much as the other three).  A
// visit here covers TWO consecutive tiles: a term's entries of both are contiguous in its stream,
// and the accumulator word of a doc offset holds the first tile's sum in its low 16 bits and the
// second tile's in its high 16 bits — 16-bit images of the 32-bit contributions, each rounded up
// by one to two units (join_post HALF), so that
//     exact 32-bit sum f >= thr  ==>  16-bit sum >= max(thr >> 15, 1)
// (f / 2^15 <= sum_i x_i + m * 65 / 2^15 with x_i = cs t_i / 2^15 and 64 units of float rounding
// per term; every 16-bit contribution exceeds x_i + 1 - 2^-9).  The docs that pass are staged by
// doc id only; k_join_rescore looks their postings up in the streams, forms f exactly as join_post
// does and applies the bin test: the candidate list k_select sees is the 32-bit kernel's, bit for
// bit.  A sum stays below 2^16: sum_i x_i < 2^15 (fx_mul) and at most 2 * 16 units of rounding.
// (The general forms scale exactly: cs / 2^15 is a power-of-two multiple, so their 16-bit value
// is the 32-bit one / 2^15 before the + 2.)  Half the visits, the same LDS bytes per visit;
// eligibility (irs_hip.hip join_half_ok): the plain disjunctions of a batch, no deleted docs in
// their segments (those entries leave the doc order k_join_rescore's search relies on).
template<int M>
__device__ __forceinline__ void join_pairs(unsigned char* smem, const JoinTileCtx& ctx,
                                           const JoinLane& T, uint32_t tile0, uint32_t ntile,
                                           uint32_t wv, uint32_t nw_log2, uint32_t& my_hits) {
  static_assert((M & kJHalf) != 0 && (M & kJCount) == 0, "join_pairs: plain disjunctions");
  const uint32_t* rng = reinterpret_cast<const uint32_t*>(smem + JoinOff::rng);
  const uint32_t* cum = reinterpret_cast<const uint32_t*>(smem + JoinOff::cum);
  const uint32_t tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const uint64_t safe = reinterpret_cast<uint64_t>(ctx.args->jterms);
  const uint32_t npair = (ntile + 1u) >> 1;
  auto begin = [&](uint32_t p, JoinRun& r) {   // (p >= npair: an empty share)
    uint32_t a = 0, n = 0, c = 0;
    if (lane < 2u * kMaxTerms && p < npair) {
      const uint32_t i = 2u * p + (lane >> 4), j = lane & (kMaxTerms - 1u);
      if (i < ntile) {   // (an odd tile count: the last pair's second tile is empty)
        a = rng[i * kMaxTerms + j];
        n = rng[(i + 1u) * kMaxTerms + j] - a;
      }
      c = cum[p * 2u * kMaxTerms + lane];
    }
    join_begin<M>(r, T, a, n, c, wv, nw_log2, safe, lane);
  };
  JoinRun r0, r1;
  begin(0, r0);
  begin(1, r1);
  const uint32_t thr = wave::uniform(ctx.thr);       // in 16-bit units, >= 1
  const uint32_t below = (thr - 1u) * 0x00010001u;   // both halves: the largest sum that is no candidate
  uint32_t hits2 = 0;                                // matches per half (<= 12 per visit and lane)
  auto end_pair = [&](uint32_t p) {
    __syncthreads();   // B1: every accumulation of the pair has landed
    const uint32_t doc0 = kDocMin + (tile0 + 2u * p) * kJoinTile;
    auto candidate = [&](uint32_t doc, uint32_t sum16) {   // rare: the doc and its 16-bit sum
      const uint64_t key = (uint64_t(sum16) << 32) | uint64_t(0xFFFFFFFFu - doc);
      const uint32_t slot = atomicAdd(ctx.ncand, 1u);
      if (slot < kJoinCands) {
        ctx.lc[slot] = key;
      } else {
        const uint32_t g = atomicAdd(&ctx.args->cand_count[ctx.q], 1u);
        if (g < ctx.cap) ctx.args->cands[uint64_t(ctx.q) * ctx.cap + g] = key;
      }
    };
    auto four = [&](uint32_t i, const uint32_t (&v)[4]) {
      wave::count_nonzero_halves4(hits2, v[0], v[1], v[2], v[3]);
      const uint32_t top = wave::pk_max_u16(wave::pk_max_u16(v[0], v[1]), wave::pk_max_u16(v[2], v[3]));
      if (wave::pk_max_u16(top, below) != below) {   // rare: some half reaches the threshold
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
          if ((v[k] & 0xFFFFu) >= thr) candidate(doc0 + i + k, v[k] & 0xFFFFu);
          if ((v[k] >> 16) >= thr) candidate(doc0 + kJoinTile + i + k, v[k] >> 16);
        }
      }
    };
    const uint32_t step = blockDim.x * 4u;
    for (uint32_t i = tid * 4u; i < kJoinTile; i += step) {
      uint32_t v0[4];
      wave::lds_take4(smem, JoinOff::acc + i * 4u, v0);
      four(i, v0);
    }
    __syncthreads();   // B2: accumulators are clear again
  };
  for (uint32_t p = 0; p < npair; p += 2u) {
    join_finish<M>(smem, r0, T, lane);
    begin(p + 2u, r0);
    end_pair(p);
    join_finish<M>(smem, r1, T, lane);
    begin(p + 3u, r1);
    if (p + 1u < npair) end_pair(p + 1u);
  }
  my_hits += (hits2 & 0xFFFFu) + (hits2 >> 16);
}

// Persistent workgroups pulling chunks of kJoinChunkTiles consecutive tiles of one unit
// (chunk-major ids, heaviest units first: score.h k_score).  Per tile and wavefront:
//   finish(u)     consume the requested entries of tile u, stream the rest of its share
//   begin(u + 1)  work out the share of tile u+1, request its first entries
//   barrier B1    every accumulation of tile u has landed
//   epilogue      every thread reads + clears its accumulators, counts matches, stages the
//                 candidates at or above the threshold bin (the requests fly behind it)
//   barrier B2    accumulators are clear again
// Candidates are staged per chunk; their global slots are reserved by one returning atomic
// whose latency hides behind the next chunk.
// The next chunk id for a workgroup that last worked on queue g0 (thread 0 only): its own
// queue's, else the first other queue that still has one; total (= base[kJoinQueues]): none.
__device__ __forceinline__ uint32_t join_pull(const JoinArgs* a, uint32_t g0, uint32_t first_raw,
                                              uint32_t& g_out) {
  g_out = g0;
  if (first_raw < a->base[g0 + 1u]) return first_raw;
  for (uint32_t k = 1; k < kJoinQueues; ++k) {
    const uint32_t g = (g0 + k) % kJoinQueues;
    if (a->base[g + 1u] == a->base[g]) continue;
    const uint32_t raw = atomicAdd(&a->work_counter[g], 1u);
    if (raw < a->base[g + 1u]) {
      g_out = g;
      return raw;
    }
  }
  return a->base[kJoinQueues];
}

enum : uint32_t {   // LDS scratch words
  kJGroup = 1,      // the queue the next chunk id came from
  kJChunk = 0,      // next chunk id
  kJNc = 2,         // kJNc + (parity): candidates staged by the current / previous chunk
  kJPendQ = 4,      // previous chunk: unit, reserved base, count
  kJPendBase = 5,
  kJPendN = 6,
};

// COUNT: the launch's units are conjunctions / min-match disjunctions (accumulators with match
// counts); a kernel of its own, so that the plain disjunctions' code stays as small as it is.
// HALF: plain disjunctions on paired tiles (join_pairs); k_join_rescore follows.
enum : int { kJKPlain = 0, kJKCount = 1, kJKHalf = 2 };
template<int KIND>
__global__ void __launch_bounds__(kTileThreadsMax) IRS_WAVES_PER_SIMD(8)
k_join_score(const JoinArgs* __restrict__ args) {
  RT_DYN_SMEM(smem);
  constexpr bool COUNT = KIND == kJKCount;
  constexpr bool HALF = KIND == kJKHalf;
  if (!wave::lds_is_at_zero(smem)) __builtin_trap();
  uint32_t* acc = reinterpret_cast<uint32_t*>(smem + JoinOff::acc);
  uint32_t* rng = reinterpret_cast<uint32_t*>(smem + JoinOff::rng);
  uint32_t* cum = reinterpret_cast<uint32_t*>(smem + JoinOff::cum);
  uint32_t* sig = reinterpret_cast<uint32_t*>(smem + JoinOff::sig);
  uint64_t* lcand = reinterpret_cast<uint64_t*>(smem + JoinOff::cand);
  uint32_t* vars = reinterpret_cast<uint32_t*>(smem + JoinOff::vars);
  const uint32_t tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const uint32_t wv = wave::uniform(tid >> 6);
  const uint32_t total_chunks = args->base[kJoinQueues];
  const uint32_t nw_log2 = args->nw_log2;
  const uint32_t cap = args->cand_cap;

  for (uint32_t i = tid; i < kJoinTile + 64u; i += blockDim.x) acc[i] = 0u;   // (+ the dummies)
  if (tid < 16u) vars[tid] = 0u;
  __syncthreads();
  if (tid == 0) {
    sig[0] = 0xFFFFFFFFu;
    const uint32_t g0 = blockIdx.x % kJoinQueues;
    uint32_t g = g0;
    vars[kJChunk] = join_pull(args, g0, atomicAdd(&args->work_counter[g0], 1u), g);
    vars[kJGroup] = g;
  }
  __syncthreads();
  uint32_t chunk = wave::uniform(vars[kJChunk]);
  uint32_t group = wave::uniform(vars[kJGroup]);
  uint32_t parity = 0;
  // thread 0: the previous chunk's reservation (a returning atomic in flight)
  uint32_t pend_q = 0, pend_n = 0, pend_base = 0;
  __syncthreads();

  while (chunk < total_chunks) {
    uint32_t next_raw = 0;   // (a returning atomic in flight until the hand-over)
    if (tid == 0) next_raw = atomicAdd(&args->work_counter[group], 1u);
    const uint32_t n_units = args->first[group + 1u] - args->first[group];
    const uint32_t local = chunk - args->base[group];
    const uint32_t q = wave::uniform(args->order[args->first[group] + local % n_units]);
    const uint32_t per_chunk = args->chunk_tiles;
    const uint32_t tile0 = (local / n_units) * per_chunk;
    const DevQuery qd = args->queries[q];
    const uint32_t n_tiles = qd.n_tiles;
    const uint32_t ntile = tile0 >= n_tiles ? 0u
                           : ((n_tiles - tile0) < per_chunk ? (n_tiles - tile0) : per_chunk);
    const uint32_t bs = args->bstar[q];
    uint32_t my_hits = 0;
    uint64_t* lc = lcand + parity * kJoinCands;
    uint32_t* ncand = vars + kJNc + parity;
    if (ntile) {
      // tile boundaries of the chunk, every term: rng[i][j] = bounds_j[tile0 + i]
      for (uint32_t e = tid; e < (ntile + 1u) * kMaxTerms; e += blockDim.x) {
        const uint32_t i = e / kMaxTerms, j = e % kMaxTerms;
        uint32_t v = 0;
        if (j < qd.n_terms)
          v = reinterpret_cast<const uint32_t*>(args->jterms[qd.first_term + j].bounds)[tile0 + i];
        rng[e] = v;
      }
      join_prologue(smem, qd, args->qterms, args->jterms);   // (its barrier publishes rng too)
      // per tile: the inclusive prefix of the terms' entry counts (what join_begin splits)
      if (HALF) {   // per pair of tiles: over the first tile's terms, then the second's
        const uint32_t npair = (ntile + 1u) >> 1;
        for (uint32_t e = tid; e < npair * 2u * kMaxTerms; e += blockDim.x) {
          const uint32_t p = e / (2u * kMaxTerms), l = e % (2u * kMaxTerms);
          uint32_t c = 0;
          for (uint32_t t = 0; t <= l; ++t) {
            const uint32_t i = 2u * p + t / kMaxTerms, j = t % kMaxTerms;
            if (i < ntile) c += rng[(i + 1u) * kMaxTerms + j] - rng[i * kMaxTerms + j];
          }
          cum[e] = c;
        }
      } else {
        for (uint32_t e = tid; e < ntile * kMaxTerms; e += blockDim.x) {
          const uint32_t i = e / kMaxTerms, j = e % kMaxTerms;
          uint32_t c = 0;
          for (uint32_t t = 0; t <= j; ++t)
            c += (rng[(i + 1u) * kMaxTerms + t] - rng[i * kMaxTerms + t]) >> kAblShift;
          cum[e] = c;
        }
      }
      __syncthreads();
      const JoinLane T = HALF ? join_lane_half(smem, lane) : join_lane(smem, lane);
      // every term through table slot 0?  (wave-uniform, the same in every wavefront)
      const bool simple = HALF ? wave::ballot((T.mode & ~uint32_t(kJoinHiHalf)) != 0u) == 0ull
                               : wave::ballot(T.mode != 0u) == 0ull;
      JoinTileCtx ctx;
      ctx.args = args;
      ctx.q = q;
      ctx.bs = bs;
      ctx.thr = bin_threshold<uint32_t>(bs, qd);
      ctx.fx_inv = qd.fx_inv;
      ctx.bin_scale = qd.bin_scale;
      ctx.cap = cap;
      ctx.lc = lc;
      ctx.ncand = ncand;
      ctx.need = query_need(qd.op);
      if (HALF) {
        const uint32_t thr16 = ctx.thr >> 15;
        ctx.thr = thr16 ? thr16 : 1u;
        if (simple) join_pairs<kJHalf | kJSimple>(smem, ctx, T, tile0, ntile, wv, nw_log2, my_hits);
        else join_pairs<kJHalf>(smem, ctx, T, tile0, ntile, wv, nw_log2, my_hits);
      } else if (COUNT) {   // conjunction / min-match: accumulators carry match counts
        if (simple) join_tiles<kJSimple | kJCount>(smem, ctx, T, tile0, ntile, wv, nw_log2, my_hits);
        else join_tiles<kJCount>(smem, ctx, T, tile0, ntile, wv, nw_log2, my_hits);
      } else {
        if (simple) join_tiles<kJSimple>(smem, ctx, T, tile0, ntile, wv, nw_log2, my_hits);
        else join_tiles<0>(smem, ctx, T, tile0, ntile, wv, nw_log2, my_hits);
      }
    }
    // ---- chunk hand-over: flush the PREVIOUS chunk's staged candidates (their reservation
    // has had a whole chunk to come back), reserve slots for this chunk's, publish hits
    if (tid == 0) {
      vars[kJPendQ] = pend_q;
      vars[kJPendBase] = pend_base;
      vars[kJPendN] = pend_n;
      uint32_t g = group;
      vars[kJChunk] = join_pull(args, group, next_raw, g);
      vars[kJGroup] = g;
    }
    my_hits = wave::reduce_add(my_hits);
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
      const uint32_t raw = *ncand;
      pend_n = raw < kJoinCands ? raw : kJoinCands;
      pend_q = q;
      pend_base = pend_n ? atomicAdd(&args->cand_count[q], pend_n) : 0u;
    }
    __syncthreads();   // everyone has read the hand-over words and the previous staging buffer
    parity ^= 1u;
    if (tid == 0) vars[kJNc + parity] = 0u;   // (the buffer flushed above becomes the next chunk's)
  }
  // the last chunk's candidates
  if (tid == 0) {
    vars[kJPendQ] = pend_q;
    vars[kJPendBase] = pend_base;
    vars[kJPendN] = pend_n;
  }
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
}
// Behind k_join_score<kJKHalf>: one workgroup per unit turns the staged docs into a candidate list
// that yields what the 32-bit kernel's yields — for a staged doc and every term of the query the
// posting is looked up in the term's stream (the tile's entries are in doc order), the
// contributions are formed and summed exactly as join_post<FORM, N, false> does (same tables, same
// expressions, same truncation: integer sums do not depend on the order), then the 32-bit
// threshold and the exact bin test of join_tiles' candidate() decide.
//
// Not every staged doc needs that.  A staged doc carries its 16-bit sum S, and
//     (S - 2 m - 1) 2^15  <  f  <  S 2^15          (m terms; join_pairs' bounds, both sides)
// so with S_k the k-th largest S of the unit, the k docs at or above it all have
// f > (S_k - 2 m - 1) 2^15 while a doc with S <= S_k - 2 m - 3 has f < (S_k - 2 m - 3) 2^15: it
// is not among the k best and k_select would drop it.  Only the docs with S >= S_k - 2 m - 2 are
// looked up (about k of the ~3 k an estimated threshold stages); if fewer than k of them pass the
// exact test — the k-th score sits within 2 m units of the threshold — the rest is looked up
// too, and the list is exactly the 32-bit kernel's.  Either way k_select sees >= k candidates
// exactly when that list has >= k, and the k best of both are the same docs with the same sums.
//
// The look-up: the docs of a term are spread evenly, so the posting of doc offset o among the n
// entries of a tile sits near n o / 12288: sixteen entries around that guess (four independent
// 16-byte loads, one round trip) settle most searches; a binary search over what is left of the
// range the others.
constexpr uint32_t kRescoreMax = 2048;      // docs in LDS at a time: doc + sum = 8 B each
constexpr uint32_t kRescoreThreads = 512;   // 4 workgroups of 19 KB LDS per CU: every unit of a batch in flight
// the 32-bit contribution of one entry: join_post<FORM, N, false>, one posting — the table entry
// the kernel reads from LDS evaluated on the spot (table_entry: what build_tables stores)
__device__ __forceinline__ uint32_t join_fx(const DevQTerm& qt, uint32_t e, float cs, uint32_t mode) {
  const int form = join_form(mode);
  const uint32_t norm = (e >> 2) & 255u;
  if (form == kJTable) {
    const float t = table_entry(qt.kind, qt.norm_const, qt.norm_length, (e >> 10) & 63u, norm);
    return static_cast<uint32_t>(wave::fma(cs, t, 1.f));
  }
  const float t = table_entry(qt.kind, qt.norm_const, qt.norm_length, 0u, norm);
  const float tf = static_cast<float>(join_tf(e));
  float scaled = (form == kJSqrt) ? wave::fast_sqrt(tf) * cs * t
                                  : wave::fma(-cs, wave::fast_rcp(wave::fma(tf, t, 1.f)), cs);
  wave::keep_f(scaled);
  return static_cast<uint32_t>(scaled) | 1u;
}
// the entry of doc offset `off` among ent[lo .. end) (ascending offsets), or 0xFFFFFFFF.
// [lo, hi] brackets the first entry at or behind the doc throughout.  A long range starts with
// one probe at the evenly-spread guess and corrects the guess by the local density (the error of
// the second guess is a few entries whatever the range); then sixteen entries around the guess
// (four independent 16-byte loads — behind the last stream kJoinSlack entries are readable), a
// gallop from the window's edge if it missed, a binary search over what is left.
__device__ __forceinline__ uint32_t join_find(const uint32_t* ent, uint32_t lo, const uint32_t end,
                                              const uint32_t off) {
  if (lo >= end) return 0xFFFFFFFFu;
  const uint32_t n = end - lo;
  uint32_t hi = end;
  uint32_t g = lo + (n * off) / kJoinTile;   // (n, off <= 12288: no overflow)
  if (g >= end) g = end - 1u;
  if (n > 256u) {
    const uint32_t eg = ent[g];
    const uint32_t og = eg >> 18;
    if (og == off) return eg;
    const int32_t d = int32_t(off) - int32_t(og);
    const int32_t g2 = int32_t(g) + (d * int32_t(n)) / int32_t(kJoinTile);
    if (og < off) lo = g + 1u; else hi = g;
    g = uint32_t(g2 < int32_t(lo) ? int32_t(lo) : g2);
    if (g > hi) g = hi;
    if (lo >= hi) return 0xFFFFFFFFu;   // (the doc would sit between g and its neighbour)
  }
  {
    uint32_t w0 = g > lo + 8u ? g - 8u : lo;
    if (w0 + 16u > hi) w0 = hi > lo + 16u ? hi - 16u : lo;
    uint32_t v[16];
#pragma unroll
    for (uint32_t q4 = 0; q4 < 4u; ++q4) {
      uint32_t x[4];
      wave::gload_u32x4_at(reinterpret_cast<uint64_t>(ent + w0 + 4u * q4), x);
      v[4u * q4] = x[0]; v[4u * q4 + 1u] = x[1]; v[4u * q4 + 2u] = x[2]; v[4u * q4 + 3u] = x[3];
    }
    uint32_t below = 0, hit = 0xFFFFFFFFu;
    const uint32_t live = hi - w0;   // window slots inside the bracket (>= 1)
#pragma unroll
    for (uint32_t i = 0; i < 16u; ++i) {
      const bool in = i < live;
      below += (in && (v[i] >> 18) < off) ? 1u : 0u;
      if (in && (v[i] >> 18) == off) hit = v[i];
    }
    if (hit != 0xFFFFFFFFu) return hit;
    const uint32_t seen = live < 16u ? live : 16u;
    if (below == 0u) {
      if (w0 == lo) return 0xFFFFFFFFu;
      hi = w0;                                   // everything seen lies behind the doc
      for (uint32_t s = 16u; hi > lo; s *= 4u) {  // gallop to the left
        if (hi - lo < s) break;
        const uint32_t p = hi - s;
        if ((ent[p] >> 18) >= off) { hi = p; } else { lo = p + 1u; break; }
      }
    } else if (below == seen) {
      lo = w0 + seen;                            // ... in front of it
      if (seen < 16u) return 0xFFFFFFFFu;        // (the bracket ended inside the window)
      for (uint32_t s = 16u; lo < hi; s *= 4u) {  // gallop to the right
        if (hi - lo < s) break;
        const uint32_t p = lo + s - 1u;
        if ((ent[p] >> 18) < off) { lo = p + 1u; } else { hi = p; break; }
      }
    } else {
      return 0xFFFFFFFFu;   // the doc would sit inside the window: no posting
    }
  }
  while (lo < hi) {   // the first entry at or behind the doc
    const uint32_t mid = (lo + hi) >> 1;
    if ((ent[mid] >> 18) < off) lo = mid + 1u; else hi = mid;
  }
  if (lo < end) {
    const uint32_t e = ent[lo];
    if ((e >> 18) == off) return e;
  }
  return 0xFFFFFFFFu;
}
__global__ void __launch_bounds__(kRescoreThreads)
k_join_rescore(const uint32_t* units, const DevQuery* queries, const DevQTerm* qterms,
               const JoinTerm* jterms, const uint32_t* bstar, uint64_t* cands,
               uint32_t* cand_count, uint32_t cap) {
  __shared__ uint32_t docs[kRescoreMax];
  __shared__ uint32_t fsum[kRescoreMax];
  __shared__ uint32_t hist[256];
  __shared__ uint32_t vars[8];
  __shared__ DevQTerm qts[kMaxTerms];
  __shared__ JoinTerm jts[kMaxTerms];
  enum : uint32_t { kOut = 0, kSel = 1, kPass = 2, kDigit = 3, kWant = 4 };
  const uint32_t tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const uint32_t q = units[blockIdx.x];
  const uint32_t n = cand_count[q];
  // (more staged docs than slots: k_select raises kStatusOverflow, the host re-runs the batch)
  if (!n || n > cap) return;
  const DevQuery qd = queries[q];
  if (tid == 0) vars[kOut] = vars[kSel] = vars[kPass] = 0u;
  if (tid < qd.n_terms) {
    qts[tid] = qterms[qd.first_term + tid];
    jts[tid] = jterms[qd.first_term + tid];
  }
  __syncthreads();
  const uint32_t bs = bstar[q];
  const uint32_t thr = bin_threshold<uint32_t>(bs, qd);
  const uint32_t nt = qd.n_terms;
  uint64_t* list = cands + uint64_t(q) * cap;
  // the docs[0 .. count) in LDS: look their postings up, sum; returns with the sums visible
  auto look_up = [&](uint32_t count) {
    for (uint32_t it = tid; it < count * nt; it += blockDim.x) {
      const uint32_t c = it / nt, j = it % nt;
      const uint32_t at = docs[c] - kDocMin;
      const uint32_t tile = at / kJoinTile, off = at - tile * kJoinTile;
      const uint32_t* bnd = reinterpret_cast<const uint32_t*>(jts[j].bounds);
      const uint32_t e = join_find(reinterpret_cast<const uint32_t*>(jts[j].entries), bnd[tile],
                                   bnd[tile + 1u], off);
      if (e != 0xFFFFFFFFu) atomicAdd(&fsum[c], join_fx(qts[j], e, jts[j].cs, jts[j].mode));
    }
    __syncthreads();
  };
  auto passes = [&](uint32_t f) {   // the exact test of join_tiles' candidate()
    const float v = f <= kMaxTerms ? 0.f : from_fixed<uint32_t>(f, qd.fx_inv);
    return f >= thr && score_bin(v, qd.bin_scale) >= bs;
  };
  auto key_of = [&](uint32_t c) {
    const uint32_t f = fsum[c];
    return make_key(f <= kMaxTerms ? 0.f : from_fixed<uint32_t>(f, qd.fx_inv), docs[c]);
  };
  // the 8-bit digit (at `shift`) holding the want-th largest S among the staged docs whose higher
  // bits equal `prefix`: histogram in LDS, suffix search by one wavefront (lane L owns the four
  // digits of group 63 - L); leaves the digit and the rank inside it in vars[kDigit], vars[kWant]
  auto digit_pass = [&](uint32_t shift, uint32_t prefix, uint32_t want) {
    if (tid < 256u) hist[tid] = 0u;
    __syncthreads();
    for (uint32_t c = tid; c < n; c += blockDim.x) {
      const uint32_t sv = uint32_t(list[c] >> 32);
      if (shift == 8u || (sv >> 8) == prefix) atomicAdd(&hist[(sv >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64u) {
      const uint32_t g = 63u - lane;
      const uint32_t sum = hist[4u * g] + hist[4u * g + 1u] + hist[4u * g + 2u] + hist[4u * g + 3u];
      const uint32_t incl = wave::inclusive_scan(sum);
      const uint64_t reach = wave::ballot(incl >= want);
      uint32_t digit = 0, rank = 1;
      if (reach) {
        const int src = __builtin_ctzll(reach);
        uint32_t cum = wave::bcast(incl - sum, src);
        const uint32_t gs = 63u - uint32_t(src);
        for (int i = 3; i >= 0; --i) {
          const uint32_t h = hist[4u * gs + uint32_t(i)];
          if (cum + h >= want) { digit = 4u * gs + uint32_t(i); rank = want - cum; break; }
          cum += h;
        }
      }
      if (lane == 0) {
        vars[kDigit] = digit;
        vars[kWant] = rank;
      }
    }
    __syncthreads();
  };
  bool done = false;
  if (n > qd.k && qd.k) {
    // the docs whose S is within 2 m + 2 of the k-th largest S: looked up first
    digit_pass(8u, 0u, qd.k);
    const uint32_t d1 = vars[kDigit], w1 = vars[kWant];
    __syncthreads();
    digit_pass(0u, d1, w1);
    const uint32_t sk = (d1 << 8) | vars[kDigit];
    const uint32_t cut = sk > 2u * nt + 2u ? sk - (2u * nt + 2u) : 0u;
    for (uint32_t c = tid; c < n; c += blockDim.x) {
      const uint64_t key = list[c];
      if (uint32_t(key >> 32) >= cut) {
        const uint32_t slot = atomicAdd(&vars[kSel], 1u);
        if (slot < kRescoreMax) {
          docs[slot] = key_hit(key).doc;
          fsum[slot] = 0u;
        }
      }
    }
    __syncthreads();
    const uint32_t sel = vars[kSel];
    if (sel <= kRescoreMax) {
      look_up(sel);
      uint32_t mine = 0;
      for (uint32_t c = tid; c < sel; c += blockDim.x) mine += passes(fsum[c]) ? 1u : 0u;
      if (mine) atomicAdd(&vars[kPass], mine);
      __syncthreads();
      // k of them pass (or nothing was left out): they are the list; else — the k-th score sits
      // within 2 m units of the threshold, rare — every staged doc is looked up below
      if (vars[kPass] >= qd.k || sel == n) {
        for (uint32_t c = tid; c < sel; c += blockDim.x)
          if (passes(fsum[c])) list[atomicAdd(&vars[kOut], 1u)] = key_of(c);
        done = true;
      }
    }
    __syncthreads();
  }
  if (!done) {
    // every staged doc, kRescoreMax at a time, compacted in place (a pass's docs sit in LDS before
    // its first survivor is written, and survivors never outnumber the docs read so far)
    for (uint32_t base = 0; base < n; base += kRescoreMax) {
      const uint32_t nb = n - base < kRescoreMax ? n - base : kRescoreMax;
      for (uint32_t c = tid; c < nb; c += blockDim.x) {
        docs[c] = key_hit(list[base + c]).doc;
        fsum[c] = 0u;
      }
      __syncthreads();
      look_up(nb);
      for (uint32_t c = tid; c < nb; c += blockDim.x)
        if (passes(fsum[c])) list[atomicAdd(&vars[kOut], 1u)] = key_of(c);
      __syncthreads();
    }
  }
  __syncthreads();
  if (tid == 0) cand_count[q] = vars[kOut];
}

}  // namespace irs_hip
