// pjoin.h — two-word phrases of FREQUENT words on joined posting streams.
//
// The block-driven k_phrase (phrase.h) pays ~1 900 wave-instructions per 128-posting block of the
// phrase's rarer word: its decode, a directory scan and block decodes in the other word.  That is
// the right shape when the rarer word is rare (its blocks are few, most of the other list is never
// touched) — the reference's Conjunction leads with the cheapest iterator for the same reason
// (conjunction.hpp:207-223, 450-453).  A phrase of two frequent words is the other extreme: every
// block of either list is needed, and the per-block bookkeeping is all overhead.  Such units run
// here instead (irs_hip.hip phrase_join_saving decides per unit):
//
//   k_join          (join.h) decodes both lists once per run into 4-byte entries — shared with
//                   every other unit of the batch that reads the same (segment, term);
//   k_phrase_acc    the conjunction as a direct-address join in LDS, doc tile by doc tile: the
//                   tile's entries of the word with FEWER postings there write their posting
//                   number at the doc's slot (plain stores: a list holds a doc once), the other
//                   word's entries probe (plain reads), a hit is a doc holding both words —
//                   PhraseIterator's conjunction (phrase_iterator.hpp:590-596) — and leaves
//                   (unit, doc, the doc's posting number in either list); the writer's entries
//                   clear their slots again.  No accumulator scan, no atomics on the slots: a tile
//                   costs 2 x (fewer) + (more) LDS accesses;
//   k_phrase_merge  one thread per such doc: position numbers of its postings from the posting-
//                   order table (DevSegment::pstart), the position merge of phrase.h
//                   (FixedPhraseFrequency::NextPosition), score with tf = phrase frequency,
//                   threshold bin of the unit (the block-driven pilot samples these units too),
//                   candidate.
// Results are the block-driven kernel's, bit for bit (same merge, same score expression, exact
// hit counts): tests/cases.py case_phrase_paths_agree.
#pragma once
#include "join.h"
#include "phrase.h"

namespace irs_hip {

constexpr uint32_t kPjStage = 2048;       // match records staged per workgroup between flushes
constexpr uint32_t kPjChunkTiles = 32;    // doc tiles per k_phrase_acc workgroup

// A doc that holds both words, with what the position merge needs of either posting: where its
// positions start in the word's list and how many there are.  Records leave k_phrase_acc in RUNS
// of one unit (a workgroup works on one unit), so the unit rides in a run directory, not here.
struct alignas(16) PhraseMatch {
  uint32_t doc;
  uint32_t p0, p1;      // first position number in the lists of slot 0 / slot 1
  uint32_t tf;          // frequencies: slot 0 | slot 1 << 8
};
static_assert(sizeof(PhraseMatch) == 16, "PhraseMatch");
struct PhraseRun {
  uint32_t unit;
  uint32_t count;
  unsigned long long first;   // index of the run's first record
};
static_assert(sizeof(PhraseRun) == 16, "PhraseRun");

// What the kernels need to know about one word of one unit (host-built, build_phrase_join);
// indexed like the unit's term slots (parallel to DevQTerm / JoinTerm).
struct alignas(16) PjTerm {
  uint64_t entries;      // the word's entry stream (JoinTerm::entries): tf of posting i
  uint64_t pstart;       // position number of the first position of posting i < n_block (u32 each)
  uint64_t tail_pstart;  // ... of tail posting i - n_block
  DevPosTerm pt;         // the word's position list
  uint32_t n_block;      // postings in full blocks: 128 * DevTerm::nblk
  uint32_t off;          // the word's offset in the phrase
  uint32_t bytes;        // encoded bytes of the word's list in `.doc` (irs_hip_batch_touched)
  uint32_t pad;
};
static_assert(sizeof(PjTerm) == 80, "PjTerm");

struct PjArgs {
  const DevSegment* segs;
  const DevQuery* queries;
  const DevQTerm* qterms;
  const JoinTerm* jterms;
  const PjTerm* pterms;         // [unit term slot]
  const uint32_t* units;        // the joined phrase units
  const uint32_t* bstar;
  PhraseMatch* matches;
  PhraseRun* runs;
  unsigned long long* counters; // [0]: records, [1]: runs
  uint64_t* cands;
  uint32_t* cand_count;
  unsigned long long* hits;
  unsigned long long* touched;  // [unit][2] (irs_hip_batch_profile bit 1), else null
  uint32_t n_units;
  uint32_t cpq;                 // workgroups (chunks of kPjChunkTiles tiles) per unit
  uint32_t cand_cap;
  uint32_t pad;
};

constexpr uint32_t kPjW = 2;   // entries per thread held in registers: of the writing word
constexpr uint32_t kPjP = 6;   //   ... of the probing word (what is beyond is loaded in the pass itself)

// One tile's entries as a thread holds them: the word with fewer postings in the tile writes.
struct PjTile {
  uint32_t wa, wn, pa, pn;   // first entry and entry count of the writing / the probing word
  uint32_t w;                // which word writes (0 / 1)
  uint32_t we[kPjW], pe[kPjP];
};
// A staged hit: the doc and its posting numbers; the positions' start and the frequencies are
// fetched when the stage is flushed (every thread a record: one round trip per flush, not per tile).
struct PjHit {
  uint32_t doc;
  uint32_t iw;   // posting number in the WRITING word's list | (which word wrote) << 31
  uint32_t ip;   // ... in the probing word's
};

// grid = n_units * cpq workgroups of kTileThreadsMax threads; dynamic LDS: two slot arrays (tile t
// is cleared while tile t + 1 is written) + the staged hits.
constexpr uint32_t kPjSmem = 2u * 2u * kJoinTile + kPjStage * uint32_t(sizeof(PjHit));
__global__ void __launch_bounds__(kTileThreadsMax)
k_phrase_acc(PjArgs A) {
  RT_DYN_SMEM(smem);
  uint16_t* slots = reinterpret_cast<uint16_t*>(smem);                              // [2][kJoinTile]
  PjHit* stage = reinterpret_cast<PjHit*>(smem + 2u * 2u * kJoinTile);              // [kPjStage]
  __shared__ uint32_t s_n;           // staged hits
  __shared__ unsigned long long s_base;
  const uint32_t tid = threadIdx.x, nthr = blockDim.x;
  const uint32_t ui = blockIdx.x / A.cpq, chunk = blockIdx.x % A.cpq;
  const uint32_t unit = A.units[ui];
  const DevQuery qd = A.queries[unit];
  if (qd.n_terms != 2u) return;
  const uint32_t n_tiles = (A.segs[qd.seg].num_docs + kJoinTile - 1u) / kJoinTile;
  const uint32_t tile0 = chunk * kPjChunkTiles;
  if (tile0 >= n_tiles) return;
  const uint32_t tile1 = tile0 + kPjChunkTiles < n_tiles ? tile0 + kPjChunkTiles : n_tiles;
  const JoinTerm j0 = A.jterms[qd.first_term], j1 = A.jterms[qd.first_term + 1u];
  const PjTerm t0 = A.pterms[qd.first_term], t1 = A.pterms[qd.first_term + 1u];
  const uint32_t* ent[2] = {reinterpret_cast<const uint32_t*>(j0.entries),
                            reinterpret_cast<const uint32_t*>(j1.entries)};
  const uint32_t* bnd[2] = {reinterpret_cast<const uint32_t*>(j0.bounds),
                            reinterpret_cast<const uint32_t*>(j1.bounds)};
  static_assert(kJoinTile < 65536u, "a tile's posting numbers fit the 16-bit slots");
  for (uint32_t i = tid; i < 2u * kJoinTile; i += nthr) slots[i] = 0u;
  if (tid == 0) {
    s_n = 0u;
    if (A.touched && chunk == 0u)   // (both lists are decoded in full, once per batch: k_join)
      atomicAdd(&A.touched[2u * unit], static_cast<unsigned long long>(t0.bytes) + t1.bytes);
  }
  __syncthreads();
  auto first_pos = [](const PjTerm& t, uint32_t i) {
    return i < t.n_block ? reinterpret_cast<const uint32_t*>(t.pstart)[i]
                         : reinterpret_cast<const uint32_t*>(t.tail_pstart)[i - t.n_block];
  };
  // the staged hits go out as one run of records: position starts and frequencies fetched here
  auto flush = [&]() {   // (whole workgroup; ends with the stage empty)
    __syncthreads();
    const uint32_t n = s_n;
    if (tid == 0 && n) {
      s_base = atomicAdd(&A.counters[0], static_cast<unsigned long long>(n));
      const unsigned long long r = atomicAdd(&A.counters[1], 1ull);
      A.runs[r] = PhraseRun{unit, n, s_base};
    }
    __syncthreads();
    const unsigned long long base = s_base;
    for (uint32_t i = tid; i < n; i += nthr) {
      const PjHit h = stage[i];
      const uint32_t w = h.iw >> 31, iw = h.iw & 0x7FFFFFFFu;
      const uint32_t i0 = w ? h.ip : iw, i1 = w ? iw : h.ip;
      PhraseMatch m;
      m.doc = h.doc;
      m.p0 = first_pos(t0, i0);
      m.p1 = first_pos(t1, i1);
      m.tf = join_tf(ent[0][i0]) | (join_tf(ent[1][i1]) << 8);
      A.matches[base + i] = m;
    }
    __syncthreads();
    if (tid == 0) s_n = 0u;
    __syncthreads();
  };
  // Software pipeline over the tiles: a tile's BOUNDS are requested two trips ahead, its first
  // ENTRIES per thread one trip ahead (they arrive while the previous tile is being joined).
  struct Bounds { uint32_t a0, n0, a1, n1; };
  auto read_bounds = [&](uint32_t tile) {
    Bounds b{0u, 0u, 0u, 0u};
    if (tile < tile1) {
      b.a0 = bnd[0][tile];
      b.n0 = bnd[0][tile + 1u] - b.a0;
      b.a1 = bnd[1][tile];
      b.n1 = bnd[1][tile + 1u] - b.a1;
    }
    return b;
  };
  auto open_tile = [&](const Bounds& b, PjTile& T) {
    T.wn = T.pn = 0u;
    T.wa = T.pa = 0u;
    T.w = 0u;
    if (!b.n0 || !b.n1) return;   // (no doc of the tile holds both words)
    T.w = b.n0 <= b.n1 ? 0u : 1u;
    T.wa = T.w ? b.a1 : b.a0;
    T.wn = T.w ? b.n1 : b.n0;
    T.pa = T.w ? b.a0 : b.a1;
    T.pn = T.w ? b.n0 : b.n1;
    const uint32_t* we = ent[T.w];
    const uint32_t* pe = ent[1u - T.w];
#pragma unroll
    for (uint32_t k = 0; k < kPjW; ++k) {
      const uint32_t i = tid + k * nthr;
      T.we[k] = i < T.wn ? we[T.wa + i] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < kPjP; ++k) {
      const uint32_t i = tid + k * nthr;
      T.pe[k] = i < T.pn ? pe[T.pa + i] : 0u;
    }
  };
  // the writing word of tile T: posting number (+ 1) at the doc's slot — plain stores, a list holds
  // a doc once; `val` false: the same slots back to zero
  auto write = [&](const PjTile& T, uint16_t* slot, bool val) {
    if (!T.wn) return;
    const uint32_t* went = ent[T.w];
#pragma unroll
    for (uint32_t k = 0; k < kPjW; ++k) {
      const uint32_t i = tid + k * nthr;
      if (i < T.wn) slot[T.we[k] >> 18] = val ? uint16_t(i + 1u) : uint16_t(0);
    }
    for (uint32_t i = tid + kPjW * nthr; i < T.wn; i += nthr)
      slot[went[T.wa + i] >> 18] = val ? uint16_t(i + 1u) : uint16_t(0);
  };
  PjTile cur, nxt;
  uint32_t staged = 0;   // s_n as every thread saw it while nobody was staging (the same in all)
  Bounds b1 = read_bounds(tile0 + 1u), b2{0u, 0u, 0u, 0u};
  open_tile(read_bounds(tile0), cur);
  write(cur, slots, true);
  __syncthreads();
  for (uint32_t tile = tile0; tile < tile1; ++tile) {
    uint16_t* slot = slots + ((tile - tile0) & 1u) * kJoinTile;
    uint16_t* other = slots + (((tile - tile0) & 1u) ^ 1u) * kJoinTile;
    b2 = read_bounds(tile + 2u);   // (requests only)
    open_tile(b1, nxt);            // (requests only: `nxt` is first used after the next barrier)
    if (cur.wn) {                  // (the same for every thread)
      const uint32_t* pent = ent[1u - cur.w];
      // a tile yields at most as many docs as the writing word has postings in it: room for them
      if (staged + cur.wn > kPjStage) flush();
      const bool dense = cur.wn > kPjStage;   // (more than the stage holds: flush pass by pass)
      // the probing word: a filled slot = a doc holding both words
      const uint32_t doc0 = kDocMin + tile * kJoinTile;
      auto probe = [&](uint32_t i, uint32_t e) {
        const uint32_t idx = e >> 18;
        const uint32_t v = slot[idx];
        if (!v) return;
        stage[atomicAdd(&s_n, 1u)] = PjHit{doc0 + idx, (cur.wa + v - 1u) | (cur.w << 31), cur.pa + i};
      };
#pragma unroll
      for (uint32_t k = 0; k < kPjP; ++k) {
        const uint32_t i = tid + k * nthr;
        if (i < cur.pn) probe(i, cur.pe[k]);
        if (dense && (k + 1u) * nthr < cur.pn) flush();   // (nthr <= kPjStage hits per pass)
      }
      for (uint32_t i0 = kPjP * nthr; i0 < cur.pn; i0 += nthr) {
        if (i0 + tid < cur.pn) probe(i0 + tid, pent[cur.pa + i0 + tid]);
        if (dense && i0 + nthr < cur.pn) flush();
      }
    }
    __syncthreads();
    staged = s_n;   // (nobody stages between this barrier and the next)
    // this tile's slots back to zero, the next tile's written (the other array): one phase
    write(cur, slot, false);
    write(nxt, other, true);
    __syncthreads();
    cur = nxt;
    b1 = b2;
  }
  flush();
}

// One wavefront per run of records (one unit): the unit's records are read ONCE per wavefront,
// then a lane per doc merges the two position lists, scores, tests the unit's threshold bin.
template<int LAYOUT>
__global__ void __launch_bounds__(kThreads)
k_phrase_merge(PjArgs A) {
  const unsigned long long n_runs = A.counters[1];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned long long wave0 = (unsigned long long)blockIdx.x * kWaves + (threadIdx.x >> 6);
  const unsigned long long n_waves = (unsigned long long)gridDim.x * kWaves;
  for (unsigned long long r = wave0; r < n_runs; r += n_waves) {
    const PhraseRun run = A.runs[r];
    const uint32_t unit = run.unit;
    const DevQuery qd = A.queries[unit];
    const PjTerm t0 = A.pterms[qd.first_term], t1 = A.pterms[qd.first_term + 1u];
    const DevQTerm qt = A.qterms[qd.first_term];   // the phrase's scorer rides on its first term
    const DevSegment& seg = A.segs[qd.seg];
    DevSegment ps{};
    ps.pos = seg.pos;
    ps.pblk_off = seg.pblk_off;
    ps.pblk_bits = seg.pblk_bits;
    ps.ptail = seg.ptail;
    ps.pos_base = seg.pos_base;
    const uint32_t bs = A.bstar[unit];
    const bool normed = needs_norm(qt.kind);
    uint32_t my_hits = 0, my_reads = 0;
    for (uint32_t i0 = 0; i0 < run.count; i0 += 64u) {
      const uint32_t i = i0 + lane;
      if (i >= run.count) continue;
      const PhraseMatch m = A.matches[run.first + i];
      const uint32_t pf = phrase_freq2<LAYOUT>(ps, t0.pt, t1.pt, t1.off, m.p0, m.tf & 0xFFu, m.p1,
                                               (m.tf >> 8) & 0xFFu, my_reads);
      if (!pf) continue;
      ++my_hits;
      const float score = score_value(qt, pf, normed ? norm_value(seg, m.doc) : 1u);
      if (score_bin(score, qd.bin_scale) >= bs) {
        const uint32_t at = atomicAdd(&A.cand_count[unit], 1u);
        if (at < A.cand_cap) A.cands[uint64_t(unit) * A.cand_cap + at] = make_key(score, m.doc);
      }
    }
    my_hits = wave::reduce_add(my_hits);
    if (lane == 0 && my_hits) atomicAdd(&A.hits[unit], static_cast<unsigned long long>(my_hits));
    if (A.touched) {
      my_reads = wave::reduce_add(my_reads);
      if (lane == 0 && my_reads)
        atomicAdd(&A.touched[2u * unit + 1u], static_cast<unsigned long long>(my_reads));
    }
  }
}

// ---- open-side: where every posting's positions start (DevSegment::pstart), once per segment

// grid = num_terms * slices, as k_posting_norms
template<int LAYOUT>
__global__ void __launch_bounds__(kThreads)
k_posting_pstart(DevSegment seg, uint32_t slices, uint32_t* pstart) {
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t slice = blockIdx.x % slices;
  const DevTerm t = seg.terms[blockIdx.x / slices];
  if (t.docs_count < 2) return;
  const uint32_t pos0 = seg.blk_pos[t.dir_off];
  for (uint32_t b = slice * kWaves + (threadIdx.x >> 6); b < t.nblk; b += slices * kWaves) {
    const uint64_t e = t.dir_off + b;
    const uint32_t bits = seg.blk_bits[e];
    uint32_t d0, d1, f0, f1;
    decode_block<LAYOUT, true>(seg.doc + t.doc_start + seg.blk_off[e], bits & 0xFFu, bits >> 8,
                               0u, lane, d0, d1, f0, f1);
    const uint32_t before = wave::inclusive_scan(f0 + f1) - f0 - f1;
    const uint32_t p = seg.blk_pos[e] - pos0 + before;
    const uint64_t both = (uint64_t(p + f0) << 32) | p;
    __builtin_memcpy(pstart + e * kBlock + 2u * lane, &both, 8);
  }
}
// ... and of the decoded tails / single docs: one thread per term
__global__ void __launch_bounds__(kThreads)
k_tail_pstart(DevSegment seg, uint32_t* tail_pstart) {
  const uint32_t term = blockIdx.x * kThreads + threadIdx.x;
  if (term >= seg.num_terms) return;
  const DevTerm t = seg.terms[term];
  const uint32_t n = t.docs_count == 1u ? 1u : t.tail_n;
  uint32_t p = seg.blk_pos[t.dir_off + t.nblk] - seg.blk_pos[t.dir_off];
  for (uint32_t i = 0; i < n; ++i) {
    tail_pstart[t.tail_row + i] = p;
    p += seg.tail_freqs[t.tail_row + i];
  }
}

}  // namespace irs_hip
