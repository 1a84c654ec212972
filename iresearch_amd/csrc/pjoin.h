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

constexpr uint32_t kPjStage = 1024;       // match records staged per workgroup between flushes
constexpr uint32_t kPjChunkTiles = 32;    // doc tiles per k_phrase_acc workgroup

// A doc that holds both words: its posting numbers in the lists of slot 0 and slot 1.
struct alignas(16) PhraseMatch {
  uint32_t unit;
  uint32_t doc;
  uint32_t i0, i1;
};
static_assert(sizeof(PhraseMatch) == 16, "PhraseMatch");

// What k_phrase_merge needs to know about one word of one unit (host-built, build_phrase_join);
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
  unsigned long long* n_matches;
  uint64_t* cands;
  uint32_t* cand_count;
  unsigned long long* hits;
  unsigned long long* touched;  // [unit][2] (irs_hip_batch_profile bit 1), else null
  uint32_t n_units;
  uint32_t cpq;                 // workgroups (chunks of kPjChunkTiles tiles) per unit
  uint32_t cand_cap;
  uint32_t pad;
};

// grid = n_units * cpq workgroups of kTileThreadsMax threads; dynamic LDS: kJoinTile slots + the
// staged records.
__global__ void __launch_bounds__(kTileThreadsMax)
k_phrase_acc(PjArgs A) {
  RT_DYN_SMEM(smem);
  uint32_t* slot = reinterpret_cast<uint32_t*>(smem);                                  // [kJoinTile]
  PhraseMatch* stage = reinterpret_cast<PhraseMatch*>(smem + 4u * kJoinTile);          // [kPjStage]
  __shared__ uint32_t s_n;           // staged records
  __shared__ unsigned long long s_base;
  const uint32_t tid = threadIdx.x;
  const uint32_t ui = blockIdx.x / A.cpq, chunk = blockIdx.x % A.cpq;
  const uint32_t unit = A.units[ui];
  const DevQuery qd = A.queries[unit];
  if (qd.n_terms != 2u) return;
  const uint32_t n_tiles = (A.segs[qd.seg].num_docs + kJoinTile - 1u) / kJoinTile;
  const uint32_t tile0 = chunk * kPjChunkTiles;
  if (tile0 >= n_tiles) return;
  const uint32_t tile1 = tile0 + kPjChunkTiles < n_tiles ? tile0 + kPjChunkTiles : n_tiles;
  const JoinTerm j0 = A.jterms[qd.first_term], j1 = A.jterms[qd.first_term + 1u];
  const uint32_t* ent[2] = {reinterpret_cast<const uint32_t*>(j0.entries),
                            reinterpret_cast<const uint32_t*>(j1.entries)};
  const uint32_t* bnd[2] = {reinterpret_cast<const uint32_t*>(j0.bounds),
                            reinterpret_cast<const uint32_t*>(j1.bounds)};
  for (uint32_t i = tid; i < kJoinTile; i += blockDim.x) slot[i] = 0u;
  if (tid == 0) {
    s_n = 0u;
    if (A.touched && chunk == 0u)   // (both lists are decoded in full, once per batch: k_join)
      atomicAdd(&A.touched[2u * unit], static_cast<unsigned long long>(A.pterms[qd.first_term].bytes) +
                                           A.pterms[qd.first_term + 1u].bytes);
  }
  __syncthreads();
  // the staged records go out: one reservation, coalesced copies
  auto flush = [&]() {   // (whole workgroup; ends with the stage empty)
    __syncthreads();
    const uint32_t n = s_n < kPjStage ? s_n : kPjStage;
    if (tid == 0 && n) s_base = atomicAdd(A.n_matches, static_cast<unsigned long long>(n));
    __syncthreads();
    const unsigned long long base = s_base;
    for (uint32_t i = tid; i < n; i += blockDim.x) A.matches[base + i] = stage[i];
    __syncthreads();
    if (tid == 0) s_n = 0u;
    __syncthreads();
  };
  for (uint32_t tile = tile0; tile < tile1; ++tile) {
    const uint32_t a0 = bnd[0][tile], b0 = bnd[0][tile + 1u];
    const uint32_t a1 = bnd[1][tile], b1 = bnd[1][tile + 1u];
    if (a0 == b0 || a1 == b1) continue;   // (the same for every thread: no doc of the tile holds both)
    // w: the word with fewer postings in this tile writes, the other probes
    const uint32_t w = (b0 - a0) <= (b1 - a1) ? 0u : 1u;
    const uint32_t wa = w ? a1 : a0, wb = w ? b1 : b0, pa = w ? a0 : a1, pb = w ? b0 : b1;
    const uint32_t* went = ent[w];
    const uint32_t* pent = ent[1u - w];
    for (uint32_t i = wa + tid; i < wb; i += blockDim.x) slot[went[i] >> 18] = i - wa + 1u;
    __syncthreads();
    const uint32_t doc0 = kDocMin + tile * kJoinTile;
    for (uint32_t i = pa + tid; i < pb; i += blockDim.x) {
      const uint32_t idx = pent[i] >> 18;
      const uint32_t v = slot[idx];
      if (v) {
        PhraseMatch m;
        m.unit = unit;
        m.doc = doc0 + idx;
        const uint32_t iw = wa + v - 1u;
        m.i0 = w ? i : iw;
        m.i1 = w ? iw : i;
        const uint32_t at = atomicAdd(&s_n, 1u);
        if (at < kPjStage) {
          stage[at] = m;
        } else {   // the stage is full: this record goes out by itself
          const unsigned long long g = atomicAdd(A.n_matches, 1ull);
          A.matches[g] = m;
        }
      }
    }
    __syncthreads();
    for (uint32_t i = wa + tid; i < wb; i += blockDim.x) slot[went[i] >> 18] = 0u;
    if (s_n >= kPjStage / 2u) flush();   // (s_n is stable here: read after the barrier above)
    else __syncthreads();
  }
  flush();
}

// One thread per record (grid-stride over the device-side count).
template<int LAYOUT>
__global__ void __launch_bounds__(kThreads)
k_phrase_merge(PjArgs A) {
  const unsigned long long n = *A.n_matches;
  const unsigned lane = threadIdx.x & 63u;
  for (unsigned long long base = (unsigned long long)blockIdx.x * kThreads; base < n;
       base += (unsigned long long)gridDim.x * kThreads) {
    const unsigned long long mi = base + threadIdx.x;
    bool cand = false, hit = false;
    uint32_t unit = 0xFFFFFFFFu, doc = 0, reads = 0;
    float score = 0.f;
    if (mi < n) {
      const PhraseMatch m = A.matches[mi];
      unit = m.unit;
      doc = m.doc;
      const DevQuery qd = A.queries[unit];
      const PjTerm t0 = A.pterms[qd.first_term], t1 = A.pterms[qd.first_term + 1u];
      const DevSegment& seg = A.segs[qd.seg];
      DevSegment ps{};
      ps.pos = seg.pos;
      ps.pblk_off = seg.pblk_off;
      ps.pblk_bits = seg.pblk_bits;
      ps.ptail = seg.ptail;
      ps.pos_base = seg.pos_base;
      auto first_pos = [](const PjTerm& t, uint32_t i) {
        return i < t.n_block ? reinterpret_cast<const uint32_t*>(t.pstart)[i]
                             : reinterpret_cast<const uint32_t*>(t.tail_pstart)[i - t.n_block];
      };
      const uint32_t P0 = first_pos(t0, m.i0), P1 = first_pos(t1, m.i1);
      const uint32_t T0 = join_tf(reinterpret_cast<const uint32_t*>(t0.entries)[m.i0]);
      const uint32_t T1 = join_tf(reinterpret_cast<const uint32_t*>(t1.entries)[m.i1]);
      const uint32_t pf = phrase_freq2<LAYOUT>(ps, t0.pt, t1.pt, t1.off, P0, T0, P1, T1, reads);
      if (pf) {
        const DevQTerm qt = A.qterms[qd.first_term];   // the phrase's scorer rides on its first term
        score = score_value(qt, pf, needs_norm(qt.kind) ? norm_value(seg, doc) : 1u);
        cand = score_bin(score, qd.bin_scale) >= A.bstar[unit];
        hit = true;
      }
    }
    // hits: the lanes of a wavefront mostly share a unit (records come out unit after unit)
    const uint32_t u0 = wave::bcast(unit, 0);
    const bool same = wave::ballot(unit != u0 && unit != 0xFFFFFFFFu) == 0ull;
    if (same) {
      const uint64_t hm = wave::ballot(hit);
      if (lane == 0 && hm && u0 != 0xFFFFFFFFu)
        atomicAdd(&A.hits[u0], static_cast<unsigned long long>(__builtin_popcountll(hm)));
      if (A.touched) {
        const uint32_t r = wave::reduce_add(reads);
        if (lane == 0 && r && u0 != 0xFFFFFFFFu)
          atomicAdd(&A.touched[2u * u0 + 1u], static_cast<unsigned long long>(r));
      }
    } else {
      if (hit) atomicAdd(&A.hits[unit], 1ull);
      if (A.touched && reads) atomicAdd(&A.touched[2u * unit + 1u], static_cast<unsigned long long>(reads));
    }
    if (cand) {
      const uint32_t at = atomicAdd(&A.cand_count[unit], 1u);
      if (at < A.cand_cap) A.cands[uint64_t(unit) * A.cand_cap + at] = make_key(score, doc);
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
