// phrase.h — positions (`.pos`) and by_phrase with fixed offsets.
//
// Reference: the position attribute of a posting list, formats_10.cpp:1457-1682
// (position_impl::prepare / read_block / read_tail_block, position::next / seek / skip),
// its writer :894-933 and EndTerm :713-760; the phrase iterator
// core/search/phrase_iterator.hpp:75-166 (FixedPhraseFrequency) and :540-626
// (PhraseIterator: conjunction of the terms, then the phrase frequency as tf).
//
// The reference walks the `.pos` stream sequentially behind every doc iterator
// ("pend_pos" positions behind).  Here the stream is RANDOM ACCESS: position number P of
// a term (0-based over the whole list) sits in pos block P / 128 at slot P % 128, a pos
// block directory built at open says where each block starts and how wide it is, and
// one packed value is extracted by itself (both layouts).  P of a doc's first position
// is the exclusive prefix sum of the frequencies in front of it: per 128-doc block that
// sum is precomputed at open (DevSegment::blk_pos), inside the block one wavefront scan
// gives the rest.  So a doc tile decodes every (term, block) exactly like k_score does,
// records (P, tf) per doc slot in LDS, and only docs holding ALL phrase terms ever touch
// the position stream.
#pragma once
#include "kernels.h"

namespace irs_hip {

constexpr uint32_t kPhraseTile = 2048;     // docs per LDS tile
constexpr uint32_t kPhraseMaxTerms = 8;    // IRS_HIP_MAX_PHRASE_TERMS
constexpr uint32_t kPhraseChunk = 8;       // tiles per workgroup
constexpr uint32_t kPhraseThreads = 256;

// Value j (0..127) of one packed block payload of `bits` (1..32) bits per value:
// packed::at for the scalar layout (bit_packing.hpp), the same for simdcomp's 4-lane one.
template<int LAYOUT>
__device__ __forceinline__ uint32_t packed_at(const uint8_t* payload, uint32_t bits, uint32_t j) {
  const uint32_t mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
  if (LAYOUT == kSimd4) {
    // value j = 4r + l: SSE lane l, bit r*bits of that lane's stream; word k of it = u32 4k + l
    const uint32_t l = j & 3u, bit = (j >> 2) * bits, k = bit >> 5;
    const uint8_t* p = payload + 4u * (4u * k + l);
    const uint64_t w0 = wave::load_u32(p), w1 = wave::load_u32(p + 16);
    return uint32_t(((w1 << 32) | w0) >> (bit & 31u)) & mask;
  }
  const uint32_t bit = j * bits;
  return uint32_t(wave::load_u64(payload + 4u * (bit >> 5)) >> (bit & 31u)) & mask;
}

// Position delta number `idx` of a term (what position::next adds to value_, :1624-1626).
template<int LAYOUT>
__device__ __forceinline__ uint32_t pos_delta(const DevSegment& seg, const DevPosTerm& pt,
                                              uint32_t term, uint32_t idx) {
  const uint32_t b = idx >> 7;
  if (b < pt.nfull) {
    const uint64_t e = pt.row + b;
    const uint8_t* blk = seg.pos + pt.pos_start + seg.pblk_off[e];
    const uint32_t bits = seg.pblk_bits[e];
    if (bits == 0) {  // ALL_EQUAL (bitpack.hpp:159)
      uint32_t len;
      return vint_from(wave::load_u64(blk + 1), &len);
    }
    return packed_at<LAYOUT>(blk + 1, bits, idx & 127u);
  }
  return seg.ptail[uint64_t(term) * kBlock + (idx - (pt.nfull << 7))];  // read_tail_block :1515
}

// ------------------------------------------------------------ open time --

// Sum of the frequencies of every full doc block (-> exclusive scan -> blk_pos).
// grid = num_terms * slices, as k_pack_payloads.
template<int LAYOUT>
__global__ void __launch_bounds__(kThreads)
k_freq_sums(DevSegment seg, uint32_t slices, uint32_t* sums) {
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t slice = blockIdx.x % slices;
  const DevTerm t = seg.terms[blockIdx.x / slices];
  if (t.docs_count < 2) return;
  for (uint32_t b = slice * kWaves + (threadIdx.x >> 6); b < t.nblk; b += slices * kWaves) {
    const uint64_t e = t.dir_off + b;
    const uint32_t bits = seg.blk_bits[e];
    uint32_t d0, d1, f0, f1;
    decode_block<LAYOUT, true>(seg.doc + t.doc_start + seg.blk_off[e], bits & 0xFFu, bits >> 8,
                               0u, lane, d0, d1, f0, f1);
    const uint32_t s = wave::reduce_add(f0 + f1);
    if (lane == 0) sums[e] = s;
  }
}

// One wavefront per term; lane 0 walks the term's pos blocks (header byte -> size,
// bitpack::skip_block32) and decodes the vint tail (read_tail_block :1515-1537).
__global__ void __launch_bounds__(kThreads)
k_pos_directory(DevSegment seg, DevPosTerm* pterms, uint32_t* pblk_off, uint8_t* pblk_bits,
                uint32_t* ptail, const uint64_t* pos_end, uint32_t* status) {
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t term = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (term >= seg.num_terms || lane != 0) return;
  const DevPosTerm pt = pterms[term];
  if (pt.total == 0) return;
  uint64_t cur = pt.pos_start;
  bool bad = false;
  for (uint32_t b = 0; b < pt.nfull; ++b) {
    if (cur + 2 > seg.pos_len) { bad = true; break; }
    const uint32_t bits = seg.pos[cur];
    uint32_t size;
    if (bits == 0) {
      uint32_t len;
      (void)vint_from(wave::load_u64(seg.pos + cur + 1), &len);
      size = 1u + len;
    } else {
      size = 1u + 16u * bits;
    }
    if (bits > 32 || cur + size > seg.pos_len || cur - pt.pos_start > 0xFFFFFFFFull) {
      bad = true;
      break;
    }
    pblk_off[pt.row + b] = uint32_t(cur - pt.pos_start);
    pblk_bits[pt.row + b] = uint8_t(bits);
    cur += size;
  }
  // where the writer says the tail starts (EndTerm :719-722; reader :2270-2278)
  if (!bad && pt.total > kBlock && pos_end[term] != cur - pt.pos_start) bad = true;
  if (!bad) {
    for (uint32_t i = 0; i < pt.tail_n; ++i) {
      if (cur + 1 > seg.pos_len) { bad = true; break; }
      uint32_t len;
      ptail[uint64_t(term) * kBlock + i] = vint_from(wave::load_u64(seg.pos + cur), &len);
      cur += len;
    }
    if (cur > seg.pos_len) bad = true;
  }
  pterms[term].bytes = uint32_t(cur - pt.pos_start);
  if (bad) atomicOr(status, kStatusCorrupt);
}

// (P, tf) of the tail postings this lane owns: entries `lane` and `lane + 64` of the
// decoded tail (or the single doc).  base = positions in front of the tail.
__device__ __forceinline__ void tail_pidx(const DevSegment& seg, uint32_t term, uint32_t n,
                                          uint32_t base, unsigned lane, uint32_t (&doc)[2],
                                          uint32_t (&tf)[2], uint32_t (&pidx)[2]) {
  const uint64_t row = uint64_t(term) * kBlock;
  tf[0] = lane < n ? seg.tail_freqs[row + lane] : 0u;
  tf[1] = lane + 64u < n ? seg.tail_freqs[row + lane + 64u] : 0u;
  doc[0] = lane < n ? seg.tail_docs[row + lane] : 0u;
  doc[1] = lane + 64u < n ? seg.tail_docs[row + lane + 64u] : 0u;
  const uint32_t ia = wave::inclusive_scan(tf[0]);
  const uint32_t ta = wave::bcast(ia, 63);
  const uint32_t ib = wave::inclusive_scan(tf[1]);
  pidx[0] = base + ia - tf[0];
  pidx[1] = base + ta + ib - tf[1];
}

// Bit-exact test surface: every position of every doc of one term, doc after doc
// (what draining the position attribute behind doc_iterator::next yields).
// grid.x = nblk + 1 wave-sized items, kWaves per workgroup (as k_decode_term).
template<int LAYOUT>
__global__ void __launch_bounds__(kThreads)
k_decode_positions(DevSegment seg, uint32_t term, uint32_t* out) {
  const unsigned lane = threadIdx.x & 63u;
  const DevTerm t = seg.terms[term];
  const DevPosTerm pt = seg.pterms[term];
  const uint32_t item = blockIdx.x * kWaves + (threadIdx.x >> 6);
  uint32_t doc[2], tf[2] = {0u, 0u}, pidx[2] = {0u, 0u};
  if (item < t.nblk && t.docs_count > 1) {
    const uint64_t e = t.dir_off + item;
    const uint32_t bits = seg.blk_bits[e];
    decode_block<LAYOUT, true>(seg.doc + t.doc_start + seg.blk_off[e], bits & 0xFFu, bits >> 8,
                               0u, lane, doc[0], doc[1], tf[0], tf[1]);
    const uint32_t incl = wave::inclusive_scan(tf[0] + tf[1]);
    pidx[0] = seg.blk_pos[e] - seg.blk_pos[t.dir_off] + incl - tf[0] - tf[1];
    pidx[1] = pidx[0] + tf[0];
  } else if (item == t.nblk) {
    const uint32_t n = t.docs_count == 1 ? 1u : t.tail_n;
    const uint32_t base = seg.blk_pos[t.dir_off + t.nblk] - seg.blk_pos[t.dir_off];
    tail_pidx(seg, term, n, base, lane, doc, tf, pidx);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint32_t v = 0;  // pos_limits::invalid(): zero-based storage, first delta is the position
    for (uint32_t k = 0; k < tf[h]; ++k) {
      v += pos_delta<LAYOUT>(seg, pt, term, pidx[h] + k);
      out[pidx[h] + k] = v;
    }
  }
}

// ------------------------------------------------------------ query time --

// The score function the phrase iterator compiles (CompileScore with the phrase's
// aggregated stats; tf = phrase frequency): the reference's float expressions,
// bm25.cpp:281-282, 313, 348-359, tfidf.cpp:185-187, 251-253.
__device__ __forceinline__ float phrase_score(const DevSegment& seg, const DevQTerm& qt,
                                              uint32_t freq, uint32_t doc) {
  const float tf = static_cast<float>(freq);
  switch (qt.kind) {
    case kBM1:
      return qt.c0;
    case kBM15:
      return qt.c0 - qt.c0 / (1.f + tf / qt.norm_const);
    case kBM25Tiny: {
      const uint32_t n = seg.norms[doc - seg.norm_min_doc];
      const float inv = n ? 1.f / (qt.norm_const + qt.norm_length * static_cast<float>(n)) : 0.f;
      return qt.c0 - qt.c0 / (1.f + tf * inv);
    }
    case kBM25One: {
      const float inv = 1.f / (qt.norm_const + qt.norm_length);
      return qt.c0 - qt.c0 / (1.f + tf * inv);
    }
    case kBM25Wide: {
      const float c1 = qt.norm_const + qt.norm_length * static_cast<float>(norm_global(seg, doc));
      return qt.c0 - qt.c0 * c1 / (c1 + tf);
    }
    case kTfidf:
      return sqrtf(tf) * qt.c0;
    case kTfidfTiny: {
      const uint32_t n = seg.norms[doc - seg.norm_min_doc];
      const float r = n ? 1.f / sqrtf(static_cast<float>(n)) : 0.f;
      return sqrtf(tf) * qt.c0 * r;
    }
    default: {  // kTfidfWide
      const uint32_t n = norm_global(seg, doc);
      const float r = n ? 1.f / sqrtf(static_cast<float>(n)) : 0.f;
      return sqrtf(tf) * qt.c0 * r;
    }
  }
}

constexpr uint32_t phrase_smem_bytes(uint32_t m) {
  return m * kPhraseTile * 8u + kPhraseTile * 2u;
}

// One workgroup = kPhraseChunk consecutive doc tiles of one (segment, query) unit.
// Per tile:
//   1. every (term, block) reaching the tile is decoded by one wavefront; each posting
//      inside the tile leaves (P, tf) in its doc slot of the term's LDS row;
//   2. slots holding ALL terms (the conjunction PhraseIterator::next runs first) are
//      compacted, and one thread per such doc merges the terms' position lists
//      (FixedPhraseFrequency::NextPosition): phrase frequency = #{p in lead : p + off_i in
//      term i for all i};
//   3. matches are scored with tf = phrase frequency and appended to the unit's candidates
//      (all of them: k_select picks the top k).
// MT = compile-time bound of the phrase length (cursor state stays in registers).
template<int LAYOUT, int MT>
__global__ void __launch_bounds__(kPhraseThreads)
k_phrase(const DevSegment* segs, const DevQuery* queries, const DevQTerm* qterms, uint32_t jt,
         uint32_t cpq, const uint32_t* first, const DevTail* tails, uint64_t* cands,
         uint32_t cand_cap, uint32_t* cand_count, unsigned long long* hits) {
  RT_DYN_SMEM(smem);
  __shared__ DevPosTerm s_pt[MT];
  __shared__ DevTail s_tl[MT];
  __shared__ uint32_t s_off[MT];
  __shared__ uint32_t s_n;
  const uint32_t tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const uint32_t wv = tid >> 6, nw = blockDim.x >> 6;
  const uint32_t unit = blockIdx.x / cpq, chunk = blockIdx.x % cpq;
  const DevQuery qd = queries[unit];
  const uint32_t m = qd.n_terms;
  if (m == 0 || m > uint32_t(MT) || chunk * kPhraseChunk >= qd.n_tiles) return;  // uniform
  const DevSegment seg = segs[qd.seg];
  uint32_t* s_pidx = reinterpret_cast<uint32_t*>(smem);        // [m][kPhraseTile]
  uint32_t* s_tf = s_pidx + m * kPhraseTile;                   // [m][kPhraseTile]
  uint16_t* s_list = reinterpret_cast<uint16_t*>(s_tf + m * kPhraseTile);  // [kPhraseTile]
  if (tid < m) {
    s_tl[tid] = tails[uint64_t(unit) * jt + tid];
    s_pt[tid] = seg.pterms[s_tl[tid].term];
    s_off[tid] = qterms[qd.first_term + tid].pad0;  // desired offset in the phrase
  }
  const DevQTerm qt = qterms[qd.first_term];  // the phrase's scorer rides on its first term
  __syncthreads();
  uint32_t my_hits = 0;
  uint32_t tile_end = (chunk + 1u) * kPhraseChunk;
  if (tile_end > qd.n_tiles) tile_end = qd.n_tiles;
  for (uint32_t tile = chunk * kPhraseChunk; tile < tile_end; ++tile) {
    const uint32_t lo = kDocMin + tile * kPhraseTile;
    const uint32_t hi = lo + (kPhraseTile - 1u);
    for (uint32_t i = tid; i < m * kPhraseTile; i += blockDim.x) s_tf[i] = 0u;
    if (tid == 0) s_n = 0u;
    __syncthreads();
    // ---- 1. postings -> (P, tf) per doc slot
    uint32_t c = 0;  // work item counter, the same in every wavefront
    const uint32_t* row = first + qd.first_off + uint64_t(tile) * jt;
    for (uint32_t i = 0; i < m; ++i) {
      const DevTail tl = s_tl[i];
      if (tl.nblk) {
        const uint32_t b0 = row[i];
        uint32_t b1 = row[jt + i];  // a block may straddle the tile's end
        if (b1 > tl.nblk - 1u) b1 = tl.nblk - 1u;
        for (uint32_t b = b0; b <= b1; ++b) {
          if ((c++ % nw) != wv) continue;
          const uint64_t e = tl.dir_off + b;
          const uint32_t bits = seg.blk_bits[e];
          const uint32_t base = b ? seg.blk_last[e - 1] : kDocMin;
          uint32_t d0, d1, f0, f1;
          decode_block<LAYOUT, true>(seg.doc + tl.doc_start + seg.blk_off[e], bits & 0xFFu,
                                     bits >> 8, base, lane, d0, d1, f0, f1);
          const uint32_t incl = wave::inclusive_scan(f0 + f1);
          const uint32_t p0 = seg.blk_pos[e] - seg.blk_pos[tl.dir_off] + incl - f0 - f1;
          const uint32_t x0 = d0 - lo, x1 = d1 - lo;  // doc < lo wraps to a huge value
          if (x0 < kPhraseTile) {
            s_pidx[i * kPhraseTile + x0] = p0;
            s_tf[i * kPhraseTile + x0] = f0;
          }
          if (x1 < kPhraseTile) {
            s_pidx[i * kPhraseTile + x1] = p0 + f0;
            s_tf[i * kPhraseTile + x1] = f1;
          }
        }
      }
      if (tl.n && tl.first_doc <= hi && tl.last_doc >= lo) {  // vint tail / single doc
        if ((c++ % nw) == wv) {
          // positions in front of the tail = all frequencies of the full blocks
          // (0 for a list without blocks, whose dir_off has no rows of its own)
          const uint32_t base = seg.blk_pos[tl.dir_off + tl.nblk] - seg.blk_pos[tl.dir_off];
          uint32_t d[2], f[2], p[2];
          tail_pidx(seg, tl.term, tl.n, base, lane, d, f, p);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t x = d[h] - lo;
            if (f[h] && x < kPhraseTile) {
              s_pidx[i * kPhraseTile + x] = p[h];
              s_tf[i * kPhraseTile + x] = f[h];
            }
          }
        }
      }
    }
    __syncthreads();
    // ---- 2. conjunction: doc slots every term reached
    for (uint32_t s = tid; s < kPhraseTile; s += blockDim.x) {
      bool all = true;
      for (uint32_t i = 0; i < m; ++i) all = all && s_tf[i * kPhraseTile + s] != 0u;
      if (all) s_list[atomicAdd(&s_n, 1u)] = uint16_t(s);
    }
    __syncthreads();
    const uint32_t n_match = s_n;
    for (uint32_t k = tid; k < n_match; k += blockDim.x) {
      const uint32_t s = s_list[k];
      uint32_t P[MT], T[MT], K[MT], V[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const bool on = uint32_t(i) < m;
        P[i] = on ? s_pidx[i * kPhraseTile + s] : 0u;
        T[i] = on ? s_tf[i * kPhraseTile + s] : 0u;
        K[i] = 0u;
        V[i] = 0u;  // pos_limits::invalid()
      }
      // FixedPhraseFrequency::NextPosition (phrase_iterator.hpp:109-151).  Its
      // lead.seek(sought - offset) only skips lead positions that cannot match; walking
      // every lead position counts the same matches.
      uint32_t pf = 0, lead = 0;
      bool done = false;
      for (uint32_t a = 0; a < T[0] && !done; ++a) {
        lead += pos_delta<LAYOUT>(seg, s_pt[0], s_tl[0].term, P[0] + a);  // lead.next()
        bool match = true;
#pragma unroll
        for (int i = 1; i < MT; ++i) {
          if (uint32_t(i) < m && match && !done) {
            const uint32_t target = lead + s_off[i];
            if (target < lead) { done = true; break; }  // !pos_limits::valid(term_position)
            // position::seek(target) :1578-1604
            while (V[i] < target && K[i] < T[i]) {
              V[i] += pos_delta<LAYOUT>(seg, s_pt[i], s_tl[i].term, P[i] + K[i]);
              ++K[i];
            }
            if (V[i] < target) done = true;           // exhausted: no later lead can match
            else if (V[i] != target) match = false;   // sought too far from the lead
          }
        }
        if (match && !done) ++pf;
      }
      if (pf) {
        const uint32_t doc = lo + s;
        const float score = phrase_score(seg, qt, pf, doc);
        const uint32_t slot = atomicAdd(&cand_count[unit], 1u);
        if (slot < cand_cap) cands[uint64_t(unit) * cand_cap + slot] = make_key(score, doc);
        ++my_hits;
      }
    }
    __syncthreads();
  }
  my_hits = wave::reduce_add(my_hits);
  if (lane == 0 && my_hits) atomicAdd(&hits[unit], static_cast<unsigned long long>(my_hits));
}

}  // namespace irs_hip
