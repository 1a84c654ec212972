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
// gives the rest.  So the conjunction of the phrase's terms is formed from the doc blocks
// alone (k_phrase below), with (P, tf) riding along, and only docs holding ALL phrase terms
// ever touch the position stream.
#pragma once
#include "score.h"

namespace irs_hip {

constexpr uint32_t kPhraseMaxTerms = 8;    // IRS_HIP_MAX_PHRASE_TERMS

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

// decode_block (decode.h) for a field with positions: absolute docs d0/d1 and frequencies
// f0/f1 of postings 2*lane, 2*lane+1, and `before` = sum of the frequencies of the block's
// postings in front of 2*lane (where this lane's positions start inside the block).  The two
// prefix sums run as one interleaved chain.
template<int LAYOUT>
__device__ __forceinline__ void decode_block_pos(const uint8_t* blk, uint32_t dbits,
                                                 uint32_t fbits, uint32_t base, unsigned lane,
                                                 uint32_t& d0, uint32_t& d1, uint32_t& f0,
                                                 uint32_t& f1, uint32_t& before) {
  uint32_t x0, x1;
  const uint32_t size = read_block_pair<LAYOUT>(blk, dbits, lane, x0, x1);
  read_block_pair<LAYOUT>(blk + size, fbits, lane, f0, f1);
  uint32_t dsum = x0 + x1, fsum = f0 + f1;
  wave::inclusive_scan2(dsum, fsum);
  d1 = base + dsum;
  d0 = d1 - x1;
  before = fsum - f0 - f1;
}

// The same from the packed-payload image (both parts 1..31-bit packed, pk_both()):
// 16-byte aligned payloads, one funnel shift + one bit-field extract per value, as k_score's
// hot loop reads them.
template<int LAYOUT>
__device__ __forceinline__ void decode_packed_pos(const uint8_t* pl, uint32_t dbits,
                                                  uint32_t fbits, uint32_t base, unsigned lane,
                                                  uint32_t& d0, uint32_t& d1, uint32_t& f0,
                                                  uint32_t& f1, uint32_t& before) {
  uint64_t da, db, fa, fb;
  raw_load_packed<LAYOUT>(pl, dbits, lane, da, db);
  raw_load_packed<LAYOUT>(pl + 16u * dbits, fbits, lane, fa, fb);
  uint32_t x0, x1;
  extract_fast<LAYOUT>(da, db, dbits, lane, x0, x1);
  extract_fast<LAYOUT>(fa, fb, fbits, lane, f0, f1);
  uint32_t dsum = x0 + x1, fsum = f0 + f1;
  wave::inclusive_scan2(dsum, fsum);
  d1 = base + dsum;
  d0 = d1 - x1;
  before = fsum - f0 - f1;
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
  return seg.ptail[pt.tail_row + (idx - (pt.nfull << 7))];  // read_tail_block :1515
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

// term_meta::freq (what sizes the position buffers and the pos block directory) must be
// what the `.doc` stream really holds: the frequencies of the term's full blocks (blk_pos
// differences) plus those of its decoded tail.  One thread per term.
__global__ void __launch_bounds__(kThreads)
k_check_freq_totals(DevSegment seg, const DevPosTerm* pterms, uint32_t* status) {
  const uint32_t term = blockIdx.x * kThreads + threadIdx.x;
  if (term >= seg.num_terms) return;
  const DevTerm t = seg.terms[term];
  if (t.docs_count == 0) return;
  uint32_t sum = seg.blk_pos[t.dir_off + t.nblk] - seg.blk_pos[t.dir_off];
  const uint32_t n = t.docs_count == 1 ? 1u : t.tail_n;
  for (uint32_t i = 0; i < n; ++i) sum += seg.tail_freqs[t.tail_row + i];
  if (sum != pterms[term].total) atomicOr(status, kStatusCorrupt);
}

// One wavefront per term walks the term's pos blocks (header byte -> size,
// bitpack::skip_block32) and decodes the vint tail (read_tail_block :1515-1537).  The walk
// is a chain of dependent one-byte reads, so the stream is staged through LDS 8 KB at a
// time (all 64 lanes copy, lane 0 walks the window at LDS latency): 0.29 s -> see DESIGN.md
// for the 10 M-doc segment, whose longest term has 540 k pos blocks.
constexpr uint32_t kWalkWindow = 8192;
struct alignas(16) Bytes16 {
  uint64_t lo, hi;
};

__global__ void __launch_bounds__(kThreads)
k_pos_directory(DevSegment seg, DevPosTerm* pterms, uint32_t* pblk_off, uint8_t* pblk_bits,
                uint32_t* ptail, const uint64_t* pos_end, uint32_t* status) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kWaves][kWalkWindow];
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t wv = threadIdx.x >> 6;
  const uint32_t term = blockIdx.x * kWaves + wv;
  if (term >= seg.num_terms) return;
  const DevPosTerm pt = pterms[term];
  if (pt.total == 0) return;
  uint8_t* win = s_win[wv];
  const uint64_t staged = seg.pos_len + kPadBytes;  // the device copy ends with zero padding
  uint64_t cur = pt.pos_start, win_lo = 0, win_hi = 0;
  uint32_t b = 0;
  uint32_t bad = 0;
  // window [win_lo, win_hi) of the stream, starting at the 16-byte line holding `at`
  auto refill = [&](uint64_t at) {
    win_lo = at & ~uint64_t(15);
    uint64_t bytes = staged - win_lo;
    if (bytes > kWalkWindow) bytes = kWalkWindow;
    bytes &= ~uint64_t(15);
    for (uint32_t o = lane * 16u; o < bytes; o += 64u * 16u)
      *reinterpret_cast<Bytes16*>(win + o) =
        *reinterpret_cast<const Bytes16*>(seg.pos + win_lo + o);
    win_hi = win_lo + bytes;
    wave::sync();
  };
  while (b < pt.nfull && !bad) {
    if (cur + 8 > seg.pos_len) { bad = 1; break; }
    refill(cur);
    if (lane == 0) {
      // every block that starts with its header and a possible 5-byte vint inside the window
      while (b < pt.nfull && cur + 6 <= win_hi) {
        const uint32_t bits = win[cur - win_lo];
        uint32_t size;
        if (bits == 0) {
          uint32_t len;
          (void)vint_bytes(win + (cur + 1 - win_lo), &len);
          size = 1u + len;
        } else {
          size = 1u + 16u * bits;
        }
        if (bits > 32 || cur + size > seg.pos_len || cur - pt.pos_start > 0xFFFFFFFFull) {
          bad = 1;
          break;
        }
        pblk_off[pt.row + b] = uint32_t(cur - pt.pos_start);
        pblk_bits[pt.row + b] = uint8_t(bits);
        cur += size;
        ++b;
      }
    }
    b = wave::bcast(b, 0);
    bad = wave::bcast(bad, 0);
    const uint32_t lo = wave::bcast(uint32_t(cur), 0), hi = wave::bcast(uint32_t(cur >> 32), 0);
    cur = (uint64_t(hi) << 32) | lo;
    wave::sync();  // the window is rewritten next
  }
  // where the writer says the tail starts (EndTerm :719-722; reader :2270-2278)
  if (!bad && pt.total > kBlock && pos_end[term] != cur - pt.pos_start) bad = 1;
  if (!bad && pt.tail_n) {
    if (cur + 1 > seg.pos_len) {
      bad = 1;
    } else {
      refill(cur);  // at most 127 vints of <= 5 bytes
      if (lane == 0) {
        for (uint32_t i = 0; i < pt.tail_n; ++i) {
          if (cur + 1 > seg.pos_len) { bad = 1; break; }
          uint32_t len;
          ptail[pt.tail_row + i] = vint_bytes(win + (cur - win_lo), &len);
          cur += len;
        }
        if (cur > seg.pos_len) bad = 1;
      }
      bad = wave::bcast(bad, 0);
      const uint32_t lo = wave::bcast(uint32_t(cur), 0), hi = wave::bcast(uint32_t(cur >> 32), 0);
      cur = (uint64_t(hi) << 32) | lo;
    }
  }
  if (lane == 0) {
    pterms[term].bytes = uint32_t(cur - pt.pos_start);
    if (bad) atomicOr(status, kStatusCorrupt);
  }
}

// (P, tf) of the tail postings this lane owns: entries `lane` and `lane + 64` of the
// decoded tail (or the single doc).  base = positions in front of the tail.
__device__ __forceinline__ void tail_pidx(const DevSegment& seg, uint32_t row, uint32_t n,
                                          uint32_t base, unsigned lane, uint32_t (&doc)[2],
                                          uint32_t (&tf)[2], uint32_t (&pidx)[2]) {
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
    tail_pidx(seg, t.tail_row, n, base, lane, doc, tf, pidx);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint32_t v = seg.pos_base;  // pos_limits::invalid() (+ pos_limits::min() in the one-based formats)
    for (uint32_t k = 0; k < tf[h]; ++k) {
      v += pos_delta<LAYOUT>(seg, pt, term, pidx[h] + k);
      out[pidx[h] + k] = v;
    }
  }
}

// ------------------------------------------------------------ query time --

constexpr uint32_t kPhraseWaves = 4;  // wavefronts (= lead blocks) per workgroup

// One workgroup of k_phrase: kPhraseWaves consecutive lead blocks of one unit.
struct PhraseWg {
  uint32_t unit;        // (segment, query) execution unit
  uint32_t first_item;  // index of the workgroup's first lead block
};

// #{i < n : sorted[i] <= x}, knowing that it lies in [a, b]
__device__ __forceinline__ uint32_t count_le(const uint32_t* sorted, uint32_t a, uint32_t b,
                                             uint32_t x) {
  while (a < b) {
    const uint32_t mid = (a + b) >> 1;
    if (sorted[mid] <= x) a = mid + 1; else b = mid;
  }
  return a;
}

// by_phrase, block driven.  The conjunction PhraseIterator::next runs first
// (phrase_iterator.hpp:590-596) is bounded by its rarest member — the reason Conjunction
// sorts its iterators by cost (conjunction.hpp:450-453) and lets the cheapest LEAD while
// the others seek().  Here ONE WAVEFRONT owns one 128-posting block (or the vint tail)
// of the lead term:
//   1. it decodes the lead block: 128 ascending docs into LDS, (P, tf) of each into the
//      lead's row;
//   2. for every other term it finds the blocks that can hold one of those docs — what
//      seek() does through the skip list (skip_list.hpp:208-249): binary search of the
//      block directory for the first block reaching the lead block's first doc, then 64
//      directory entries per step, a lane each, tested against the lead docs ("is any of
//      them in (previous last, last]"); only the blocks that pass are decoded, and each
//      decoded posting looks its doc up among the 128 lead docs (binary search in LDS):
//      a hit leaves (P, tf) in the term's row;
//   3. lead docs that every term reached are the conjunction's matches: one lane each
//      merges the terms' position lists (FixedPhraseFrequency::NextPosition,
//      phrase_iterator.hpp:109-151): phrase frequency = #{p in first term : p + off_i in
//      term i for all i};
//   4. matches are scored with tf = phrase frequency and appended to the unit's
//      candidates (all of them: k_select picks the top k).
// No barrier after the prologue: wavefronts are independent.  MT = compile-time bound of
// the phrase length (cursor state stays in registers).
template<int LAYOUT, int MT>
__global__ void __launch_bounds__(kPhraseWaves * 64)
k_phrase(const DevSegment* segs, const DevQuery* queries, const DevQTerm* qterms, uint32_t jt,
         const PhraseWg* wgs, const DevTail* tails, uint64_t* cands, uint32_t cand_cap,
         uint32_t* cand_count, unsigned long long* hits,
         unsigned long long* touched /*[unit][2]: `.doc` bytes decoded, positions read*/,
         const uint32_t* bstar /*threshold bin per unit*/, uint32_t* hist /*[unit][kBins]*/,
         uint32_t pilot /*1: histogram the scores of the sampled lead items (launched one
                          wavefront per workgroup over the pilot list), no candidates*/) {
  __shared__ DevPosTerm s_pt[MT];
  __shared__ DevTail s_tl[MT];
  __shared__ uint32_t s_off[MT];
  __shared__ uint32_t s_docs[kPhraseWaves][kBlock];
  __shared__ uint32_t s_pidx[kPhraseWaves][MT][kBlock];
  __shared__ uint32_t s_tf[kPhraseWaves][MT][kBlock];
  const uint32_t tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const uint32_t wv = tid >> 6;
  const PhraseWg wg = wgs[blockIdx.x];
  const uint32_t unit = wg.unit;
  const DevQuery qd = queries[unit];
  const uint32_t m = qd.n_terms;
  const DevSegment seg = segs[qd.seg];
  if (tid < m && tid < uint32_t(MT)) {
    s_tl[tid] = tails[uint64_t(unit) * jt + tid];
    s_pt[tid] = seg.pterms[s_tl[tid].term];
    s_off[tid] = qterms[qd.first_term + tid].pad0;  // desired offset in the phrase
  }
  const DevQTerm qt = qterms[qd.first_term];  // the phrase's scorer rides on its first term
  const uint32_t bs = pilot ? 0u : bstar[unit];
  __syncthreads();
  if (m == 0 || m > uint32_t(MT)) return;
  uint32_t lead = 0, lead_n = 0xFFFFFFFFu;  // the term with the fewest postings leads
  for (uint32_t i = 0; i < m; ++i) {
    const uint32_t n = s_tl[i].nblk * kBlock + s_tl[i].n;
    if (n < lead_n) { lead_n = n; lead = i; }
  }
  const DevTail ld = s_tl[lead];
  const uint32_t item = wg.first_item + wv;
  if (item >= ld.nblk + (ld.n ? 1u : 0u)) return;  // whole wavefront
  uint32_t* docs = s_docs[wv];

  // ---- 1. the lead block: entry index 2*lane + h (block) or lane + 64*h (tail)
  uint32_t n = kBlock, e0, estep;
  uint32_t bytes = 0;   // (wave-uniform) encoded bytes of the doc blocks this wavefront decodes
  auto block_bytes = [](uint32_t bits) {
    const uint32_t db = bits & 0xFFu, fb = bits >> 8;
    return 2u + (db ? 16u * db : 1u) + (fb ? 16u * fb : 1u);
  };
  {
    uint32_t d[2], f[2], p[2];
    if (item < ld.nblk) {
      const uint64_t e = ld.dir_off + item;
      const uint32_t bits = seg.blk_bits[e];
      bytes += block_bytes(bits);
      const uint32_t base = item ? seg.blk_last[e - 1] : kDocMin;
      uint32_t before;
      if (pk_both(bits & 0xFFu, bits >> 8)) {
        decode_packed_pos<LAYOUT>(seg.pk + (uint64_t(seg.blk_aoff[e]) << 4), bits & 0xFFu,
                                  bits >> 8, base, lane, d[0], d[1], f[0], f[1], before);
      } else {
        decode_block_pos<LAYOUT>(seg.doc + ld.doc_start + seg.blk_off[e], bits & 0xFFu, bits >> 8,
                                 base, lane, d[0], d[1], f[0], f[1], before);
      }
      p[0] = seg.blk_pos[e] - seg.blk_pos[ld.dir_off] + before;
      p[1] = p[0] + f[0];
      e0 = 2u * lane;
      estep = 1u;
    } else {
      n = ld.n;
      // positions in front of the tail = all frequencies of the full blocks (0 for a
      // list without blocks, whose dir_off has no rows of its own)
      const uint32_t base = seg.blk_pos[ld.dir_off + ld.nblk] - seg.blk_pos[ld.dir_off];
      tail_pidx(seg, ld.tail_row, n, base, lane, d, f, p);
      e0 = lane;
      estep = 64u;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t idx = e0 + uint32_t(h) * estep;
      docs[idx] = idx < n ? d[h] : 0xFFFFFFFFu;
      for (uint32_t i = 0; i < m; ++i) {
        s_pidx[wv][i][idx] = i == lead ? p[h] : 0u;
        s_tf[wv][i][idx] = (i == lead && idx < n) ? f[h] : 0u;
      }
    }
  }
  wave::sync();
  const uint32_t dlo = docs[0], dhi = docs[n - 1];

  // a decoded posting of term i: is its doc one of the lead docs?
  // (w0, w1] = ranks of the lead docs that can equal it: those inside its block's doc range
  auto put = [&](uint32_t i, uint32_t doc, uint32_t f, uint32_t p, uint32_t w0, uint32_t w1) {
    if (f == 0 || doc < dlo || doc > dhi) return;
    const uint32_t c = count_le(docs, w0, w1, doc);
    if (c > w0 && docs[c - 1] == doc) {
      s_pidx[wv][i][c - 1] = p;
      s_tf[wv][i][c - 1] = f;
    }
  };
  // ---- 2. the other terms
  for (uint32_t i = 0; i < m; ++i) {
    if (i == lead) continue;
    const DevTail tl = s_tl[i];
    if (tl.nblk) {
      const uint32_t* last = seg.blk_last + tl.dir_off;
      const uint32_t pos0 = seg.blk_pos[tl.dir_off];
      uint32_t a = 0, b = tl.nblk;  // lower_bound(last, dlo): first block reaching dlo
      while (a < b) {
        const uint32_t mid = (a + b) >> 1;
        if (last[mid] < dlo) a = mid + 1; else b = mid;
      }
      for (uint32_t b0 = a; b0 < tl.nblk; b0 += 64) {
        const uint32_t bl = b0 + lane;
        const bool valid = bl < tl.nblk;
        const uint32_t lst = valid ? last[bl] : 0xFFFFFFFFu;
        const uint32_t prv = (valid && bl) ? last[bl - 1] : 0u;  // block holds docs in (prv, lst]
        const bool reach = valid && prv < dhi;
        const uint32_t cp_l = reach ? count_le(docs, 0u, n, prv) : 0u;
        const uint32_t cl_l = reach ? count_le(docs, cp_l, n, lst) : 0u;
        const bool want = cl_l > cp_l;   // some lead doc lies in (prv, lst]
        // the directory words of the wanted blocks, one lane each (coalesced), handed to
        // the whole wavefront by readlane when the block's turn comes
        const uint64_t e_l = tl.dir_off + bl;
        uint32_t bits_l = 0, off_l = 0, pos_l = 0, aoff_l = 0;
        if (want) {
          bits_l = seg.blk_bits[e_l];
          off_l = seg.blk_off[e_l];
          pos_l = seg.blk_pos[e_l];
          aoff_l = seg.blk_aoff[e_l];
        }
        const uint32_t base_l = bl ? prv : kDocMin;
        uint64_t mask = wave::ballot(want);
        const bool more = wave::ballot(valid && !reach) == 0;  // no block started behind dhi yet
        while (mask) {
          const uint32_t k = uint32_t(__builtin_ctzll(mask));
          mask &= mask - 1;
          const uint32_t bits = wave::read_lane(bits_l, k);
          bytes += block_bytes(bits);
          const uint32_t base = wave::read_lane(base_l, k);
          uint32_t d0, d1, f0, f1, before;
          const uint32_t dbits = bits & 0xFFu, fbits = bits >> 8;
          if (pk_both(dbits, fbits)) {
            // both parts 1..31-bit packed: the 16-byte aligned copy in the packed image,
            // one funnel shift + one bit-field extract per value (as k_score's hot loop)
            decode_packed_pos<LAYOUT>(seg.pk + (uint64_t(wave::read_lane(aoff_l, k)) << 4), dbits,
                                      fbits, base, lane, d0, d1, f0, f1, before);
          } else {
            decode_block_pos<LAYOUT>(seg.doc + tl.doc_start + wave::read_lane(off_l, k), dbits,
                                     fbits, base, lane, d0, d1, f0, f1, before);
          }
          const uint32_t p0 = wave::read_lane(pos_l, k) - pos0 + before;
          const uint32_t w0 = wave::read_lane(cp_l, k), w1 = wave::read_lane(cl_l, k);
          put(i, d0, f0, p0, w0, w1);
          put(i, d1, f1, p0 + f0, w0, w1);
        }
        if (!more) break;
      }
    }
    if (tl.n && tl.first_doc <= dhi && tl.last_doc >= dlo) {  // vint tail / single doc
      const uint32_t base = seg.blk_pos[tl.dir_off + tl.nblk] - seg.blk_pos[tl.dir_off];
      uint32_t d[2], f[2], p[2];
      tail_pidx(seg, tl.tail_row, tl.n, base, lane, d, f, p);
      put(i, d[0], f[0], p[0], 0u, n);
      put(i, d[1], f[1], p[1], 0u, n);
    }
  }
  wave::sync();

  // ---- 3./4. lead docs every term reached: merge the position lists, score, emit
  uint32_t my_hits = 0, my_pos = 0;
  for (uint32_t s = lane; s < n; s += 64) {
    uint32_t P[MT], T[MT], K[MT], V[MT];
    bool all = true;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const bool on = uint32_t(i) < m;
      P[i] = on ? s_pidx[wv][i][s] : 0u;
      T[i] = on ? s_tf[wv][i][s] : 0u;
      K[i] = 0u;
      V[i] = seg.pos_base;  // pos_limits::invalid() (+ min() before the first delta, one-based)
      all = all && (!on || T[i] != 0u);
    }
    if (!all) continue;
    // Walking every position of the first term counts the same matches as the
    // reference's lead.seek(sought - offset), which only skips positions that cannot match.
    uint32_t pf = 0, head = seg.pos_base;
    bool done = false;
    for (uint32_t a = 0; a < T[0] && !done; ++a) {
      head += pos_delta<LAYOUT>(seg, s_pt[0], s_tl[0].term, P[0] + a);  // lead.next()
      ++my_pos;
      bool match = true;
#pragma unroll
      for (int i = 1; i < MT; ++i) {
        if (uint32_t(i) < m && match && !done) {
          const uint32_t target = head + s_off[i];
          if (target < head) { done = true; break; }  // !pos_limits::valid(term_position)
          // position::seek(target) :1578-1604
          // (value_ is invalid until the first position is read: K[i] == 0)
          while ((K[i] == 0u || V[i] < target) && K[i] < T[i]) {
            V[i] += pos_delta<LAYOUT>(seg, s_pt[i], s_tl[i].term, P[i] + K[i]);
            ++K[i];
            ++my_pos;
          }
          if (V[i] < target) done = true;           // exhausted: no later position can match
          else if (V[i] != target) match = false;   // sought too far
        }
      }
      if (match && !done) ++pf;
    }
    if (pf) {
      const uint32_t doc = docs[s];
      const float score = score_value(qt, pf, norm_value(seg, doc));
      const uint32_t bin = score_bin(score, qd.bin_scale);
      if (pilot) {
        atomicAdd(&hist[uint64_t(unit) * kBins + bin], 1u);
      } else if (bin >= bs) {   // below the pilot's threshold bin: cannot be among the top k
        const uint32_t slot = atomicAdd(&cand_count[unit], 1u);
        if (slot < cand_cap) cands[uint64_t(unit) * cand_cap + slot] = make_key(score, doc);
      }
      ++my_hits;
    }
  }
  if (pilot) return;
  my_hits = wave::reduce_add(my_hits);
  if (lane == 0 && my_hits) atomicAdd(&hits[unit], static_cast<unsigned long long>(my_hits));
  if (touched) {   // (only when the batch counts: irs_hip_batch_profile bit 1)
    my_pos = wave::reduce_add(my_pos);
    if (lane == 0) {
      atomicAdd(&touched[2u * unit], static_cast<unsigned long long>(bytes));
      if (my_pos) atomicAdd(&touched[2u * unit + 1u], static_cast<unsigned long long>(my_pos));
    }
  }
}

}  // namespace irs_hip
