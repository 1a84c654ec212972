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
  const uint64_t at = reinterpret_cast<uint64_t>(pl);
  raw_load_packed_g<LAYOUT>(at, dbits, lane, da, db);
  raw_load_packed_g<LAYOUT>(at + 16u * dbits, fbits, lane, fa, fb);
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

// A position list read front to back: the block a delta lives in (its payload address and bit
// width: two dependent directory reads) is looked up once per 128 positions, not per position.
struct PosCursor {
  uint64_t payload;   // address of the current block's packed payload / of its all-equal value
  uint32_t blk;       // index of the current block (0xFFFFFFFF: none yet)
  uint32_t bits;      // its bit width (0: all-equal)
};
template<int LAYOUT>
__device__ __forceinline__ uint32_t pos_delta_cached(const DevSegment& seg, const DevPosTerm& pt,
                                                     PosCursor& c, uint32_t idx) {
  const uint32_t b = idx >> 7;
  if (b >= pt.nfull) return seg.ptail[pt.tail_row + (idx - (pt.nfull << 7))];  // read_tail_block :1515
  if (b != c.blk) {
    const uint64_t e = pt.row + b;
    c.blk = b;
    c.bits = seg.pblk_bits[e];
    c.payload = reinterpret_cast<uint64_t>(seg.pos + pt.pos_start + seg.pblk_off[e] + 1);
  }
  const uint8_t* pl = reinterpret_cast<const uint8_t*>(c.payload);
  if (c.bits == 0) {  // ALL_EQUAL (bitpack.hpp:159)
    uint32_t len;
    return vint_from(wave::load_u64(pl), &len);
  }
  return packed_at<LAYOUT>(pl, c.bits, idx & 127u);
}

// ------------------------------------------------------------ open time --

// Sum of the frequencies of every full doc block (-> exclusive scan -> blk_pos).
// Work split by directory row, as k_pack_payloads.
template<int LAYOUT>
__global__ void __launch_bounds__(kThreads)
k_freq_sums(DevSegment seg, uint64_t rows, uint32_t* sums) {
  const unsigned lane = threadIdx.x & 63u;
  IRS_FOR_ROWS(e, rows) {
    const uint32_t bits = seg.blk_bits[e];
    uint32_t d0, d1, f0, f1;
    decode_block<LAYOUT, true>(row_block(seg, e), bits & 0xFFu, bits >> 8,
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

// One workgroup per term lists the term's pos blocks (header byte -> size,
// bitpack::skip_block32) and decodes the vint tail (read_tail_block :1515-1537).  The headers
// form a chain of dependent one-byte reads; the stream is staged through LDS 8 KB at a time
// and the chain of a window is followed by pointer doubling (chain_orbit, kernels.h), all
// threads write the directory rows.  The 10 M-doc segment's longest term has 540 k pos blocks.
__global__ void __launch_bounds__(kChainThreads)
k_pos_directory(DevSegment seg, DevPosTerm* pterms, uint32_t* pblk_off, uint8_t* pblk_bits,
                uint32_t* ptail, const uint64_t* pos_end, uint32_t* status) {
  __shared__ __attribute__((aligned(16))) uint8_t win[kChainWindow + 64];
  __shared__ ChainTables s_chain;
  __shared__ uint32_t s_hdr[kChainCap];   // (offset in the window << 8) | bits
  __shared__ uint64_t s_cur;
  __shared__ uint32_t s_bad;
  const uint32_t tid = threadIdx.x;
  const uint32_t term = blockIdx.x;
  if (term >= seg.num_terms) return;
  const DevPosTerm pt = pterms[term];
  if (pt.total == 0) return;
  const uint64_t staged = seg.pos_len + kPadBytes;  // the device copy ends with zero padding
  uint64_t cur = pt.pos_start, win_lo = 0;
  uint32_t b = 0, bad = 0, lim = 0;
  // window [win_lo, win_lo + lim) of the stream, starting at the 16-byte line holding `at`
  auto refill = [&](uint64_t at) {
    win_lo = at & ~uint64_t(15);
    uint64_t bytes = staged - win_lo;
    if (bytes > kChainWindow) bytes = kChainWindow;
    bytes &= ~uint64_t(15);
    for (uint32_t o = tid * 16u; o < bytes; o += kChainThreads * 16u)
      *reinterpret_cast<ChainLine*>(win + o) =
        *reinterpret_cast<const ChainLine*>(seg.pos + win_lo + o);
    lim = uint32_t(bytes);
    __syncthreads();
  };
  while (b < pt.nfull && !bad) {
    if (cur + 8 > seg.pos_len) { bad = 1; break; }
    refill(cur);
    const uint32_t left = pt.nfull - b;
    uint32_t next;
    const uint32_t n = chain_orbit(win, s_chain, uint32_t(cur - win_lo), lim, seg.pos_len - win_lo,
                                   left < kChainCap ? left : kChainCap, s_hdr, &next, &bad,
                                   block_link{win});
    // a window from `cur` holds at least one whole block of a valid stream (<= 513 bytes)
    if (n == 0) bad = 1;
    // (offsets are kept in 32 bits: a term's positions beyond 4 GB are refused)
    if (n && win_lo + (s_hdr[n - 1] >> 8) - pt.pos_start > 0xFFFFFFFFull) bad = 1;
    for (uint32_t i = tid; i < n && !bad; i += kChainThreads) {
      const uint32_t rec = s_hdr[i];
      pblk_off[pt.row + b + i] = uint32_t(win_lo + (rec >> 8) - pt.pos_start);
      pblk_bits[pt.row + b + i] = uint8_t(rec);
    }
    b += n;
    cur = win_lo + next;
    __syncthreads();  // the window and the list are rewritten next
  }
  // where the writer says the tail starts (EndTerm :719-722; reader :2270-2278)
  if (!bad && pt.total > kBlock && pos_end[term] != cur - pt.pos_start) bad = 1;
  if (!bad && pt.tail_n) {
    if (cur + 1 > seg.pos_len) {
      bad = 1;
    } else {
      refill(cur);  // at most 127 vints of <= 5 bytes
      if (tid == 0) {
        uint64_t c = cur;
        uint32_t tb = 0;
        for (uint32_t i = 0; i < pt.tail_n; ++i) {
          if (c + 1 > seg.pos_len) { tb = 1; break; }
          uint32_t len;
          ptail[pt.tail_row + i] = vint_bytes(win + (c - win_lo), &len);
          c += len;
        }
        if (c > seg.pos_len) tb = 1;
        s_cur = c;
        s_bad = tb;
      }
      __syncthreads();
      cur = s_cur;
      bad = s_bad;
    }
  }
  if (tid == 0) {
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
constexpr uint32_t kConjWords = 32;   // 32-bit words of a wavefront's doc-range bitmaps: 1024 buckets
constexpr uint32_t kConjBuckets = 32u * kConjWords;

// One entry of a pilot list: a sampled lead item of a unit.
struct PhraseWg {
  uint32_t unit;        // (segment, query) execution unit
  uint32_t first_item;  // index of the lead item
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

// ---- block-driven execution (irs::And in conj.h, by_phrase below): shared pieces

// A wavefront's bucket table (conj.h ConjWave::first): the entry index (< n) of lead doc `doc` of
// bucket bk, or n: not a lead doc
__device__ __forceinline__ uint32_t lead_index(const uint8_t* first, const uint32_t* docs, uint32_t n,
                                               uint32_t bk, uint32_t s, uint32_t doc) {
  uint32_t t = first[bk];
  if (!t) return n;
  --t;
  if (s) {   // (wave-uniform) a bucket may hold several lead docs, in entry order
    while (docs[t] < doc && t + 1u < n) ++t;
  }
  return docs[t] == doc ? t : n;
}
struct alignas(16) ConjQuad {   // 16 bytes cleared at once
  uint32_t x, y, z, w;
};

// One lead item (a 128-posting block of the conjunction's rarest term, or its decoded vint
// tail), everything its wavefront needs to start decoding — written by the pre-pass so that
// the wavefront's first load is this record (one scalar load) instead of a chain of dependent
// ones (unit -> query -> term record -> directory words).
// A lead block whose docs spread over many blocks of the other terms (a rare lead against frequent
// terms: the reference's AndHighLow) is cut into 2^lg lead items of 128 >> lg consecutive postings:
// one wavefront would decode up to 128 blocks per other term one after the other, a chain of
// dependent loads with nobody to overlap it.
constexpr uint32_t kConjItemBlock = 0xFFFFFFu;   // ConjItem::item: block index (nblk <= 2^24)
constexpr uint32_t kConjSplitMax = 4;            //   | piece << 24 | lg << 28: 16 pieces at most
struct alignas(32) ConjItem {
  uint32_t unit;
  uint32_t item;    // block index in the lead's list; == its nblk: the vint tail (+ piece, lg above)
  uint32_t base;    // doc the block's first delta is relative to (formats_10.cpp:636)
  uint32_t r_lo;    // the item's docs lie in [r_lo, r_hi] (from the directory)
  uint32_t r_hi;
  uint32_t aoff;    // block in the packed-payload image, 16-byte units
  uint32_t bits;    // header bytes: dbits | fbits << 8
  uint32_t off;     // block in `.doc`, relative to the term's doc_start
};
static_assert(sizeof(ConjItem) == 32, "one s_load_dwordx8 per lead item");

// Pre-pass, one thread per lead item of every conjunction: the item's record, and where the
// other terms start for it — the binary search of a term's block directory for the first
// block reaching the lead item's first doc, SkipReader::Seek (skip_list.hpp:208-249), done
// once per (lead item, term) by ONE THREAD (inside k_conj a wavefront would walk the same
// dependent chain 64 lanes wide).  seek[(item_base + item) * (jt - 1) + slot of term i among the non-lead terms].
__global__ void __launch_bounds__(kThreads)
k_conj_seek(const DevSegment* segs, const DevQuery* queries, const DevTail* tails, uint32_t jt,
            const uint32_t* conj_units, const uint32_t* item_base /*[n_conj + 1]*/,
            uint32_t n_conj, const uint32_t* lead_of /*[unit] slot of the lead term; null: 0*/,
            const uint32_t* split_lg /*[n_conj] log2 of the pieces per lead block; null: 0*/,
            uint32_t* seek, ConjItem* recs) {
  const uint32_t t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= item_base[n_conj]) return;
  uint32_t lo = 0, hi = n_conj;   // the unit whose items hold t: last c with item_base[c] <= t
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (item_base[mid] <= t) lo = mid; else hi = mid;
  }
  const uint32_t lg = split_lg ? split_lg[lo] : 0u;
  const uint32_t unit = conj_units[lo], item = (t - item_base[lo]) >> lg;
  const uint32_t piece = (t - item_base[lo]) & ((1u << lg) - 1u);
  const DevQuery qd = queries[unit];
  const DevSegment& seg = segs[qd.seg];
  const DevTail* tl = tails + uint64_t(unit) * jt;
  const uint32_t lead = lead_of ? lead_of[unit] : 0u;
  const DevTail ld = tl[lead];
  ConjItem r{};
  r.unit = unit;
  r.item = item | (piece << 24) | (lg << 28);
  if (item < ld.nblk) {
    const uint64_t e = ld.dir_off + item;
    r.base = item ? seg.blk_last[e - 1] : kDocMin;
    r.r_lo = item ? r.base + 1u : kDocMin;
    r.r_hi = seg.blk_last[e];
    r.aoff = seg.blk_aoff[e];
    r.bits = seg.blk_bits[e];
    r.off = seg.blk_off[e];
  } else {
    r.r_lo = ld.first_doc;
    r.r_hi = ld.last_doc;
  }
  recs[t] = r;
  for (uint32_t i = 0; i < qd.n_terms; ++i) {   // row: the other terms in slot order
    if (i == lead) continue;
    const uint32_t* last = seg.blk_last + tl[i].dir_off;
    uint32_t a = 0, b = tl[i].nblk;  // lower_bound(last, r_lo)
    while (a < b) {
      const uint32_t mid = (a + b) >> 1;
      if (last[mid] < r.r_lo) a = mid + 1; else b = mid;
    }
    seek[uint64_t(t) * (jt - 1u) + (i < lead ? i : i - 1u)] = a;
  }
}

struct ConjArgs {
  const DevSegment* segs;
  const DevQuery* queries;
  const DevQTerm* qterms;
  const PhraseWg* wgs;          // pilot pass: {unit, lead item} per wavefront (the sampled items)
  uint32_t n_pilot;             // entries of the pilot list
  uint32_t n_items;             // full pass: lead items of all conjunctions (= records)
  const DevTail* tails;         // [unit][jt] (k_plan)
  const uint32_t* bstar;        // threshold bin per unit (0 = none)
  const uint32_t* seek;         // k_conj_seek
  const ConjItem* recs;         // k_conj_seek
  const uint32_t* unit_items;   // [nq] first record of the unit's lead items (conj units)
  const uint32_t* lead_of;      // [nq] by_phrase: slot of the unit's lead (rarest) term
  uint64_t* cands;
  uint32_t* cand_count;
  unsigned long long* hits;
  uint32_t* item_hits;          // [n_items] matches of every lead item (full pass; zeroed per run):
                                // a store instead of an atomic on the unit's counter — one hot
                                // address per unit kept every wavefront resident until its atomic
                                // had come back (1.6 of 6.4 ms per 1000 AND-2 queries);
                                // k_conj_hits adds them up
  unsigned long long* touched;  // [unit][2]: `.doc` + norm bytes actually decoded / read (full
                                // pass; per unit: one hot address would serialise the atomics);
                                // null unless the batch counts (irs_hip_batch_profile bit 1)
  uint32_t* hist;               // [unit][kBins], pilot pass only
  uint32_t jt;
  uint32_t cand_cap;
  uint32_t pilot_stride;        // pilot pass: lead items {phase, phase + P, ...}
  uint32_t wand;                // prune lead blocks by block-max bounds
  uint32_t* pruned;             // [unit] set when a lead block was skipped (k_select's underflow check)
};


// by_phrase, block driven.  The conjunction PhraseIterator::next runs first
// (phrase_iterator.hpp:590-596) is bounded by its rarest member — the reason Conjunction
// sorts its iterators by cost (conjunction.hpp:450-453) and lets the cheapest LEAD while
// the others seek().  Here ONE WAVEFRONT owns one 128-posting block (or the vint tail)
// of the lead term (its record and the other terms' start blocks come from k_conj_seek):
//   1. it decodes the lead block: 128 ascending docs into LDS and into a bitmap over the
//      block's doc range (conj.h: ConjWave), (P, tf) of each into the lead's row;
//   2. for every other term it finds the blocks that can hold one of the docs every term
//      so far reached — what seek() does through the skip list (skip_list.hpp:208-249): 64
//      directory entries per step, a lane each, tested against the alive bitmap ("is any of
//      them in (previous last, last]": two prefix-count reads); only the blocks that pass are
//      decoded, and each decoded posting tests its doc's bit: a hit leaves (P, tf) in the
//      term's row at the doc's rank and marks the doc alive for the next term;
//   3. docs that every term reached are the conjunction's matches: compacted, one lane each
//      merges the terms' position lists (FixedPhraseFrequency::NextPosition,
//      phrase_iterator.hpp:109-151): phrase frequency = #{p in first term : p + off_i in
//      term i for all i};
//   4. matches are scored with tf = phrase frequency; those at or above the unit's threshold
//      bin (pilot pass: histogrammed instead) are appended to its candidates, one reservation
//      per wavefront.
// Wavefronts are independent (no workgroup barrier).  MT = compile-time bound of the phrase
// length (cursor state stays in registers).
template<int MT>
struct PhraseWave {
  uint32_t docs[kBlock];
  uint32_t pidx[MT][kBlock];          // first position number of the doc in term i's list
  uint32_t tf[MT][kBlock];            // its frequency there (0: the term has not reached the doc)
  alignas(16) uint8_t first[kConjBuckets];   // bucket -> 1 + entry index of its first lead doc (conj.h)
  uint32_t bm[3][kConjWords + 4];     // bitmaps over [dlo, dhi]: lead docs, alive / marked
  uint8_t lpre[kConjWords + 4];       // lead-bitmap bits in the words before word w
  uint8_t apre[kConjWords + 4];       // the same for the alive bitmap of the current term
  DevPosTerm pt[MT];                  // the merge stage's per-term records
  uint32_t term[MT];
  uint32_t off[MT];
};

template<int LAYOUT, int MT>
__device__ __forceinline__ void phrase_item(const ConjArgs& A, uint32_t pilot /*1: histogram the
                                            scores of the sampled lead items, no candidates*/,
                                            PhraseWave<MT>* s_wave) {
  const uint32_t tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const uint32_t wv = wave::uniform(tid >> 6);
  uint32_t e = blockIdx.x * kPhraseWaves + wv;
  if (pilot) {
    if (e >= A.n_pilot) return;
    const PhraseWg w = A.wgs[e];
    e = wave::uniform(A.unit_items[w.unit] + w.first_item);
  } else if (e >= A.n_items) {
    return;
  }
  const ConjItem R = wave::sload<ConjItem>(reinterpret_cast<uint64_t>(A.recs) + uint64_t(e) * sizeof(ConjItem));
  const uint32_t unit = R.unit, item = R.item;
  const DevQuery qd = wave::sload<DevQuery>(reinterpret_cast<uint64_t>(A.queries) + uint64_t(unit) * sizeof(DevQuery));
  const uint32_t m = qd.n_terms;
  if (m == 0 || m > uint32_t(MT)) return;
  const DevSegment& seg = A.segs[qd.seg];   // (read field by field)
  const uint64_t tl_at = reinterpret_cast<uint64_t>(A.tails) + uint64_t(unit) * A.jt * sizeof(DevTail);
  auto term_tail = [&](uint32_t i) { return wave::sload<DevTail>(tl_at + i * sizeof(DevTail)); };
  const uint32_t lead = A.lead_of[unit];   // the term with the fewest postings leads
  const DevTail ld = term_tail(lead);
  const DevQTerm qt = A.qterms[qd.first_term];  // the phrase's scorer rides on its first term
  const uint32_t bs = pilot ? 0u : A.bstar[unit];
  PhraseWave<MT>& W = s_wave[wv];
  uint32_t* docs = W.docs;
  const uint32_t* seek = A.seek + uint64_t(e) * (A.jt - 1u);
  if (lane < m) {   // what the position merges read, a lane per term
    const DevTail t = A.tails[uint64_t(unit) * A.jt + lane];
    W.pt[lane] = seg.pterms[t.term];
    W.term[lane] = t.term;
    W.off[lane] = A.qterms[qd.first_term + lane].pad0;  // desired offset in the phrase
  }

  // ---- 1. the lead block: entry index 2*lane + h (block) or lane + 64*h (tail)
  uint32_t n = kBlock;
  uint32_t bytes = 0;   // (wave-uniform) encoded bytes of the doc blocks this wavefront decodes
  const bool counting = !pilot && A.touched != nullptr;   // (irs_hip_batch_profile bit 1)
  auto block_bytes = [](uint32_t bits) {
    const uint32_t db = bits & 0xFFu, fb = bits >> 8;
    return 2u + (db ? 16u * db : 1u) + (fb ? 16u * fb : 1u);
  };
  uint32_t ld_d[2], ld_e[2];
  bool live[2];   // the lane's lead docs that are docs of the segment at all (not deleted)
  {
    uint32_t f[2], p[2], estep;
    if (item < ld.nblk) {
      const uint64_t eb = ld.dir_off + item;
      if (counting) bytes += block_bytes(R.bits);
      uint32_t before;
      if (pk_both(R.bits & 0xFFu, R.bits >> 8)) {
        decode_packed_pos<LAYOUT>(seg.pk + (uint64_t(R.aoff) << 4), R.bits & 0xFFu, R.bits >> 8,
                                  R.base, lane, ld_d[0], ld_d[1], f[0], f[1], before);
      } else {
        decode_block_pos<LAYOUT>(seg.doc + ld.doc_start + R.off, R.bits & 0xFFu, R.bits >> 8,
                                 R.base, lane, ld_d[0], ld_d[1], f[0], f[1], before);
      }
      p[0] = seg.blk_pos[eb] - seg.blk_pos[ld.dir_off] + before;
      p[1] = p[0] + f[0];
      ld_e[0] = 2u * lane;
      estep = 1u;
    } else {
      n = ld.n;
      // positions in front of the tail = all frequencies of the full blocks (0 for a
      // list without blocks, whose dir_off has no rows of its own)
      const uint32_t base = seg.blk_pos[ld.dir_off + ld.nblk] - seg.blk_pos[ld.dir_off];
      tail_pidx(seg, ld.tail_row, n, base, lane, ld_d, f, p);
      ld_e[0] = lane;
      estep = 64u;
    }
    ld_e[1] = ld_e[0] + estep;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t idx = ld_e[h];
      docs[idx] = idx < n ? ld_d[h] : 0xFFFFFFFFu;
      // a deleted doc (SegmentReaderImpl::mask) keeps its place among the lead docs but counts as
      // not reached by the lead: it can never be a match
      live[h] = idx < n && !(seg.dead && doc_dead(seg.dead, ld_d[h]));
      for (uint32_t i = 0; i < m; ++i) {
        W.pidx[i][idx] = i == lead ? p[h] : 0u;
        W.tf[i][idx] = (i == lead && live[h]) ? f[h] : 0u;
      }
    }
    if (lane < (kConjWords + 4u) / 2u) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        W.bm[k][2u * lane] = 0u;
        W.bm[k][2u * lane + 1u] = 0u;
      }
    }
    static_assert(kConjBuckets == 64u * 16u, "one 16-byte store per lane clears `first`");
    reinterpret_cast<ConjQuad*>(W.first)[lane] = ConjQuad{0u, 0u, 0u, 0u};
  }
  wave::sync();
  const uint32_t dlo = wave::uniform(docs[0]), dhi = wave::uniform(docs[n - 1]);
  // bucket of a doc: (doc - dlo) >> s, below kConjBuckets (s = 0: one doc per bucket)
  const uint32_t span = dhi - dlo;
  const uint32_t s = span < kConjBuckets ? 0u
                     : 32u - uint32_t(__builtin_clz(span)) - (5u + uint32_t(__builtin_ctz(kConjWords)));
  const bool masked = seg.dead != nullptr;   // (wave-uniform)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (ld_e[h] < n) {
      const uint32_t bk = (ld_d[h] - dlo) >> s;
      atomicOr(&W.bm[0][bk >> 5], 1u << (bk & 31u));
      // (conj.h: the first other term may only reach the LIVE lead docs)
      if (masked && live[h]) atomicOr(&W.bm[2][bk >> 5], 1u << (bk & 31u));
      // the bucket's first lead doc (entries are in doc order: its predecessor lies in another)
      if (ld_e[h] == 0u || ((docs[ld_e[h] - 1u] - dlo) >> s) != bk) W.first[bk] = uint8_t(ld_e[h] + 1u);
    }
  }
  wave::sync();
  // bits in the words before word w (lane: words 2*lane, 2*lane+1; [kConjWords] = all)
  auto prefix = [&](const uint32_t* bmap, uint8_t* pre) {
    uint32_t p0 = 0, p1 = 0;
    if (lane < kConjWords / 2u) {
      p0 = uint32_t(__builtin_popcount(bmap[2u * lane]));
      p1 = uint32_t(__builtin_popcount(bmap[2u * lane + 1u]));
    }
    const uint32_t incl = wave::inclusive_scan(p0 + p1);
    if (lane < kConjWords / 2u) {
      pre[2u * lane] = uint8_t(incl - p0 - p1);
      pre[2u * lane + 1u] = uint8_t(incl - p1);
    }
    if (lane == kConjWords / 2u - 1u) pre[kConjWords] = uint8_t(incl);
    wave::sync();
    return wave::read_lane(incl, 63);
  };
  prefix(W.bm[0], W.lpre);

  // ---- 2. the other terms
  uint32_t step = 0;   // terms done so far (the lead aside)
  for (uint32_t i = 0; i < m; ++i) {
    if (i == lead) continue;
    const DevTail tl = term_tail(i);
    // alive = docs every term so far reached: the lead bitmap, then what the previous term
    // marked; `mark` collects what this term reaches
    const uint32_t mk = 1u + (step & 1u);
    const uint32_t* alive = step == 0u ? (masked ? W.bm[2] : W.bm[0]) : W.bm[3u - mk];
    uint32_t* mark = W.bm[mk];
    const uint8_t* apre = W.lpre;
    if (step > 0u || masked) {
      if (step > 0u && lane < kConjWords / 2u) {
        mark[2u * lane] = 0u;
        mark[2u * lane + 1u] = 0u;
      }
      apre = W.apre;
      if (prefix(alive, W.apre) == 0u) {   // no doc reached by every term so far: done
        if (!pilot && A.touched && lane == 0)
          atomicAdd(&A.touched[2u * unit], static_cast<unsigned long long>(bytes));
        return;
      }
    }
    ++step;
    const bool more_terms = MT > 2 && step + 1u < m;   // (the last term marks nothing: nobody reads it)
    auto alive_below = [&](uint32_t x) {
      return uint32_t(apre[x >> 5]) + uint32_t(__builtin_popcount(alive[x >> 5] & ((1u << (x & 31u)) - 1u)));
    };
    // a decoded posting of term i: is its doc one of the lead docs still alive?
    auto put = [&](uint32_t doc, uint32_t f, uint32_t p) {
      const uint32_t x = doc - dlo;
      if (f == 0 || x > span) return;
      const uint32_t bk = x >> s;
      const uint32_t t = lead_index(W.first, docs, n, bk, s, doc);
      if (t == n) return;
      // every earlier term — and the lead, whose deleted docs count as not reached — holds THIS doc
      if (W.tf[lead][t] == 0u) return;
      for (uint32_t j = 0; j < i; ++j)
        if (j != lead && W.tf[j][t] == 0u) return;
      W.pidx[i][t] = p;
      W.tf[i][t] = f;
      if (more_terms) atomicOr(&mark[bk >> 5], 1u << (bk & 31u));   // (alive for the next term)
    };
    if (tl.nblk) {
      // (integer addresses: loads through the global address space, see raw_load_packed_g)
      const uint64_t last_at = reinterpret_cast<uint64_t>(seg.blk_last + tl.dir_off);
      const uint64_t dir_at = reinterpret_cast<uint64_t>(seg.blk_dir + tl.dir_off);
      const uint64_t pos_at = reinterpret_cast<uint64_t>(seg.blk_pos + tl.dir_off);
      const uint64_t pk_at = reinterpret_cast<uint64_t>(seg.pk);
      const uint32_t pos0 = seg.blk_pos[tl.dir_off];
      const uint32_t b_first = seek[i < lead ? i : i - 1u];
      for (uint32_t b0 = b_first; b0 < tl.nblk; b0 += 64) {
        const uint32_t bl = b0 + lane;
        const bool valid = bl < tl.nblk;
        // the block's last doc, its directory record (which also holds the preceding block's
        // last doc) and its first position number in ONE round trip (conj.h)
        const uint32_t lst = valid ? wave::gload_u32(last_at, bl * 4u) : 0xFFFFFFFFu;
        BlkDir d{};
        uint32_t pos_l = 0;
        if (valid) {
          uint32_t w[4];
          wave::gload_u32x4(dir_at, bl * uint32_t(sizeof(BlkDir)), w);
          d = BlkDir{w[0], w[1], w[2], w[3]};
          pos_l = wave::gload_u32(pos_at, bl * 4u);
        }
        const uint32_t prv = bl ? d.prev_last : 0u;   // the block holds docs in (prv, lst]
        const bool reach = valid && prv < dhi && lst >= dlo;
        bool want = false;
        if (reach) {
          const uint32_t x0 = prv + 1u > dlo ? prv + 1u - dlo : 0u;
          const uint32_t x1 = (lst < dhi ? lst : dhi) - dlo;
          want = alive_below((x1 >> s) + 1u) > alive_below(x0 >> s);
        }
        uint64_t mask = wave::ballot(want);
        const bool more = wave::ballot(valid && prv >= dhi) == 0;  // no block started behind dhi yet
        while (mask) {
          const uint32_t k = uint32_t(__builtin_ctzll(mask));
          mask &= mask - 1;
          const uint32_t bits = wave::read_lane(d.bits, k);
          if (counting) bytes += block_bytes(bits);
          const uint32_t base = wave::read_lane(d.prev_last, k);
          uint32_t d0, d1, f0, f1, before;
          const uint32_t dbits = bits & 0xFFu, fbits = bits >> 8;
          // two wanted blocks at a time where both live in the packed image: their payload
          // loads are in flight together
          if (mask && pk_both(dbits, fbits)) {
            const uint32_t k2 = uint32_t(__builtin_ctzll(mask));
            const uint32_t bits2 = wave::read_lane(d.bits, k2);
            if (pk_both(bits2 & 0xFFu, bits2 >> 8)) {
              mask &= mask - 1;
              if (counting) bytes += block_bytes(bits2);
              const uint64_t pl1 = pk_at + (uint64_t(wave::read_lane(d.aoff, k)) << 4);
              const uint64_t pl2 = pk_at + (uint64_t(wave::read_lane(d.aoff, k2)) << 4);
              uint64_t da1, db1, fa1, fb1, da2, db2, fa2, fb2;
              raw_load_packed_g<LAYOUT>(pl1, dbits, lane, da1, db1);
              raw_load_packed_g<LAYOUT>(pl1 + 16u * dbits, fbits, lane, fa1, fb1);
              raw_load_packed_g<LAYOUT>(pl2, bits2 & 0xFFu, lane, da2, db2);
              raw_load_packed_g<LAYOUT>(pl2 + 16u * (bits2 & 0xFFu), bits2 >> 8, lane, fa2, fb2);
              uint32_t x0, x1;
              extract_fast<LAYOUT>(da1, db1, dbits, lane, x0, x1);
              extract_fast<LAYOUT>(fa1, fb1, fbits, lane, f0, f1);
              uint32_t dsum = x0 + x1, fsum = f0 + f1;
              wave::inclusive_scan2(dsum, fsum);
              uint32_t p0 = wave::read_lane(pos_l, k) - pos0 + (fsum - f0 - f1);
              put(base + dsum - x1, f0, p0);
              put(base + dsum, f1, p0 + f0);
              extract_fast<LAYOUT>(da2, db2, bits2 & 0xFFu, lane, x0, x1);
              extract_fast<LAYOUT>(fa2, fb2, bits2 >> 8, lane, f0, f1);
              dsum = x0 + x1;
              fsum = f0 + f1;
              wave::inclusive_scan2(dsum, fsum);
              const uint32_t base2 = wave::read_lane(d.prev_last, k2);
              p0 = wave::read_lane(pos_l, k2) - pos0 + (fsum - f0 - f1);
              put(base2 + dsum - x1, f0, p0);
              put(base2 + dsum, f1, p0 + f0);
              continue;
            }
          }
          if (pk_both(dbits, fbits)) {
            // both parts 1..31-bit packed: the 16-byte aligned copy in the packed image,
            // one funnel shift + one bit-field extract per value (as k_score's hot loop)
            decode_packed_pos<LAYOUT>(seg.pk + (uint64_t(wave::read_lane(d.aoff, k)) << 4), dbits,
                                      fbits, base, lane, d0, d1, f0, f1, before);
          } else {
            decode_block_pos<LAYOUT>(seg.doc + tl.doc_start + wave::read_lane(d.off, k), dbits,
                                     fbits, base, lane, d0, d1, f0, f1, before);
          }
          const uint32_t p0 = wave::read_lane(pos_l, k) - pos0 + before;
          put(d0, f0, p0);
          put(d1, f1, p0 + f0);
        }
        if (!more) break;
      }
    }
    if (tl.n && tl.first_doc <= dhi && tl.last_doc >= dlo) {  // vint tail / single doc
      const uint32_t base = seg.blk_pos[tl.dir_off + tl.nblk] - seg.blk_pos[tl.dir_off];
      uint32_t d[2], f[2], p[2];
      tail_pidx(seg, tl.tail_row, tl.n, base, lane, d, f, p);
      put(d[0], f[0], p[0]);
      put(d[1], f[1], p[1]);
    }
    wave::sync();
  }

  // ---- 3./4. lead docs every term reached, compacted: merge the position lists, score, emit
  bool h01[2];
  uint64_t m01[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t sl = lane + 64u * uint32_t(h);
    bool all = sl < n;
    for (uint32_t i = 0; i < m; ++i) all = all && W.tf[i][sl < n ? sl : 0u] != 0u;
    h01[h] = all;
    m01[h] = wave::ballot(all);
  }
  const uint64_t below = (1ull << lane) - 1ull;
  const uint32_t c0 = uint32_t(__builtin_popcountll(m01[0]));
  const uint32_t total = c0 + uint32_t(__builtin_popcountll(m01[1]));
  uint8_t* list = W.first;   // (the bucket table has served: room for the 128 entry indices)
  if (h01[0]) list[__builtin_popcountll(m01[0] & below)] = uint8_t(lane);
  if (h01[1]) list[c0 + uint32_t(__builtin_popcountll(m01[1] & below))] = uint8_t(lane + 64u);
  wave::sync();
  uint32_t my_hits = 0, my_pos = 0;
  // (the fields the position merges read, in registers: through the reference every
  // pos_delta of the serial merge loops would load them again)
  DevSegment ps{};
  ps.pos = seg.pos;
  ps.pblk_off = seg.pblk_off;
  ps.pblk_bits = seg.pblk_bits;
  ps.ptail = seg.ptail;
  ps.pos_base = seg.pos_base;
  for (uint32_t q0 = 0; q0 < total; q0 += 64) {
    bool cand = false;
    float score = 0.f;
    uint32_t doc = 0;
    if (q0 + lane < total) {
      const uint32_t sl = list[q0 + lane];
      uint32_t P[MT], T[MT], K[MT], V[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const bool on = uint32_t(i) < m;
        P[i] = on ? W.pidx[i][sl] : 0u;
        T[i] = on ? W.tf[i][sl] : 0u;
        K[i] = 0u;
        V[i] = ps.pos_base;  // pos_limits::invalid() (+ min() before the first delta, one-based)
      }
      // Walking every position of the first term counts the same matches as the
      // reference's lead.seek(sought - offset), which only skips positions that cannot match.
      uint32_t pf = 0, head = ps.pos_base;
      bool done = false;
      if (MT == 2 && m == 2u) {
        // Two terms: ONE loop that reads one position per trip, of whichever list is behind
        // (the term's records selected per lane).  The nested form below costs a wavefront the
        // SUM over lead positions of the LONGEST seek among its 64 docs; this one the longest
        // tf_a + tf_b — what very frequent phrases (thousands of matching docs per block of
        // the lead) are bound by: 43 -> 36 ms per 1000 such queries.  (Keeping two deltas of
        // either list in flight was tried and is slower: the loop is bound by the instructions
        // of a random-access position read, not by its latency.)
        const DevPosTerm pa = W.pt[0], pb = W.pt[1];
        const uint32_t off = W.off[1];
        uint32_t ka = 0, kb = 0, va = ps.pos_base, vb = ps.pos_base;
        PosCursor ca{0, 0xFFFFFFFFu, 0}, cb{0, 0xFFFFFFFFu, 0};
        for (;;) {
          bool adv_a;
          if (ka == 0u) {
            adv_a = true;                       // lead.next()
          } else if (kb == 0u || vb < va + off) {
            adv_a = false;                      // position::seek(target) :1578-1604
          } else {
            pf += vb == va + off ? 1u : 0u;     // reached the target, or sought too far
            adv_a = true;
          }
          if (adv_a ? ka == T[0] : kb == T[1]) break;   // exhausted: no later position can match
          DevPosTerm pt;
          pt.pos_start = adv_a ? pa.pos_start : pb.pos_start;
          pt.row = adv_a ? pa.row : pb.row;
          pt.nfull = adv_a ? pa.nfull : pb.nfull;
          pt.tail_row = adv_a ? pa.tail_row : pb.tail_row;
          PosCursor cur = adv_a ? ca : cb;
          const uint32_t d = pos_delta_cached<LAYOUT>(ps, pt, cur, adv_a ? P[0] + ka : P[1] + kb);
          ++my_pos;
          if (adv_a) {
            ca = cur;
            va += d;
            ++ka;
            if (va + off < va) break;           // !pos_limits::valid(term_position)
          } else {
            cb = cur;
            vb += d;
            ++kb;
          }
        }
        done = true;
      }
      for (uint32_t a = 0; a < T[0] && !done; ++a) {
        head += pos_delta<LAYOUT>(ps, W.pt[0], W.term[0], P[0] + a);  // lead.next()
        ++my_pos;
        bool match = true;
#pragma unroll
        for (int i = 1; i < MT; ++i) {
          if (uint32_t(i) < m && match && !done) {
            const uint32_t target = head + W.off[i];
            if (target < head) { done = true; break; }  // !pos_limits::valid(term_position)
            // position::seek(target) :1578-1604
            // (value_ is invalid until the first position is read: K[i] == 0)
            while ((K[i] == 0u || V[i] < target) && K[i] < T[i]) {
              V[i] += pos_delta<LAYOUT>(ps, W.pt[i], W.term[i], P[i] + K[i]);
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
        doc = docs[sl];
        // (a posting of the lead block: its norm from the posting-order copy of the column)
        const uint32_t nv = !seg.pnorm ? norm_value(seg, doc)
                            : (item < ld.nblk ? seg.pnorm[(ld.dir_off + item) * kBlock + sl]
                                              : seg.tail_norms[ld.tail_row + sl]);
        score = score_value(qt, pf, nv);
        const uint32_t bin = score_bin(score, qd.bin_scale);
        if (pilot) atomicAdd(&A.hist[uint64_t(unit) * kBins + bin], 1u);
        else cand = bin >= bs;   // below the pilot's threshold bin: cannot be among the top k
        ++my_hits;
      }
    }
    // one reservation per wavefront for all its candidates of this pass
    const uint64_t cm = wave::ballot(cand);
    if (cm) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&A.cand_count[unit], uint32_t(__builtin_popcountll(cm)));
      base = wave::read_lane(base, 0);
      const uint32_t slot = base + uint32_t(__builtin_popcountll(cm & below));
      if (cand && slot < A.cand_cap) A.cands[uint64_t(unit) * A.cand_cap + slot] = make_key(score, doc);
    }
  }
  if (pilot) return;
  my_hits = wave::reduce_add(my_hits);
  if (lane == 0 && my_hits) A.item_hits[e] = my_hits;
  if (A.touched) {   // (only when the batch counts: irs_hip_batch_profile bit 1)
    my_pos = wave::reduce_add(my_pos);
    if (lane == 0) {
      atomicAdd(&A.touched[2u * unit], static_cast<unsigned long long>(bytes));
      if (my_pos) atomicAdd(&A.touched[2u * unit + 1u], static_cast<unsigned long long>(my_pos));
    }
  }
}

template<int LAYOUT, int MT>
__global__ void __launch_bounds__(kPhraseWaves * 64)
k_phrase(ConjArgs A, uint32_t pilot) {
  __shared__ PhraseWave<MT> s_wave[kPhraseWaves];
  phrase_item<LAYOUT, MT>(A, pilot, s_wave);
}
// Two-word phrases (the usual case, BASELINE config 5) at the hardware's 8 wavefronts per SIMD: the
// kernel is a chain of dependent loads, resident wavefronts count for more than the 4 registers
// over 64 it would like (12 bytes of scratch): 7.34 -> 6.44 ms per 1000 phrases.  Longer phrases
// keep their registers (their LDS rows limit the occupancy anyway).
template<int LAYOUT>
__global__ void __launch_bounds__(kPhraseWaves * 64) RT_WAVES_PER_SIMD(8)
k_phrase2(ConjArgs A, uint32_t pilot) {
  __shared__ PhraseWave<2> s_wave[kPhraseWaves];
  phrase_item<LAYOUT, 2>(A, pilot, s_wave);
}

}  // namespace irs_hip
