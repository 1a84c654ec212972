// oracle/search_oracle.cpp — TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// CPU restatement of the reference's query execution for by_term / Or / And of
// terms on one field, driven the way utils/index-search.cpp drives it:
//   statistics   core/search/bm25.cpp:366-410 (BM25::collect),
//                core/search/tfidf.cpp:263-278 (TFIDF::collect),
//                core/search/term_filter.cpp:102-125 (summed over segments)
//   score fns    core/search/bm25.cpp:262-364, 416-490; tfidf.cpp:185-261, 280-352
//   OR of 2      core/search/disjunction.hpp:204-358 (basic_disjunction)
//   OR of >=3    core/search/disjunction.hpp:889-1369 (block_disjunction, 512-doc window)
//   AND          core/search/conjunction.hpp:95-223, 436-490
//   top-k        utils/index-search.cpp:719-787 (std heap algorithms, as there)
// Scores are float, evaluated left to right without FMA contraction
// (built with -ffp-contract=off, no -march flags).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "oracle.h"
#include "oracle_internal.h"

namespace {

constexpr uint32_t kEof = UINT32_MAX;  // doc_limits::eof()

enum Kind {
  kBM1,         // bm25.cpp:262-286  constant num
  kBM15,        // bm25.cpp:288-319
  kBM25Tiny,    // bm25.cpp:348-353  (Norm2, 1 byte, norm_cache)
  kBM25Wide,    // bm25.cpp:355-359  (Norm2, wider)
  kBM25NoNorm,  // bm25.cpp:487-489  norm == 1 via the tiny path
  kTfidf,       // tfidf.cpp:251      no norms
  kTfidfTiny,   // tfidf.cpp:253 with kRSQRT.get<false>
  kTfidfWide,   // tfidf.cpp:253 with kRSQRT.get<true>
  kBM25Legacy,  // bm25.cpp:333-337 + BM25NormAdapter<kNorm> :242-249 (legacy `Norm`: float 1/sqrt(len))
  kTfidfLegacy, // tfidf.cpp:214-219 TFIDFNormAdapter<kNorm>: the stored float as it is
};

struct TermScorer {
  Kind kind;
  float c0;  // BM25: boost*(k+1)*idf (bm25.cpp:201); TFIDF: boost*idf (:199)
  float norm_const;
  float norm_length;
  const float* cache;  // BM25Stats::norm_cache
  const uint8_t* norms;
  uint32_t width;
};

inline uint32_t read_norm(const TermScorer& s, uint32_t doc) {
  // dense_fixed_length_column: data_ + len*(doc - min), min == 1
  // (columnstore2.cpp:736-740); Norm2 values are big-endian (norm.hpp:170-182)
  const uint8_t* p = s.norms + size_t(s.width) * (doc - 1);
  switch (s.width) {
    case 1:
      return p[0];
    case 2:
      return (uint32_t(p[0]) << 8) | p[1];
    default:
      return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) |
             (uint32_t(p[2]) << 8) | p[3];
  }
}

inline float read_legacy_norm(const TermScorer& s, uint32_t doc) {
  // Norm::MakeReader (norm.hpp:57-69): the float the writer stored for the doc, 1/sqrt(len);
  // here a dense little-endian float array, doc 1 first
  float v;
  std::memcpy(&v, s.norms + size_t(4) * (doc - 1), 4);
  return v;
}

inline float rsqrt_cached(uint32_t i) {
  // kRSQRT = cache_func<uint32_t, 2048>(1, 1/sqrt(i)); cache_[0] == 0
  // (tfidf.cpp:74-76, misc.hpp:60-66); computed fallback beyond the table
  if (i == 0) return 0.f;
  return 1.f / std::sqrt(static_cast<float>(i));
}

inline float score_posting(const TermScorer& s, uint32_t freq, uint32_t doc) {
  switch (s.kind) {
    case kBM1:
      return s.c0;
    case kBM15: {
      const float tf = static_cast<float>(freq);
      const float c1 = s.norm_const;
      return s.c0 - s.c0 / (1.f + tf / c1);
    }
    case kBM25Tiny: {
      const float tf = static_cast<float>(freq);
      const float inv_c1 = s.cache[read_norm(s, doc) & 0xFFu];
      return s.c0 - s.c0 / (1.f + tf * inv_c1);
    }
    case kBM25NoNorm: {
      const float tf = static_cast<float>(freq);
      const float inv_c1 = s.cache[1];
      return s.c0 - s.c0 / (1.f + tf * inv_c1);
    }
    case kBM25Wide: {
      const float tf = static_cast<float>(freq);
      const float c1 =
        s.norm_const + s.norm_length * static_cast<float>(read_norm(s, doc));
      return s.c0 - s.c0 * c1 / (c1 + tf);
    }
    case kTfidf:
      return std::sqrt(static_cast<float>(freq)) * s.c0;
    case kTfidfTiny:
    case kTfidfWide:
      return std::sqrt(static_cast<float>(freq)) * s.c0 *
             rsqrt_cached(read_norm(s, doc));
    case kBM25Legacy: {
      const float tf = std::sqrt(static_cast<float>(freq));  // kSQRT.get<true>, bm25.cpp:336
      const float c1 = s.norm_const + s.norm_length * (1.f / read_legacy_norm(s, doc));
      return s.c0 - s.c0 * c1 / (c1 + tf);
    }
    case kTfidfLegacy:
      return std::sqrt(static_cast<float>(freq)) * s.c0 * read_legacy_norm(s, doc);
  }
  return 0.f;
}

struct Sub {  // ScoreAdapter: iterator + its score function
  orc_doc_iterator it;
  TermScorer sc;
  uint32_t cost;  // term_meta::docs_count (formats_10.cpp:2264)
  inline float score() const { return score_posting(sc, it.freq, it.doc); }
};

// Global statistics of one query term: BM25::collect / TFIDF::collect output
struct TermStats {
  orc_bm25_stats bm25;
  float tfidf_idf;
};

Kind pick_kind(const orc_scorer& s, const orc_segment& seg) {
  if (s.kind == ORC_SCORER_BM25) {
    if (s.k == 0.f) return kBM1;   // IsBM1  bm25.cpp:447
    if (s.b == 0.f) return kBM15;  // IsBM15 bm25.cpp:451
    if (!seg.norms) return kBM25NoNorm;
    if (seg.norm_width == ORC_NORM_LEGACY_F32) return kBM25Legacy;  // :478-483
    return seg.norm_width == 1 ? kBM25Tiny : kBM25Wide;  // :466-476
  }
  if (!s.with_norms || !seg.norms) return kTfidf;
  if (seg.norm_width == ORC_NORM_LEGACY_F32) return kTfidfLegacy;
  return seg.norm_width == 1 ? kTfidfTiny : kTfidfWide;
}

void make_term_scorer(const orc_scorer& s, const orc_segment& seg,
                      const TermStats& st, float boost, TermScorer* out) {
  out->kind = pick_kind(s, seg);
  out->norms = seg.norms;
  out->width = seg.norm_width;
  out->cache = st.bm25.norm_cache;
  out->norm_const = st.bm25.norm_const;
  out->norm_length = st.bm25.norm_length;
  if (s.kind == ORC_SCORER_BM25) {
    out->c0 = boost * (s.k + 1) * st.bm25.idf;  // BM1Context ctor bm25.cpp:201
  } else {
    out->c0 = boost * st.tfidf_idf;  // TFIDFContext ctor tfidf.cpp:199
  }
}

// ---------------------------------------------------------------------------
// Iterators. `emit(doc, score)` plays the role of the harness loop body.

// SumMerger / MaxMerger / MinMerger — scorer.hpp:390-423.  `dst` is whatever the caller
// started from: the first sub-score in a conjunction (conjunction.hpp:105-126), 0 for a
// sub-iterator that is not on the doc in basic_disjunction (score_iterator_impl,
// disjunction.hpp:338-351) and the zeroed score buffer in block_disjunction (:1308-1351) —
// so a kMin disjunction scores 0 unless it has exactly two sub-iterators and both match
// ("probably can work strange with Max/MinMerger", scorer.hpp:387-388).
inline void merge_score(int merge, float& dst, float src) {
  if (merge == ORC_MERGE_MAX) {
    if (dst < src) dst = src;
  } else if (merge == ORC_MERGE_MIN) {
    if (src < dst) dst = src;
  } else {
    dst += src;
  }
}

// basic_disjunction — disjunction.hpp:233-253 (next), 302-310 (score),
// 338-351 (score_iterator_impl)
template<typename Emit>
void run_or2(Sub& lhs, Sub& rhs, int merge, Emit&& emit) {
  uint32_t doc = 0;
  auto next_impl = [&](Sub& s) {
    const uint32_t v = s.it.doc;
    if (doc == v) {
      orc_it_next(&s.it);
    } else if (v < doc) {
      orc_it_seek(&s.it, doc + uint32_t(doc != kEof));
    }
  };
  for (;;) {
    next_impl(lhs);
    next_impl(rhs);
    doc = std::min(lhs.it.doc, rhs.it.doc);
    if (doc == kEof) return;
    float res = lhs.it.doc == doc ? lhs.score() : 0.f;
    const float tmp = rhs.it.doc == doc ? rhs.score() : 0.f;
    merge_score(merge, res, tmp);  // merger(res, merger.temp()) :303
    emit(doc, res);
  }
}

// block_disjunction<kMatch, no readahead, 8 x 64> — disjunction.hpp:889-1369
// min_match > 1 selects the kMinMatch traits (min_match_iterator, :1378-1383):
// per-slot match counters (min_match_buffer :56-78), docs below min_match are
// skipped in next() (:963-970), buffers are reset on every refill round (:1255-1259).
template<typename Emit>
void run_block_or(std::vector<Sub>& itrs, int merge, Emit&& emit, uint32_t min_match = 1) {
  constexpr uint32_t kWindow = 512;  // kBlockSize * kNumBlocks :1087-1092
  uint64_t mask[8];
  float score_buf[kWindow];
  uint32_t match_count[kWindow];
  const bool mm = min_match > 1;
  uint32_t doc_base = 0;  // doc_limits::invalid()
  uint32_t min = 1;       // doc_limits::min()  :1358
  uint32_t max = 0;

  for (;;) {
    // refill() :1240-1306
    if (itrs.empty()) return;
    std::memset(mask, 0, sizeof mask);  // reset()
    std::memset(score_buf, 0, sizeof score_buf);
    if (mm) std::memset(match_count, 0, sizeof match_count);
    bool empty = true;
    bool first_round = true;
    do {
      if (mm && !first_round) {  // kMinMatch: reset() inside the loop
        std::memset(mask, 0, sizeof mask);
        std::memset(score_buf, 0, sizeof score_buf);
        std::memset(match_count, 0, sizeof match_count);
      }
      first_round = false;
      doc_base = min;
      max = min + kWindow;
      min = kEof;
      // visit_and_purge :1194-1216 with refill<Score> :1308-1351
      size_t i = 0;
      while (i < itrs.size()) {
        Sub& s = itrs[i];
        bool alive;
        if ((s.it.doc < doc_base && !orc_it_next(&s.it)) || s.it.doc == kEof) {
          alive = false;
        } else {
          for (;;) {
            const uint32_t value = s.it.doc;
            if (value >= max) {
              min = std::min(value, min);
              alive = true;
              break;
            }
            const uint32_t offset = value - doc_base;
            mask[offset / 64] |= uint64_t(1) << (offset % 64);
            merge_score(merge, score_buf[offset], s.score());  // :1334-1336
            if (mm) {
              empty &= (++match_count[offset] < min_match);  // match_buf_.inc :1341
            } else {
              empty = false;
            }
            if (!orc_it_next(&s.it)) {
              alive = false;
              break;
            }
          }
        }
        if (!alive) {
          std::swap(itrs[i], itrs.back());  // irstd::swap_remove std.hpp:62-66
          itrs.pop_back();
        } else {
          ++i;
        }
      }
      // kMinMatch: fewer iterators left than min_match -> nothing later can match (:1218-1226)
      if (mm && itrs.size() < min_match) itrs.clear();
    } while (empty && !itrs.empty());
    if (empty) return;
    // next() :939-986 — ascending set bits
    for (uint32_t w = 0; w < 8; ++w) {
      uint64_t cur = mask[w];
      while (cur) {
        const uint32_t off = uint32_t(__builtin_ctzll(cur));
        cur &= cur - 1;
        if (mm && match_count[w * 64 + off] < min_match) continue;  // next() :963-970
        emit(doc_base + w * 64 + off, score_buf[w * 64 + off]);
      }
    }
  }
}

// Conjunction — conjunction.hpp:191-223 (next/converge), 105-126 (Score2/N)
template<typename Emit>
void run_and(std::vector<Sub>& itrs, int merge, Emit&& emit) {
  Sub& front = itrs.front();
  for (;;) {
    if (!orc_it_next(&front.it)) return;
    uint32_t target = front.it.doc;
  restart:
    for (size_t i = 1; i < itrs.size(); ++i) {
      const uint32_t doc = orc_it_seek(&itrs[i].it, target);
      if (target < doc) {
        target = orc_it_seek(&front.it, doc);
        if (target != kEof) goto restart;
        return;
      }
    }
    float res = itrs[0].score();
    for (size_t i = 1; i < itrs.size(); ++i) merge_score(merge, res, itrs[i].score());
    emit(target, res);
  }
}

// SegmentReaderImpl::mask (core/index/segment_reader_impl.cpp:286-292) -> MaskDocIterator
// (:69-101): next() skips every doc the segment's DocumentMask contains.
struct DocMask {
  std::vector<uint64_t> bits;
  explicit DocMask(const orc_segment& seg) {
    if (!seg.doc_mask || !seg.doc_mask_count) return;   // docs_mask_.empty(): the iterator as it is
    bits.assign(size_t(seg.num_docs) / 64 + 2, 0);
    for (uint64_t i = 0; i < seg.doc_mask_count; ++i) {
      const uint32_t d = seg.doc_mask[i];
      if (d <= seg.num_docs) bits[d >> 6] |= 1ull << (d & 63u);
    }
  }
  bool contains(uint32_t doc) const {
    return !bits.empty() && (doc >> 6) < bits.size() && ((bits[doc >> 6] >> (doc & 63u)) & 1u);
  }
};

// filter->execute(segment) for by_term / Or / And
template<typename Emit>
void execute_segment_unmasked(const orc_segment& seg, const orc_term_meta* metas,
                              uint32_t n_terms, int32_t op, const orc_scorer& scorer,
                              const float* boosts, const TermStats* stats, Emit&& emit) {
  // op = ORC_OP_OR | ORC_OP_AND | ORC_OP_MINMATCH + (min_match << 8), + (ORC_MERGE_* << 24)
  const int merge = (op >> 24) & 3;   // boolean_filter::merge_type(), boolean_filter.hpp:39-43
  op &= 0xFFFFFF;
  uint32_t min_match = 1;
  if ((op & 0xFF) == ORC_OP_MINMATCH) {
    // MinMatchQuery::execute — boolean_query.cpp:212-247
    min_match = std::max<uint32_t>(1u, uint32_t(op) >> 8);
    if (min_match > n_terms) return;
    op = min_match == n_terms ? ORC_OP_AND : ORC_OP_OR;
  }
  std::vector<Sub> itrs;
  itrs.reserve(n_terms);
  for (uint32_t t = 0; t < n_terms; ++t) {
    // TermQuery::execute: no cookie for this segment -> empty iterator
    // (term_query.cpp:41-43); MakeScoreAdapters drops empties for OR and
    // collapses AND to empty (boolean_query.cpp:47-57)
    if (metas[t].docs_count == 0) {
      if (op == ORC_OP_AND) return;
      continue;
    }
    itrs.emplace_back();
    Sub& s = itrs.back();
    orc_it_prepare_wand(&s.it, seg.doc_file, seg.doc_file_len, seg.layout, &metas[t], 1,
                        seg.wand_count);
    make_term_scorer(scorer, seg, stats[t], boosts ? boosts[t] : 1.f, &s.sc);
    s.cost = metas[t].docs_count;
  }
  if (itrs.empty()) return;
  if (op == ORC_OP_OR && min_match > 1) {
    // MakeWeakDisjunction — disjunction.hpp:1469-1513
    if (min_match > itrs.size()) return;
    if (min_match == itrs.size()) {
      op = ORC_OP_AND;  // pure conjunction :1491-1494
    } else {
      run_block_or(itrs, merge, emit, min_match);
      return;
    }
  }
  if (itrs.size() == 1) {  // MakeDisjunction :1422-1426 / MakeConjunction :444
    Sub& s = itrs[0];
    while (orc_it_next(&s.it)) emit(s.it.doc, s.score());
    return;
  }
  if (op == ORC_OP_AND) {
    std::sort(itrs.begin(), itrs.end(),  // MakeConjunction :450-453
              [](const Sub& a, const Sub& b) { return a.cost < b.cost; });
    run_and(itrs, merge, emit);
  } else if (itrs.size() == 2) {
    run_or2(itrs[0], itrs[1], merge, emit);  // MakeDisjunction :1433-1440
  } else {
    run_block_or(itrs, merge, emit);  // :1465
  }
}

// segment.mask(filter->execute(segment))
template<typename Emit>
void execute_segment(const orc_segment& seg, const orc_term_meta* metas,
                     uint32_t n_terms, int32_t op, const orc_scorer& scorer,
                     const float* boosts, const TermStats* stats, Emit&& emit) {
  const DocMask mask(seg);
  execute_segment_unmasked(seg, metas, n_terms, op, scorer, boosts, stats,
                           [&](uint32_t doc, float score_value) {
                             if (!mask.contains(doc)) emit(doc, score_value);
                           });
}

// ---------------------------------------------------------------------------
// by_phrase (fixed offsets)

struct PSub {  // doc iterator + its position attribute + desired offset in the phrase
  orc_doc_iterator it;
  orc_pos_iterator pos;
  uint32_t offset;
  uint32_t cost;
};
inline bool pnext(PSub& s) {  // doc_iterator::next with IteratorTraits::position() :2106-2113
  if (!orc_it_next(&s.it)) return false;
  orc_pos_notify(&s.pos, s.it.freq);
  return true;
}
inline uint32_t pseek(PSub& s, uint32_t target) {
  while (s.it.doc < target) {
    if (!pnext(s)) break;
  }
  return s.it.doc;
}

// FixedPhraseFrequency<false, true>::NextPosition — phrase_iterator.hpp:109-151
uint32_t phrase_frequency(std::vector<PSub*>& pos) {
  uint32_t phrase_freq = 0;
  PSub& lead = *pos.front();
  orc_pos_next(&lead.pos, lead.it.freq);
  while (lead.pos.value != UINT32_MAX) {
    const uint32_t base_position = lead.pos.value;
    bool match = true;
    for (size_t i = 1; i < pos.size(); ++i) {
      PSub& p = *pos[i];
      const uint32_t term_position = base_position + p.offset;
      if (term_position == 0) return phrase_freq;  // !pos_limits::valid
      const uint32_t sought = orc_pos_seek(&p.pos, p.it.freq, term_position);
      if (sought == UINT32_MAX) return phrase_freq;  // exhausted
      if (sought != term_position) {  // sought too far from the lead
        match = false;
        orc_pos_seek(&lead.pos, lead.it.freq, sought - p.offset);
        break;
      }
    }
    if (match) {
      ++phrase_freq;
      orc_pos_next(&lead.pos, lead.it.freq);
    }
  }
  return phrase_freq;
}

// FixedPhraseQuery::execute (phrase_query.cpp:44-111) + PhraseIterator::next
// (phrase_iterator.hpp:590-596): conjunction ordered by cost, then EvaluateFreq
template<typename Emit>
void execute_phrase(const orc_segment& seg, const orc_term_meta* metas, uint32_t n_terms,
                    const uint32_t* offsets, Emit&& emit) {
  if (!seg.pos_file) return;
  const DocMask mask(seg);   // segment.mask(...): a deleted doc never leaves the iterator
  std::vector<PSub> subs(n_terms);
  for (uint32_t t = 0; t < n_terms; ++t) {
    if (metas[t].docs_count == 0) return;  // phrase state absent for the segment
    PSub& s = subs[t];
    orc_it_prepare_wand(&s.it, seg.doc_file, seg.doc_file_len, seg.layout, &metas[t], 1,
                        seg.wand_count);
    orc_pos_prepare(&s.pos, seg.pos_file, seg.pos_file_len, seg.layout, &metas[t],
                    seg.pos_one_based);
    s.offset = offsets[t];
    s.cost = metas[t].docs_count;
  }
  std::vector<PSub*> phrase(n_terms), conj(n_terms);
  for (uint32_t t = 0; t < n_terms; ++t) phrase[t] = conj[t] = &subs[t];
  std::stable_sort(conj.begin(), conj.end(),
                   [](const PSub* a, const PSub* b) { return a->cost < b->cost; });
  PSub& front = *conj.front();
  for (;;) {  // Conjunction::next/converge — conjunction.hpp:191-223
    if (!pnext(front)) return;
    uint32_t target = front.it.doc;
  restart:
    for (size_t i = 1; i < conj.size(); ++i) {
      const uint32_t doc = pseek(*conj[i], target);
      if (target < doc) {
        target = pseek(front, doc);
        if (target != kEof) goto restart;
        return;
      }
    }
    const uint32_t pf = phrase_frequency(phrase);
    if (pf && !mask.contains(target)) emit(target, pf);
  }
}

void collect_stats(const orc_scorer& scorer, uint64_t dwf, uint64_t dwt,
                   uint64_t ttf, TermStats* st) {
  std::memset(st, 0, sizeof *st);  // stats are zero-initialised scorer.hpp:143
  if (scorer.kind == ORC_SCORER_BM25) {
    orc_bm25_collect(scorer.k, scorer.b, dwf, dwt, ttf, &st->bm25);
  } else {
    st->tfidf_idf = orc_tfidf_idf(dwf, dwt);
  }
}

struct ByScoreDesc {  // index-search.cpp:733-735 comparator (min-heap on score)
  bool operator()(const orc_hit& l, const orc_hit& r) const noexcept {
    return l.score > r.score;
  }
};

int64_t search_one(const orc_segment* segs, uint32_t nsegs,
                   const orc_term_meta* metas, uint32_t n_terms, int32_t op,
                   const orc_scorer& scorer, const float* boosts,
                   const uint64_t* dwf_seg, const uint64_t* ttf_seg, uint32_t k,
                   orc_hit* out, uint64_t* hits_total) {
  // by_term::prepare — statistics over ALL segments (term_filter.cpp:102-125)
  uint64_t dwf = 0, ttf = 0;
  for (uint32_t s = 0; s < nsegs; ++s) {
    dwf += dwf_seg[s];
    ttf += ttf_seg[s];
  }
  std::vector<TermStats> stats(n_terms);
  for (uint32_t t = 0; t < n_terms; ++t) {
    uint64_t dwt = 0;
    for (uint32_t s = 0; s < nsegs; ++s)
      dwt += metas[size_t(s) * n_terms + t].docs_count;
    collect_stats(scorer, dwf, dwt, ttf, &stats[t]);
  }

  // index-search.cpp:719-787
  std::vector<orc_hit> sorted;
  sorted.reserve(k);
  uint64_t doc_count = 0;
  uint32_t left = k;
  for (uint32_t s = 0; s < nsegs; ++s) {
    execute_segment(
      segs[s], metas + size_t(s) * n_terms, n_terms, op, scorer, boosts,
      stats.data(), [&](uint32_t doc, float score_value) {
        ++doc_count;
        if (left) {
          sorted.push_back(orc_hit{score_value, doc, s});
          if (0 == --left) {
            std::make_heap(sorted.begin(), sorted.end(), ByScoreDesc{});
          }
        } else if (k && sorted.front().score < score_value) {
          std::pop_heap(sorted.begin(), sorted.end(), ByScoreDesc{});
          sorted.back() = orc_hit{score_value, doc, s};
          std::push_heap(sorted.begin(), sorted.end(), ByScoreDesc{});
        }
      });
  }
  std::sort(sorted.begin(), sorted.end(), ByScoreDesc{});
  if (!sorted.empty())
    std::memcpy(out, sorted.data(), sorted.size() * sizeof(orc_hit));
  if (hits_total) *hits_total = doc_count;
  return int64_t(sorted.size());
}

}  // namespace

extern "C" {

// BM25::collect — bm25.cpp:366-410
void orc_bm25_collect(float k, float b, uint64_t docs_with_field,
                      uint64_t docs_with_term, uint64_t total_term_freq,
                      orc_bm25_stats* stats) {
  stats->idf += float(
    std::log1p((static_cast<double>(docs_with_field - docs_with_term) + 0.5) /
               (static_cast<double>(docs_with_term) + 0.5)));
  const bool needs_norm = !(k == 0.f) && !(b == 0.f);  // bm25.hpp:105
  if (!needs_norm) {
    stats->norm_const = k;
    return;
  }
  const float kb = k * b;
  stats->norm_const = k - kb;
  if (total_term_freq && docs_with_field) {
    const float avg_dl = static_cast<float>(total_term_freq) /
                         static_cast<float>(docs_with_field);
    stats->norm_length = kb / avg_dl;
  } else {
    stats->norm_length = kb;
  }
  stats->norm_cache[0] = 0.f;
  float i = 1.f;
  for (uint32_t n = 1; n < 256; ++n) {
    stats->norm_cache[n] = 1.f / (stats->norm_const + stats->norm_length * i);
    i += 1.f;
  }
}

// TFIDF::collect — tfidf.cpp:263-278
float orc_tfidf_idf(uint64_t docs_with_field, uint64_t docs_with_term) {
  return static_cast<float>(
    std::log1p((double(docs_with_field) + 1.0) / (double(docs_with_term) + 1.0)));
}

int64_t orc_search(const orc_segment* segs, uint32_t nsegs,
                   const orc_term_meta* metas, uint32_t n_terms, int32_t op,
                   const orc_scorer* scorer, const float* boosts,
                   const uint64_t* docs_with_field,
                   const uint64_t* total_term_freq, uint32_t k, orc_hit* out,
                   uint64_t* hits_total) {
  if (!segs || !metas || !scorer || !n_terms) return -1;
  return search_one(segs, nsegs, metas, n_terms, op, *scorer, boosts,
                    docs_with_field, total_term_freq, k, out, hits_total);
}

int64_t orc_search_batch(const orc_segment* segs, uint32_t nsegs,
                         const orc_term_meta* metas, uint32_t n_queries,
                         uint32_t n_terms, int32_t op, const orc_scorer* scorer,
                         const uint64_t* docs_with_field,
                         const uint64_t* total_term_freq, uint32_t k,
                         uint32_t threads, orc_hit* out, uint32_t* counts,
                         uint64_t* hits_total) {
  if (!segs || !metas || !scorer || !n_terms) return -1;
  if (threads == 0) threads = 1;
  std::atomic<uint32_t> next{0};  // task_provider.pop() index-search.cpp:689
  std::atomic<int> failed{0};
  auto worker = [&] {
    for (;;) {
      const uint32_t q = next.fetch_add(1);
      if (q >= n_queries) return;
      uint64_t hits = 0;
      const int64_t n = search_one(
        segs, nsegs, metas + size_t(q) * nsegs * n_terms, n_terms, op, *scorer,
        nullptr, docs_with_field, total_term_freq, k, out + size_t(q) * k, &hits);
      if (n < 0) {
        failed = 1;
        return;
      }
      counts[q] = uint32_t(n);
      if (hits_total) hits_total[q] = hits;
    }
  };
  std::vector<std::thread> pool;
  for (uint32_t t = 1; t < threads; ++t) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
  return failed ? -1 : int64_t(n_queries);
}

namespace {
// term_stats.finish(stats_buf, term_idx, ...) for every phrase term on ONE buffer
// (phrase_filter.cpp:281-287): BM25::collect / TFIDF::collect accumulate idf
void phrase_stats(const orc_scorer& scorer, uint64_t dwf, const uint64_t* dwt,
                  uint32_t n_terms, uint64_t ttf, TermStats* st) {
  std::memset(st, 0, sizeof *st);
  for (uint32_t t = 0; t < n_terms; ++t) {
    if (scorer.kind == ORC_SCORER_BM25)
      orc_bm25_collect(scorer.k, scorer.b, dwf, dwt[t], ttf, &st->bm25);
    else
      st->tfidf_idf += orc_tfidf_idf(dwf, dwt[t]);
  }
}
}  // namespace

int64_t orc_search_phrase(const orc_segment* segs, uint32_t nsegs,
                          const orc_term_meta* metas, uint32_t n_terms,
                          const uint32_t* offsets, const orc_scorer* scorer, float boost,
                          const uint64_t* docs_with_field, const uint64_t* total_term_freq,
                          uint32_t k, orc_hit* out, uint64_t* hits_total) {
  if (!segs || !metas || !scorer || !n_terms || !offsets || offsets[0] != 0) return -1;
  uint64_t dwf = 0, ttf = 0;
  for (uint32_t s = 0; s < nsegs; ++s) {
    dwf += docs_with_field[s];
    ttf += total_term_freq[s];
  }
  std::vector<uint64_t> dwt(n_terms, 0);
  for (uint32_t t = 0; t < n_terms; ++t)
    for (uint32_t s = 0; s < nsegs; ++s) dwt[t] += metas[size_t(s) * n_terms + t].docs_count;
  TermStats st;
  phrase_stats(*scorer, dwf, dwt.data(), n_terms, ttf, &st);

  std::vector<orc_hit> sorted;  // index-search.cpp:719-787
  sorted.reserve(k);
  uint64_t doc_count = 0;
  uint32_t left = k;
  for (uint32_t s = 0; s < nsegs; ++s) {
    TermScorer sc;
    make_term_scorer(*scorer, segs[s], st, boost, &sc);
    execute_phrase(segs[s], metas + size_t(s) * n_terms, n_terms, offsets,
                   [&](uint32_t doc, uint32_t pf) {
                     const float score_value = score_posting(sc, pf, doc);
                     ++doc_count;
                     if (left) {
                       sorted.push_back(orc_hit{score_value, doc, s});
                       if (0 == --left) std::make_heap(sorted.begin(), sorted.end(), ByScoreDesc{});
                     } else if (k && sorted.front().score < score_value) {
                       std::pop_heap(sorted.begin(), sorted.end(), ByScoreDesc{});
                       sorted.back() = orc_hit{score_value, doc, s};
                       std::push_heap(sorted.begin(), sorted.end(), ByScoreDesc{});
                     }
                   });
  }
  std::sort(sorted.begin(), sorted.end(), ByScoreDesc{});
  if (!sorted.empty()) std::memcpy(out, sorted.data(), sorted.size() * sizeof(orc_hit));
  if (hits_total) *hits_total = doc_count;
  return int64_t(sorted.size());
}

int64_t orc_score_all_phrase(const orc_segment* seg, const orc_term_meta* metas,
                             uint32_t n_terms, const uint32_t* offsets,
                             const orc_scorer* scorer, float boost, uint64_t docs_with_field,
                             const uint64_t* docs_with_term, uint64_t total_term_freq,
                             float* scores, uint32_t* phrase_freq) {
  if (!seg || !metas || !scorer || !n_terms || !offsets || offsets[0] != 0) return -1;
  TermStats st;
  phrase_stats(*scorer, docs_with_field, docs_with_term, n_terms, total_term_freq, &st);
  TermScorer sc;
  make_term_scorer(*scorer, *seg, st, boost, &sc);
  std::memset(scores, 0, sizeof(float) * (size_t(seg->num_docs) + 1));
  std::memset(phrase_freq, 0, sizeof(uint32_t) * (size_t(seg->num_docs) + 1));
  int64_t n = 0;
  execute_phrase(*seg, metas, n_terms, offsets, [&](uint32_t doc, uint32_t pf) {
    if (doc <= seg->num_docs) {
      scores[doc] = score_posting(sc, pf, doc);
      phrase_freq[doc] = pf;
      ++n;
    }
  });
  return n;
}

int64_t orc_score_all(const orc_segment* seg, const orc_term_meta* metas,
                      uint32_t n_terms, int32_t op, const orc_scorer* scorer,
                      const float* boosts, uint64_t docs_with_field,
                      const uint64_t* docs_with_term, uint64_t total_term_freq,
                      float* scores, uint8_t* matched) {
  if (!seg || !metas || !scorer || !n_terms) return -1;
  std::vector<TermStats> stats(n_terms);
  for (uint32_t t = 0; t < n_terms; ++t)
    collect_stats(*scorer, docs_with_field, docs_with_term[t], total_term_freq,
                  &stats[t]);
  std::memset(scores, 0, sizeof(float) * (size_t(seg->num_docs) + 1));
  std::memset(matched, 0, size_t(seg->num_docs) + 1);
  int64_t n = 0;
  execute_segment(*seg, metas, n_terms, op, *scorer, boosts, stats.data(),
                  [&](uint32_t doc, float s) {
                    if (doc <= seg->num_docs) {
                      scores[doc] = s;
                      matched[doc] = 1;
                      ++n;
                    }
                  });
  return n;
}

}  // extern "C"
