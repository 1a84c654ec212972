// irs_hip.hpp — C++ host side above the C ABI (include/irs_hip.h), header only.
//
// Mirrors, for the hot path only, the interfaces an IResearch caller uses — same names,
// argument meaning and error behaviour — so that code (and tests) written against the
// reference read the same here:
//   irs::BM25 / irs::TFIDF ::collect           core/search/bm25.cpp:366-410, tfidf.cpp:263-278
//   irs::by_term / irs::Or / irs::And          core/search/term_filter.hpp, boolean_filter.hpp
//   irs::by_phrase (plain terms)               core/search/phrase_filter.hpp:50-110
//   filter::prepare  -> filter::prepared       core/search/term_filter.cpp:92-129,
//                                              phrase_filter.cpp:212-293 (statistics over ALL segments)
//   prepared::execute(segment) + harness loop  core/search/filter.hpp:52-78, utils/index-search.cpp:719-787
//   exceptions instead of status codes         io_error / index_error / illegal_argument
//                                              (core/error/error.hpp), as formats_10.cpp:3410-3415
// Nothing per posting happens here: every posting is decoded and scored by libirs_hip.so on
// the GPU.  The Python module iresearch_amd/search.py is the same layer for the test and
// bench plumbing; both produce bit-identical scorer parameters.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <utility>
#include <string>
#include <variant>
#include <vector>

#include "irs_hip.h"

namespace irs_hip_host {

// ---- errors: the reference throws, the C ABI returns codes ---------------------------------
struct error : std::runtime_error {
  int status;
  error(int s, const std::string& what) : std::runtime_error(what), status(s) {}
};
struct illegal_argument : error { using error::error; };   // IRS_HIP_EINVAL
struct index_error : error { using error::error; };        // IRS_HIP_ECORRUPT
struct io_error : error { using error::error; };           // IRS_HIP_EHIP / ENOMEM / EOVERFLOW / EPEER
struct not_supported : error { using error::error; };      // IRS_HIP_EUNSUPPORTED

inline void check(int rc, const char* what) {
  if (rc == IRS_HIP_OK) return;
  const std::string msg = std::string(what) + ": " + irs_hip_strerror(rc);
  switch (rc) {
    case IRS_HIP_EINVAL: throw illegal_argument(rc, msg);
    case IRS_HIP_ECORRUPT: throw index_error(rc, msg);
    case IRS_HIP_EUNSUPPORTED: throw not_supported(rc, msg);
    default: throw io_error(rc, msg);
  }
}

// ---- scorers ---------------------------------------------------------------------------------
// The part of BM25Stats / TFIDF's idf that crosses the ABI (bm25.hpp:48-57).  `collect`
// ACCUMULATES idf like the reference (`stats.idf += ...`, bm25.cpp:381-383): a phrase finishes
// all of its terms into one blob (phrase_filter.cpp:281-287).
struct TermStats {
  float idf = 0.f;
  float norm_const = 0.f;
  float norm_length = 0.f;
};

class BM25 {
 public:
  explicit BM25(float k = 1.2f, float b = 0.75f) noexcept : k_{k}, b_{b} {}
  float k() const noexcept { return k_; }
  float b() const noexcept { return b_; }
  bool IsBM1() const noexcept { return k_ == 0.f; }                // bm25.hpp:101
  bool IsBM15() const noexcept { return !IsBM1() && b_ == 0.f; }   // bm25.hpp:103

  void collect(TermStats& stats, uint64_t docs_with_field, uint64_t docs_with_term,
               uint64_t total_term_freq) const {
    stats.idf += static_cast<float>(
      std::log1p((static_cast<double>(docs_with_field - docs_with_term) + 0.5) /
                 (static_cast<double>(docs_with_term) + 0.5)));
    if (k_ == 0.f || b_ == 0.f) {  // !NeedsNorm(), bm25.cpp:387-390
      stats.norm_const = k_;
      return;
    }
    const float kb = k_ * b_;
    stats.norm_const = k_ - kb;
    if (total_term_freq && docs_with_field) {
      const float avg_dl =
        static_cast<float>(total_term_freq) / static_cast<float>(docs_with_field);
      stats.norm_length = kb / avg_dl;
    } else {
      stats.norm_length = kb;
    }
  }
  irs_hip_term_scorer term_scorer(const TermStats& st, float boost) const noexcept {
    irs_hip_term_scorer t{};
    t.kind = IsBM1() ? IRS_HIP_SCORE_BM1 : IsBM15() ? IRS_HIP_SCORE_BM15 : IRS_HIP_SCORE_BM25;
    t.c0 = boost * (k_ + 1.f) * st.idf;  // BM1Context, bm25.cpp:201
    t.norm_const = st.norm_const;
    t.norm_length = st.norm_length;
    return t;
  }

 private:
  float k_, b_;
};

class TFIDF {
 public:
  explicit TFIDF(bool normalize = false) noexcept : normalize_{normalize} {}
  bool normalize() const noexcept { return normalize_; }
  void collect(TermStats& stats, uint64_t docs_with_field, uint64_t docs_with_term,
               uint64_t /*total_term_freq*/) const {
    stats.idf += static_cast<float>(std::log1p((static_cast<double>(docs_with_field) + 1.0) /
                                               (static_cast<double>(docs_with_term) + 1.0)));
  }
  irs_hip_term_scorer term_scorer(const TermStats& st, float boost) const noexcept {
    irs_hip_term_scorer t{};
    t.kind = normalize_ ? IRS_HIP_SCORE_TFIDF_NORM : IRS_HIP_SCORE_TFIDF;
    t.c0 = boost * st.idf;  // TFIDFContext, tfidf.cpp:199
    return t;
  }

 private:
  bool normalize_;
};

// ---- filters ---------------------------------------------------------------------------------
// Terms are ordinals of the segment's staged term table (the term dictionary walk is the
// adapter's business: INTEGRATION.md).
struct by_term {
  uint32_t term = IRS_HIP_NO_TERM;
  float boost = 1.f;
};
struct Or {
  std::vector<by_term> subs;
  uint32_t min_match_count = 1;  // irs::Or::min_match_count()
  irs_hip_merge merge_type = IRS_HIP_MERGE_SUM;  // boolean_filter::merge_type()
};
struct And {
  std::vector<by_term> subs;
  irs_hip_merge merge_type = IRS_HIP_MERGE_SUM;
};
struct by_phrase {
  std::vector<uint32_t> terms;
  std::vector<uint32_t> offsets;  // relative to the first term; empty = consecutive words
  float boost = 1.f;
  // by_phrase_options::push_back<by_term_options>(offs): `offs` positions after the end
  by_phrase& push_back(uint32_t term, uint32_t offs = 0) {
    const uint32_t next = offsets.empty() ? 0u : offsets.back() + 1u;
    terms.push_back(term);
    offsets.push_back(next + offs);
    return *this;
  }
};
using filter = std::variant<by_term, Or, And, by_phrase>;

// What by_term::prepare reads from one segment without touching postings.
struct SegmentStats {
  uint64_t docs_with_field = 0;
  uint64_t total_term_freq = 0;
  const irs_hip_term_meta* terms = nullptr;
  uint32_t num_terms = 0;
  uint64_t docs_count(uint32_t term) const noexcept {
    return term < num_terms ? terms[term].docs_count : 0;
  }
};

// filter::prepared: the op plus the (term, scorer values) entries of one query.
struct PreparedQuery {
  int32_t op = IRS_HIP_OP_OR;
  uint32_t min_match = 0;
  uint32_t merge = IRS_HIP_MERGE_SUM;
  std::vector<irs_hip_term_scorer> terms;  // .term = the ordinal; same for every segment here ...
  // ... unless segment_terms[s][i] names slot i's ordinal in segment s (IRS_HIP_NO_TERM: the
  // segment has no state for it) — what a scored multi-term filter prepares (prepare_expansion)
  std::vector<std::vector<uint32_t>> segment_terms;
};

// filter::prepare for a list of filters against ALL segments (statistics are index-global:
// D = sum docs_with_field, d = sum docs_count of the term, avgdl from the summed frequency).
template<typename Scorer>
std::vector<PreparedQuery> prepare(const std::vector<filter>& filters, const Scorer& scorer,
                                   const std::vector<SegmentStats>& index) {
  uint64_t dwf = 0, ttf = 0;
  for (const auto& s : index) {
    dwf += s.docs_with_field;
    ttf += s.total_term_freq;
  }
  auto docs_with_term = [&](uint32_t term) {
    uint64_t d = 0;
    for (const auto& s : index) d += s.docs_count(term);
    return d;
  };
  auto one = [&](const by_term& t) {
    TermStats st;
    scorer.collect(st, dwf, docs_with_term(t.term), ttf);
    irs_hip_term_scorer e = scorer.term_scorer(st, t.boost);
    e.term = t.term;
    return e;
  };
  std::vector<PreparedQuery> out;
  out.reserve(filters.size());
  for (const filter& f : filters) {
    PreparedQuery q;
    if (const auto* t = std::get_if<by_term>(&f)) {
      q.op = IRS_HIP_OP_OR;
      q.terms.push_back(one(*t));
    } else if (const auto* o = std::get_if<Or>(&f)) {
      q.op = o->min_match_count > 1 ? IRS_HIP_OP_MINMATCH : IRS_HIP_OP_OR;
      q.min_match = o->min_match_count > 1 ? o->min_match_count : 0;
      q.merge = o->merge_type;
      for (const auto& t : o->subs) q.terms.push_back(one(t));
    } else if (const auto* a = std::get_if<And>(&f)) {
      q.op = IRS_HIP_OP_AND;
      q.merge = a->merge_type;
      for (const auto& t : a->subs) q.terms.push_back(one(t));
    } else {
      const auto& p = std::get<by_phrase>(f);
      if (!p.offsets.empty() && (p.offsets.size() != p.terms.size() || p.offsets[0] != 0))
        throw illegal_argument(IRS_HIP_EINVAL, "by_phrase: offsets are relative to the first term");
      q.op = IRS_HIP_OP_PHRASE;
      TermStats st;  // ONE blob for the phrase (FixedPrepareCollect)
      for (uint32_t t : p.terms) scorer.collect(st, dwf, docs_with_term(t), ttf);
      irs_hip_term_scorer e = scorer.term_scorer(st, p.boost);
      for (size_t i = 0; i < p.terms.size(); ++i) {
        e.term = p.terms[i];
        e.phrase_offset = p.offsets.empty() ? uint32_t(i) : p.offsets[i];
        q.terms.push_back(e);
      }
    }
    if (q.terms.empty()) throw illegal_argument(IRS_HIP_EINVAL, "empty filter");
    out.push_back(std::move(q));
  }
  return out;
}

// ---- one segment on one GPU ------------------------------------------------------------------
// irs::SubReader + postings_reader of one segment (postings_reader::prepare ... CountMappedMemory).
class SegmentReader {
 public:
  explicit SegmentReader(const irs_hip_segment_desc& desc) : num_terms_{desc.num_terms} {
    check(irs_hip_segment_open(&desc, &h_), "irs_hip_segment_open");
  }
  SegmentReader(const SegmentReader&) = delete;
  SegmentReader& operator=(const SegmentReader&) = delete;
  ~SegmentReader() { irs_hip_segment_close(h_); }
  irs_hip_segment* handle() const noexcept { return h_; }
  uint32_t num_terms() const noexcept { return num_terms_; }
  uint64_t CountMappedMemory() const noexcept { return irs_hip_segment_device_bytes(h_); }
  // SubReader::live_docs_count: docs that are not in the segment's DocumentMask
  uint64_t live_docs_count() const noexcept { return irs_hip_segment_live_docs(h_); }

  // postings_reader::iterator(...) drained: (docs, freqs) of one term
  void postings(uint32_t term, std::vector<uint32_t>& docs, std::vector<uint32_t>* freqs,
                uint32_t docs_count) const {
    docs.resize(docs_count ? docs_count : 1);
    if (freqs) freqs->resize(docs.size());
    uint32_t n = 0;
    check(irs_hip_decode_term(h_, term, docs.data(), freqs ? freqs->data() : nullptr,
                              uint32_t(docs.size()), &n),
          "irs_hip_decode_term");
    docs.resize(n);
    if (freqs) freqs->resize(n);
  }
  void positions(uint32_t term, std::vector<uint32_t>& out, uint64_t total_freq) const {
    out.resize(total_freq ? total_freq : 1);
    uint64_t n = 0;
    check(irs_hip_decode_positions(h_, term, out.data(), out.size(), &n),
          "irs_hip_decode_positions");
    out.resize(n);
  }
  // postings_reader::bit_union: returns the sum of docs_count like the reference
  uint64_t bit_union(const std::vector<uint32_t>& terms, std::vector<uint64_t>& set) const {
    uint64_t count = 0;
    check(irs_hip_bit_union(h_, terms.data(), uint32_t(terms.size()), set.data(), set.size(),
                            &count),
          "irs_hip_bit_union");
    return count;
  }
  // the population of the union of each term set's postings, all sets in one call; only the
  // counts come back (irs_hip_bit_union_counts: the `hits=` of multi-term filters)
  std::vector<uint64_t> bit_union_counts(const std::vector<std::vector<uint32_t>>& sets) const {
    std::vector<uint32_t> terms, offsets{0};
    for (const auto& set : sets) {
      terms.insert(terms.end(), set.begin(), set.end());
      offsets.push_back(uint32_t(terms.size()));
    }
    std::vector<uint64_t> counts(sets.size());
    if (!sets.empty())
      check(irs_hip_bit_union_counts(h_, terms.data(), offsets.data(), uint32_t(sets.size()),
                                     counts.data()),
            "irs_hip_bit_union_counts");
    return counts;
  }

 private:
  irs_hip_segment* h_ = nullptr;
  uint32_t num_terms_ = 0;
};

// ---- a batch of prepared queries on one or several segments of one device -------------------
class QueryBatch {
 public:
  struct Results {
    uint32_t n_segments = 0, n_queries = 0, k = 0;
    std::vector<irs_hip_hit> hits;        // [segment][query][k]
    std::vector<uint32_t> counts;         // [segment][query]
    std::vector<uint64_t> total_hits;     // [segment][query]: index-search `hits=`
    const irs_hip_hit* of(uint32_t seg, uint32_t q) const {
      return hits.data() + (size_t(seg) * n_queries + q) * k;
    }
    uint32_t count(uint32_t seg, uint32_t q) const { return counts[size_t(seg) * n_queries + q]; }
    uint64_t total(uint32_t seg, uint32_t q) const {
      return total_hits[size_t(seg) * n_queries + q];
    }
  };

  // A term a segment lacks (ordinal beyond its table) becomes IRS_HIP_NO_TERM there
  // (TermQuery::execute: no state for the segment -> empty iterator, term_query.cpp:41-43).
  // The device runs phrase queries and boolean queries as separate batches (different
  // kernels); a caller mixes them freely, as with the reference: they are split here and the
  // results stitched back in the caller's order.
  QueryBatch(const std::vector<const SegmentReader*>& segments,
             const std::vector<PreparedQuery>& prepared, uint32_t k)
    : n_segments_{uint32_t(segments.size())}, n_queries_{uint32_t(prepared.size())}, k_{k} {
    for (uint32_t q = 0; q < n_queries_; ++q)
      part_[prepared[q].op == IRS_HIP_OP_PHRASE ? 1 : 0].index.push_back(q);
    try {
      for (Part& part : part_) {
        if (part.index.empty()) continue;
        std::vector<irs_hip_query> queries;
        std::vector<irs_hip_term_scorer> entries;
        for (uint32_t q : part.index) {
          const PreparedQuery& p = prepared[q];
          queries.push_back(irs_hip_query{p.op, uint32_t(p.terms.size()),
                                          uint32_t(entries.size()), k, p.min_match, p.merge});
          entries.insert(entries.end(), p.terms.begin(), p.terms.end());
        }
        std::vector<irs_hip_term_scorer> all;
        std::vector<irs_hip_segment*> handles;
        for (size_t si = 0; si < segments.size(); ++si) {
          const SegmentReader* s = segments[si];
          handles.push_back(s->handle());
          const size_t base = all.size();
          for (irs_hip_term_scorer e : entries) {
            if (e.term >= s->num_terms()) e.term = IRS_HIP_NO_TERM;
            all.push_back(e);
          }
          for (size_t i = 0; i < part.index.size(); ++i) {
            const PreparedQuery& p = prepared[part.index[i]];
            if (p.segment_terms.empty()) continue;
            if (p.segment_terms.size() != segments.size() ||
                p.segment_terms[si].size() != p.terms.size())
              throw illegal_argument(IRS_HIP_EINVAL, "segment_terms: [segment][term slot]");
            for (size_t j = 0; j < p.terms.size(); ++j)
              all[base + queries[i].first_term + j].term = p.segment_terms[si][j];
          }
        }
        check(irs_hip_batch_create_multi(handles.data(), n_segments_, queries.data(),
                                         uint32_t(queries.size()), all.data(),
                                         uint32_t(entries.size()), &part.h),
              "irs_hip_batch_create_multi");
      }
    } catch (...) {
      for (Part& part : part_) irs_hip_batch_destroy(part.h);
      throw;
    }
  }
  QueryBatch(const QueryBatch&) = delete;
  QueryBatch& operator=(const QueryBatch&) = delete;
  ~QueryBatch() {
    for (Part& part : part_) irs_hip_batch_destroy(part.h);
  }

  // ExecutionContext::wand (index-search --search-mode wand): block-max pruning; before run()
  QueryBatch& set_wand(bool enable) {
    for (Part& part : part_)
      if (part.h) check(irs_hip_batch_set_wand(part.h, enable ? 1 : 0), "irs_hip_batch_set_wand");
    return *this;
  }
  // the per-segment lists will be MERGED (search(), search_sharded()): one threshold per query
  // for all of its segments here (irs_hip_batch_set_shared_threshold); before run()
  QueryBatch& set_shared_threshold(bool enable) {
    for (Part& part : part_)
      if (part.h)
        check(irs_hip_batch_set_shared_threshold(part.h, enable ? 1 : 0),
              "irs_hip_batch_set_shared_threshold");
    return *this;
  }
  // ... and one threshold per query across RANKS (irs_hip_batch_set_comm): every rank attaches its
  // communicator to its batch of the same queries; nullptr detaches
  QueryBatch& set_comm(irs_hip_comm* comm) {
    for (Part& part : part_)
      if (part.h) check(irs_hip_batch_set_comm(part.h, comm), "irs_hip_batch_set_comm");
    return *this;
  }
  // irs::score::Min per query: the k-th best score the caller's heap holds so far (empty: none)
  QueryBatch& set_min_scores(const std::vector<float>& min_scores) {
    for (Part& part : part_) {
      if (!part.h) continue;
      std::vector<float> m(part.index.size());
      for (size_t i = 0; i < part.index.size(); ++i)
        m[i] = min_scores.empty() ? 0.f : min_scores.at(part.index[i]);
      check(irs_hip_batch_set_min_scores(part.h, min_scores.empty() ? nullptr : m.data()),
            "irs_hip_batch_set_min_scores");
    }
    return *this;
  }
  // queue the planning stage of the next run() on `stream` (irs_hip_batch_plan)
  QueryBatch& plan(void* stream = nullptr) {
    for (Part& part : part_)
      if (part.h) check(irs_hip_batch_plan(part.h, stream), "irs_hip_batch_plan");
    return *this;
  }
  // run() hands the host half of a run to the library's worker thread and returns at once (a
  // failure then surfaces from the next call on the batch, and NOTHING of the run may be on
  // `stream` yet: the caller's own work on the stream is ordered behind the run only through a
  // call on the batch — results(), host_results(), device results).  set_async(false) keeps the
  // run on the caller's thread: launch-time semantics of a plain kernel launch, status from run()
  QueryBatch& set_async(bool enable) {
    for (Part& part : part_)
      if (part.h) check(irs_hip_batch_set_async(part.h, enable ? 1 : 0), "irs_hip_batch_set_async");
    return *this;
  }
  // joined plain disjunctions on paired doc tiles (default) or on 32-bit tiles
  // (irs_hip_batch_set_paired_tiles: the results are bit-identical, a tuning / test knob)
  QueryBatch& set_paired_tiles(bool enable) {
    for (Part& part : part_)
      if (part.h)
        check(irs_hip_batch_set_paired_tiles(part.h, enable ? 1 : 0), "irs_hip_batch_set_paired_tiles");
    return *this;
  }
  bool paired_tiles() {   // whether the last run took them (any part)
    bool any = false;
    for (Part& part : part_) {
      int used = 0;
      if (part.h) check(irs_hip_batch_paired_tiles(part.h, &used), "irs_hip_batch_paired_tiles");
      any = any || used != 0;
    }
    return any;
  }
  QueryBatch& run(void* stream = nullptr) {
    for (Part& part : part_)
      if (part.h) check(irs_hip_batch_run(part.h, stream), "irs_hip_batch_run");
    return *this;
  }
  Results results() {
    Results r;
    r.n_segments = n_segments_;
    r.n_queries = n_queries_;
    r.k = k_;
    const size_t units = size_t(n_segments_) * n_queries_;
    r.hits.resize(units * k_);
    r.counts.resize(units);
    r.total_hits.resize(units);
    for (Part& part : part_) {
      if (!part.h) continue;
      const size_t nq = part.index.size(), pu = size_t(n_segments_) * nq;
      std::vector<irs_hip_hit> hits(pu * k_);
      std::vector<uint32_t> counts(pu);
      std::vector<uint64_t> totals(pu);
      check(irs_hip_batch_results(part.h, hits.data(), k_, counts.data(), totals.data()),
            "irs_hip_batch_results");
      for (uint32_t s = 0; s < n_segments_; ++s)
        for (size_t i = 0; i < nq; ++i) {
          const size_t from = s * nq + i, to = size_t(s) * n_queries_ + part.index[i];
          std::copy_n(hits.begin() + from * k_, counts[from], r.hits.begin() + to * k_);
          r.counts[to] = counts[from];
          r.total_hits[to] = totals[from];
        }
    }
    return r;
  }
  // The serving-loop form of results(): results_to_host() verifies the run and queues the copy
  // to page-locked memory of the batch behind ITS kernels only (irs_hip_batch_results_to_host:
  // the hits of this batch cross PCIe while the next batch executes); host_results() waits for
  // that copy and stitches the caller's order back together.
  QueryBatch& results_to_host(void* stream = nullptr) {
    for (Part& part : part_)
      if (part.h) check(irs_hip_batch_results_to_host(part.h, stream), "irs_hip_batch_results_to_host");
    return *this;
  }
  Results host_results() {
    Results r;
    r.n_segments = n_segments_;
    r.n_queries = n_queries_;
    r.k = k_;
    const size_t units = size_t(n_segments_) * n_queries_;
    r.hits.resize(units * k_);
    r.counts.resize(units);
    r.total_hits.resize(units);
    for (Part& part : part_) {
      if (!part.h) continue;
      const irs_hip_hit* hits = nullptr;
      const uint32_t* counts = nullptr;
      const uint64_t* totals = nullptr;
      uint32_t stride = 0;
      check(irs_hip_batch_host_results(part.h, &hits, &stride, &counts, &totals),
            "irs_hip_batch_host_results");
      const size_t nq = part.index.size();
      for (uint32_t s = 0; s < n_segments_; ++s)
        for (size_t i = 0; i < nq; ++i) {
          const size_t from = s * nq + i, to = size_t(s) * n_queries_ + part.index[i];
          std::copy_n(hits + from * stride, counts[from], r.hits.begin() + to * k_);
          r.counts[to] = counts[from];
          r.total_hits[to] = totals[from];
        }
    }
    return r;
  }
  // the one device batch of a list of queries that are all boolean or all by_phrase
  // (what search_sharded hands to irs_hip_batch_results_to_device)
  irs_hip_batch* single_part() const {
    if (part_[0].h && part_[1].h) throw not_supported(IRS_HIP_EUNSUPPORTED, "mixed boolean / phrase batch");
    return part_[0].h ? part_[0].h : part_[1].h;
  }
  uint32_t reruns() const {
    uint32_t total = 0;
    for (const Part& part : part_) {
      uint32_t n = 0;
      if (part.h) check(irs_hip_batch_reruns(part.h, &n), "irs_hip_batch_reruns");
      total += n;
    }
    return total;
  }

 private:
  struct Part {  // [0] boolean queries, [1] phrase queries
    irs_hip_batch* h = nullptr;
    std::vector<uint32_t> index;  // position of each of its queries in the caller's list
  };
  Part part_[2];
  uint32_t n_segments_, n_queries_, k_;
};

// ---- the harness: one heap over all segments (utils/index-search.cpp:719-787) ---------------
struct ScoredDoc {
  float score;
  uint32_t segment;  // ordinal in the list the batch was created with
  uint32_t doc;      // segment-local id, as index-search prints it (:807)
};

// Global top-k of every query from the per-segment top-k lists, ordered (score desc,
// segment asc, doc asc) — the order tests/search/wand_test.cpp:72-86 fixes.  (Host-side
// counterpart of irs_hip_merge_topk, which does the same on device-resident lists.)
inline std::vector<std::vector<ScoredDoc>> merge(const QueryBatch::Results& r) {
  std::vector<std::vector<ScoredDoc>> out(r.n_queries);
  for (uint32_t q = 0; q < r.n_queries; ++q) {
    auto& v = out[q];
    for (uint32_t s = 0; s < r.n_segments; ++s) {
      const irs_hip_hit* h = r.of(s, q);
      for (uint32_t i = 0; i < r.count(s, q); ++i) v.push_back(ScoredDoc{h[i].score, s, h[i].doc});
    }
    std::sort(v.begin(), v.end(), [](const ScoredDoc& x, const ScoredDoc& y) {
      if (x.score != y.score) return x.score > y.score;
      if (x.segment != y.segment) return x.segment < y.segment;
      return x.doc < y.doc;
    });
    if (v.size() > r.k) v.resize(r.k);
  }
  return out;
}

// filter.prepare(index) -> execute on every segment -> one top-k per query.
template<typename Scorer>
std::vector<std::vector<ScoredDoc>> search(const std::vector<const SegmentReader*>& segments,
                                           const std::vector<SegmentStats>& index,
                                           const std::vector<filter>& filters,
                                           const Scorer& scorer, uint32_t k) {
  QueryBatch batch(segments, prepare(filters, scorer, index), k);
  batch.set_shared_threshold(true);   // (merged right here: one threshold per query)
  return merge(batch.run().results());
}

// ---- adapter groundwork: what a postings_reader built on this library does on the HOST --------
// before anything reaches the GPU: validate the files, decode the term
// dictionary's term_meta entries, read the Norm2 column header.
namespace format10 {

constexpr int32_t kFormatMagic = 0x3fd76c17;                 // format_utils.hpp:36
constexpr uint32_t kFooterLen = 2 * 4 + 8;                   // format_utils.hpp:38
constexpr const char* kDocFormatName = "iresearch_10_postings_documents";   // formats_10.cpp:325-326
constexpr const char* kPosFormatName = "iresearch_10_postings_positions";   // :327-328
constexpr uint32_t kBlockSize = IRS_HIP_BLOCK_SIZE;

inline uint32_t be32(const uint8_t* p) {
  return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3];
}
inline uint64_t be64(const uint8_t* p) { return (uint64_t(be32(p)) << 32) | be32(p + 4); }

// bytes_io<T>::vread (bytes_utils.hpp:176-206): LEB128, least significant group first
template<typename T>
inline T vread(const uint8_t*& p) {
  T v = 0;
  for (unsigned shift = 0;; shift += 7) {
    const uint8_t b = *p++;
    v |= T(b & 0x7Fu) << shift;
    if (!(b & 0x80u) || shift + 7 >= sizeof(T) * 8 + 6) break;
  }
  return v;
}

// CRC-32C (Castagnoli), bit-reflected, as absl::ExtendCrc32c from 0 (utils/crc.hpp:31-54)
inline uint32_t crc32c(const uint8_t* p, size_t n) {
  static const std::array<uint32_t, 256> table = [] {
    std::array<uint32_t, 256> t{};
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? 0x82F63B78u : 0u);
      t[i] = c;
    }
    return t;
  }();
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

// format_utils::check_header (format_utils.cpp:74-105): magic, format name, version range.
// Returns the version; *header_len = where the postings start.
inline int32_t check_header(const uint8_t* f, uint64_t len, const char* format, int32_t min_ver,
                            int32_t max_ver, size_t* header_len = nullptr) {
  const size_t nlen = std::char_traits<char>::length(format);
  const size_t expected = 4 + 1 + nlen + 4;   // header_length(): name sizes < 128 take one vint byte
  if (len < expected) throw index_error(IRS_HIP_ECORRUPT, "While checking header, error: file too short");
  if (int32_t(be32(f)) != kFormatMagic)
    throw index_error(IRS_HIP_ECORRUPT, "While checking header, error: invalid magic");
  if (f[4] != nlen || std::char_traits<char>::compare(reinterpret_cast<const char*>(f + 5), format, nlen))
    throw index_error(IRS_HIP_ECORRUPT, "While checking header, error: format mismatch");
  const int32_t ver = int32_t(be32(f + 5 + nlen));
  if (ver < min_ver || ver > max_ver)
    throw index_error(IRS_HIP_ECORRUPT, "While checking header, error: invalid version");
  if (header_len) *header_len = expected;
  return ver;
}

// validate_footer + check_footer (format_utils.cpp:32-53, format_utils.hpp:44-69): footer magic,
// algorithm id 0 and — with `verify_checksum` — the CRC-32C of everything in front of it.
inline void check_footer(const uint8_t* f, uint64_t len, bool verify_checksum = true) {
  if (len < kFooterLen) throw index_error(IRS_HIP_ECORRUPT, "While validating footer, error: invalid position");
  const uint8_t* t = f + len - kFooterLen;
  if (int32_t(be32(t)) != -kFormatMagic)
    throw index_error(IRS_HIP_ECORRUPT, "While validating footer, error: invalid magic number");
  if (be32(t + 4) != 0)
    throw index_error(IRS_HIP_ECORRUPT, "While validating footer, error: invalid algorithm");
  if (verify_checksum && be64(t + 8) != crc32c(f, len - 8))
    throw index_error(IRS_HIP_ECORRUPT, "While validating footer, error: checksum mismatch");
}

// postings_reader_base::decode (formats_10.cpp:3421-3456): one term_meta entry of a term
// dictionary block, delta-coded against the previous entry of the block (`state` carries it:
// doc_start / pos_start / pay_start accumulate; a block's first entry starts from zeros).
// Returns the bytes consumed.
inline size_t decode_term_meta(const uint8_t* in, bool has_freq, bool has_pos, bool has_pay_or_offs,
                               irs_hip_term_meta& state) {
  const uint8_t* p = in;
  state.docs_count = vread<uint32_t>(p);
  if (has_freq) state.freq = state.docs_count + vread<uint32_t>(p);
  state.doc_start += vread<uint64_t>(p);
  if (has_freq && state.freq && has_pos) {
    state.pos_start += vread<uint64_t>(p);
    state.pos_end = state.freq > kBlockSize ? vread<uint64_t>(p) : ~uint64_t(0);   // address_limits::invalid()
    if (has_pay_or_offs) state.pay_start += vread<uint64_t>(p);
  }
  if (state.docs_count == 1) {
    state.e_skip_start = vread<uint32_t>(p);     // e_single_doc (the union's other member)
  } else if (state.docs_count > kBlockSize) {
    state.e_skip_start = vread<uint64_t>(p);
  }
  return size_t(p - in);
}

// Norm2Header::Read (norm.cpp:117-146; layout :107-115): version byte, bytes per value,
// min and max field length (big-endian).
struct Norm2Header {
  uint32_t num_bytes = 0, min = 0, max = 0;
  // Norm2ReaderContext::max_num_bytes: 1 selects the norm_cache BM25 variant (bm25.cpp:466-470)
  uint32_t max_num_bytes() const { return max <= 0xFFu ? 1u : (max <= 0xFFFFu ? 2u : 4u); }
};
inline bool read_norm2_header(const uint8_t* payload, size_t size, Norm2Header& out) {
  if (size != 10 || payload[0] != 0) return false;             // ByteSize(), Norm2Version::kMin
  if (payload[1] != 1 && payload[1] != 2 && payload[1] != 4) return false;   // CheckNumBytes
  out.num_bytes = payload[1];
  out.min = be32(payload + 2);
  out.max = be32(payload + 6);
  return true;
}


// ---- the term dictionary (`.tm`), walked WITHOUT the term index ------------------------------
// What the reference's term iterator yields — (term bytes, term_meta) for every term of a
// field (term_reader::iterator + next(), formats_burst_trie.cpp:3196-3297; the cookie holds the
// version10::term_meta, :802-826) — recovered from the blocks of `.tm` alone.  The term index
// (`.ti`, an FST) only exists to seek; the blocks link to each other:
//   block  = vint (entries << 1 | last block of its group)
//            vlong (suffix bytes << 1 | leaf), the suffixes, vlong stats bytes, the stats
//            (block_iterator::load :1765-1825, written by field_writer::WriteBlock :1023-1112)
//   suffix = vint length (leaf) or (length << 1 | is-block), the bytes and, for a block entry,
//            vlong (this block's start - the sub-block's start)   (read_entry_nonleaf :1827-1848)
//   stats  = one postings_writer::encode record per TERM entry, delta-coded from zeros at the
//            block's first (decode_term_meta above)
// A term = the prefix its block's group stands for + the suffix.  Blocks are written children
// first, so every sub-block reference points backwards; the blocks of one prefix that was too
// large for one block ("floor" blocks) follow each other and only the first is referenced —
// the "last of its group" bit says where a group ends.  Hence: parse all blocks front to back,
// form the groups, and hand prefixes down from the unreferenced (root) group of each field.
struct DictTerm {
  std::string term;
  irs_hip_term_meta meta;
};

inline std::vector<DictTerm> walk_term_dictionary(const uint8_t* tm, uint64_t len, bool has_freq,
                                                  bool has_pos, bool has_pay_or_offs) {
  size_t hl = 0;
  const int32_t version = check_header(tm, len, "block_tree_terms_dict", 0, 3, &hl);
  check_footer(tm, len);
  const uint8_t* p = tm + hl;
  const uint8_t* const end = tm + len - kFooterLen;
  auto need = [&](uint64_t n) {
    if (uint64_t(end - p) < n) throw index_error(IRS_HIP_ECORRUPT, "term dictionary: truncated");
  };
  if (version > 0) {   // irs::encrypt (encryption.cpp:37-45): the cipher header, empty = none
    need(5);
    const uint32_t enc = vread<uint32_t>(p);
    if (enc) throw not_supported(IRS_HIP_EUNSUPPORTED, "term dictionary: encrypted segment");
  }
  {                    // postings_reader_base::prepare (formats_10.cpp:3404-3416)
    size_t h2 = 0;
    check_header(p, uint64_t(end - p), "iresearch_10_postings_terms", 0, 0, &h2);
    p += h2;
    need(5);
    if (vread<uint32_t>(p) != kBlockSize)
      throw index_error(IRS_HIP_ECORRUPT, "term dictionary: invalid postings block size");
  }
  struct Ent {
    uint64_t suffix_at;               // into the file
    uint32_t suffix_len;
    bool is_block;
    uint64_t child;                   // is_block: file offset of the sub-block
    irs_hip_term_meta meta;           // term
  };
  struct Block {
    uint64_t start;
    bool last;
    std::vector<Ent> ents;
  };
  std::vector<Block> blocks;
  while (p < end) {
    Block b;
    b.start = uint64_t(p - tm);
    need(2);
    const uint32_t head = vread<uint32_t>(p);
    b.last = (head & 1u) != 0;
    const uint32_t n = head >> 1;
    need(1);
    const uint64_t sz = vread<uint64_t>(p);
    const bool leaf = (sz & 1u) != 0;
    const uint64_t suffix_bytes = sz >> 1;
    need(suffix_bytes);
    const uint8_t* sp = p;
    const uint8_t* const send = p + suffix_bytes;
    p = send;
    need(1);
    const uint64_t stats_bytes = vread<uint64_t>(p);
    need(stats_bytes);
    const uint8_t* st = p;
    const uint8_t* const stend = p + stats_bytes;
    p = stend;
    irs_hip_term_meta state{};   // (a block's first record is coded against zeros)
    state.pos_end = ~uint64_t(0);
    for (uint32_t i = 0; i < n; ++i) {
      if (sp >= send) throw index_error(IRS_HIP_ECORRUPT, "term dictionary: suffix block overrun");
      Ent e{};
      uint32_t v = vread<uint32_t>(sp);
      e.is_block = !leaf && (v & 1u);
      e.suffix_len = leaf ? v : (v >> 1);
      e.suffix_at = uint64_t(sp - tm);
      if (uint64_t(send - sp) < e.suffix_len)
        throw index_error(IRS_HIP_ECORRUPT, "term dictionary: suffix block overrun");
      sp += e.suffix_len;
      if (e.is_block) {
        const uint64_t back = vread<uint64_t>(sp);
        if (back == 0 || back > b.start)
          throw index_error(IRS_HIP_ECORRUPT, "term dictionary: sub-block pointer out of range");
        e.child = b.start - back;
      } else {
        if (st >= stend) throw index_error(IRS_HIP_ECORRUPT, "term dictionary: stats block overrun");
        {   // (decoded from a copy: a damaged record cannot send the varint reads past the block)
          uint8_t buf[64] = {0};
          const size_t have = size_t(std::min<uint64_t>(uint64_t(stend - st), sizeof buf));
          std::memcpy(buf, st, have);
          const size_t used = decode_term_meta(buf, has_freq, has_pos, has_pay_or_offs, state);
          if (used > have) throw index_error(IRS_HIP_ECORRUPT, "term dictionary: stats block overrun");
          st += used;
        }
        e.meta = state;
      }
      b.ents.push_back(e);
    }
    if (sp != send || st != stend)
      throw index_error(IRS_HIP_ECORRUPT, "term dictionary: block sizes do not add up");
    blocks.push_back(std::move(b));
  }
  // groups of floor blocks; who references which group
  std::vector<size_t> group_first;            // index of the first block of every group
  std::vector<size_t> group_of(blocks.size());
  for (size_t i = 0; i < blocks.size();) {
    group_first.push_back(i);
    size_t j = i;
    for (;; ++j) {
      if (j >= blocks.size()) throw index_error(IRS_HIP_ECORRUPT, "term dictionary: open block group");
      group_of[j] = group_first.size() - 1;
      if (blocks[j].last) break;
    }
    i = j + 1;
  }
  std::vector<char> referenced(group_first.size(), 0);
  auto group_at = [&](uint64_t start) -> size_t {
    size_t lo = 0, hi = group_first.size();
    while (lo < hi) {   // groups ascend by start
      const size_t mid = (lo + hi) / 2;
      if (blocks[group_first[mid]].start < start) lo = mid + 1; else hi = mid;
    }
    if (lo == group_first.size() || blocks[group_first[lo]].start != start)
      throw index_error(IRS_HIP_ECORRUPT, "term dictionary: sub-block pointer hits no block");
    return lo;
  };
  for (const Block& b : blocks)
    for (const Ent& e : b.ents)
      if (e.is_block) referenced[group_at(e.child)] = 1;
  std::vector<DictTerm> out;
  // `.tm` holds the blocks of EVERY field of the segment (field_writer opens it once): without
  // the term index this walk can only serve a dictionary of one field — several root groups
  // mean several fields, whose terms (and stats layouts) must not be mixed: read_term_index +
  // walk_field are the way in then
  if (std::count(referenced.begin(), referenced.end(), char(0)) > 1)
    throw not_supported(IRS_HIP_EUNSUPPORTED, "term dictionary: several fields (root blocks) — needs the term index (.ti)");
  // depth first from the root group
  struct Todo {
    size_t group;
    std::string prefix;
  };
  for (size_t g = 0; g < group_first.size(); ++g) {
    if (referenced[g]) continue;
    std::vector<Todo> todo{{g, std::string()}};
    while (!todo.empty()) {
      Todo t = std::move(todo.back());
      todo.pop_back();
      for (size_t bi = group_first[t.group];; ++bi) {
        for (const Ent& e : blocks[bi].ents) {
          std::string full = t.prefix;
          full.append(reinterpret_cast<const char*>(tm + e.suffix_at), e.suffix_len);
          if (e.is_block) todo.push_back(Todo{group_at(e.child), std::move(full)});
          else out.push_back(DictTerm{std::move(full), e.meta});
        }
        if (blocks[bi].last) break;
      }
    }
  }
  std::sort(out.begin(), out.end(), [](const DictTerm& a, const DictTerm& b) { return a.term < b.term; });
  return out;
}

// ---- the term index (`.ti`) and the segment meta (`.sm`): what a field IS ----------------------
// A bounds-checked cursor over untrusted file bytes: every read says how much it needs.
class Cursor {
 public:
  Cursor(const uint8_t* p, const uint8_t* end, const char* what) : p_{p}, end_{end}, what_{what} {}
  uint64_t left() const { return uint64_t(end_ - p_); }
  const uint8_t* at() const { return p_; }
  void need(uint64_t n) const {
    if (left() < n) throw index_error(IRS_HIP_ECORRUPT, std::string(what_) + ": truncated");
  }
  uint8_t u8() { need(1); return *p_++; }
  uint32_t u32() { need(4); const uint32_t v = be32(p_); p_ += 4; return v; }
  uint64_t u64() { need(8); const uint64_t v = be64(p_); p_ += 8; return v; }
  template<typename T>
  T v() {   // vint / vlong: at most ceil(bits / 7) bytes, all inside the file
    T out = 0;
    for (unsigned shift = 0;; shift += 7) {
      const uint8_t b = u8();
      out |= T(b & 0x7Fu) << shift;
      if (!(b & 0x80u)) break;
      if (shift + 7 >= sizeof(T) * 8 + 6) throw index_error(IRS_HIP_ECORRUPT, std::string(what_) + ": overlong varint");
    }
    return out;
  }
  std::string str() {   // read_string: vint size + bytes
    const uint32_t n = v<uint32_t>();
    need(n);
    std::string out(reinterpret_cast<const char*>(p_), n);
    p_ += n;
    return out;
  }
  void skip(uint64_t n) { need(n); p_ += n; }

 private:
  const uint8_t* p_;
  const uint8_t* end_;
  const char* what_;
};

// One field of a segment as the term index records it: term_reader_base::prepare
// (formats_burst_trie.cpp:1509-1545) — name, read_field_features (:741-766: index features, the
// feature -> column id pairs), term / doc / posting counts, min and max term, the summed
// frequency of a field with FREQ, the wand mask — followed by the field's FST
// (ImmutableFstImpl::Read, utils/fstext/immutable_fst.hpp:136-203), of which the final weight of
// the START state is kept: the record of the field's ROOT block (MergeBlocks :856-911: meta byte,
// vlong start, the other floor blocks of the group), i.e. where block_iterator starts from
// (:1751-1764).  The prefix arcs only accelerate seeks.
struct FieldRecord {
  std::string name;
  uint32_t index_features = 0;      // IndexFeatures: FREQ 1, POS 2, OFFS 4, PAY 8
  int64_t norm_column = -1;         // field_meta::features[Norm2] (-1: the field has no norms)
  uint64_t terms_count = 0;
  uint64_t docs_with_field = 0;     // term_reader::docs_count()
  uint64_t total_doc_freq = 0;
  uint64_t total_term_freq = 0;     // irs::frequency of the field (FREQ only)
  std::string min_term, max_term;
  uint64_t wand_mask = 0;
  uint8_t root_meta = 0;
  uint64_t root_start = 0;          // file offset of the root block in `.tm`
  bool has_freq() const { return (index_features & 1u) != 0; }
  bool has_pos() const { return (index_features & 2u) != 0; }
  bool has_offs_or_pay() const { return (index_features & 12u) != 0; }
  uint32_t wand_count() const { return uint32_t(__builtin_popcountll(wand_mask)); }
};
struct TermIndex {
  uint32_t index_features = 0;            // of the segment
  std::vector<std::string> features;      // the segment's feature names, by id
  std::vector<FieldRecord> fields;        // in name order
  const FieldRecord* find(const std::string& name) const {
    for (const FieldRecord& f : fields)
      if (f.name == name) return &f;
    return nullptr;
  }
};
constexpr const char* kNorm2Feature = "iresearch::norm2";   // irs::Norm2::type_name(), norm.hpp:204-206

// field_reader::prepare (formats_burst_trie.cpp:3323-3440), the `.ti` half
inline TermIndex read_term_index(const uint8_t* ti, uint64_t len) {
  size_t hl = 0;
  const int32_t version = check_header(ti, len, "block_tree_terms_index", 0, 3, &hl);
  check_footer(ti, len);
  if (version < 2) throw not_supported(IRS_HIP_EUNSUPPORTED, "term index: formats before the immutable FST (1_3)");
  if (len < hl + 8 + kFooterLen) throw index_error(IRS_HIP_ECORRUPT, "term index: truncated");
  const uint64_t fields_count = be64(ti + len - kFooterLen - 8);
  Cursor in(ti + hl, ti + len - kFooterLen - 8, "term index");
  if (in.v<uint32_t>()) throw not_supported(IRS_HIP_EUNSUPPORTED, "term index: encrypted segment");
  TermIndex out;
  out.index_features = in.u32();   // read_segment_features :711-733
  if (out.index_features > 15u) throw index_error(IRS_HIP_ECORRUPT, "term index: invalid segment index features");
  for (uint64_t n = in.v<uint64_t>(); n; --n) out.features.push_back(in.str());
  for (uint64_t f = 0; f < fields_count; ++f) {
    FieldRecord r;
    r.name = in.str();
    if (!out.fields.empty() && !(out.fields.back().name < r.name))
      throw index_error(IRS_HIP_ECORRUPT, "term index: invalid field order");
    r.index_features = in.u32();
    if (r.index_features > 15u) throw index_error(IRS_HIP_ECORRUPT, "term index: invalid field index features");
    for (uint64_t n = in.v<uint64_t>(); n; --n) {
      const uint64_t id = in.v<uint64_t>();
      const uint64_t column = in.v<uint64_t>();   // field_id + 1
      if (id >= out.features.size()) throw index_error(IRS_HIP_ECORRUPT, "term index: unknown feature id");
      if (out.features[id] == kNorm2Feature) r.norm_column = int64_t(column) - 1;
    }
    r.terms_count = in.v<uint64_t>();
    r.docs_with_field = in.v<uint64_t>();
    r.total_doc_freq = in.v<uint64_t>();
    r.min_term = in.str();
    r.max_term = in.str();
    if (r.has_freq()) r.total_term_freq = in.v<uint64_t>();
    if (version >= 3) r.wand_mask = in.u64();
    // the FST: header, states (+ arcs), weights; the start state's final weight is the root
    if (in.u8() != 0) throw index_error(IRS_HIP_ECORRUPT, "term index: unknown FST version");
    in.u64();   // properties
    const uint64_t total_weight = in.u64();
    const uint32_t nstates = in.u32();
    const uint32_t back = in.v<uint32_t>();
    if (!nstates || back == 0 || back > nstates) throw index_error(IRS_HIP_ECORRUPT, "term index: FST without a start state");
    const uint32_t start = nstates - back;
    in.v<uint64_t>();   // zig-zag(arcs - states)
    uint64_t weight_at = 0, root_at = 0, root_len = 0;
    for (uint32_t st = 0; st < nstates; ++st) {
      const uint64_t packed = in.v<uint64_t>();
      const uint64_t wsize = packed >> 1;
      if (st == start) {
        root_at = weight_at;
        root_len = wsize;
      }
      weight_at += wsize;
      if (!(packed & 1u)) {   // has arcs
        for (uint32_t arcs = uint32_t(in.u8()) + 1u; arcs; --arcs) {
          in.u8();               // label
          in.v<uint32_t>();      // next state
          weight_at += in.v<uint64_t>();
        }
      }
    }
    if (weight_at != total_weight || root_at + root_len > total_weight)
      throw index_error(IRS_HIP_ECORRUPT, "term index: FST weights do not add up");
    in.need(total_weight);
    {
      Cursor w(in.at() + root_at, in.at() + root_at + root_len, "term index: root block record");
      r.root_meta = w.u8();
      r.root_start = w.v<uint64_t>();
    }
    in.skip(total_weight);
    out.fields.push_back(std::move(r));
  }
  if (in.left()) throw index_error(IRS_HIP_ECORRUPT, "term index: bytes behind the last field");
  return out;
}

// SegmentMetaReader::read (formats_10.cpp:3147-3218)
struct SegmentMetaFile {
  std::string name;
  uint64_t version = 0, docs_count = 0, live_docs_count = 0, byte_size = 0;
  bool column_store = false;
  std::vector<std::string> files;
};
inline SegmentMetaFile read_segment_meta(const uint8_t* sm, uint64_t len) {
  size_t hl = 0;
  const int32_t version = check_header(sm, len, "iresearch_10_segment_meta", 0, 1, &hl);
  check_footer(sm, len);
  Cursor in(sm + hl, sm + len - kFooterLen, "segment meta");
  SegmentMetaFile m;
  m.name = in.str();
  m.version = in.v<uint64_t>();
  m.live_docs_count = in.v<uint64_t>();
  m.docs_count = in.v<uint64_t>() + m.live_docs_count;
  m.byte_size = in.v<uint64_t>();
  const uint8_t flags = in.u8();
  if (flags & ~3u) throw index_error(IRS_HIP_ECORRUPT, "segment meta: unsupported flags");
  uint64_t sort = 0;
  if (version > 0) sort = in.v<uint64_t>();   // 1 + the sort column, 0 = none
  if (((flags & 2u) != 0) != (sort != 0)) throw index_error(IRS_HIP_ECORRUPT, "segment meta: sorted flag and column disagree");
  m.column_store = (flags & 1u) != 0;
  for (uint64_t n = in.v<uint64_t>(); n; --n) m.files.push_back(in.str());
  return m;
}

// DocumentMaskReader::read (formats_10.cpp:3275-3312): the ids of a segment's deleted docs, in
// file order (the writer iterates a hash set: any order) — what index_utils::ReadDocumentMask
// (index_utils.cpp:476) hands SegmentReaderImpl::Update, and irs_hip_segment_desc::doc_mask here.
inline std::vector<uint32_t> read_document_mask(const uint8_t* dm, uint64_t len) {
  size_t hl = 0;
  check_header(dm, len, "iresearch_10_doc_mask", 0, 0, &hl);
  check_footer(dm, len);
  Cursor in(dm + hl, dm + len - kFooterLen, "document mask");
  std::vector<uint32_t> docs;
  for (uint32_t n = in.v<uint32_t>(); n; --n) docs.push_back(in.v<uint32_t>());
  if (in.left()) throw index_error(IRS_HIP_ECORRUPT, "document mask: bytes behind the last doc id");
  return docs;
}

// The terms of ONE field of `.tm`, in order, from its root block (FieldRecord::root_start): a
// block group = the blocks from the referenced one to the first with the "last of its group"
// bit; an entry is a term (suffix + stats record) or a sub-block (suffix + back pointer) whose
// group is walked in place — entries are stored in suffix order, so the terms come out sorted.
inline std::vector<DictTerm> walk_field(const uint8_t* tm, uint64_t len, uint64_t root_start,
                                        bool has_freq, bool has_pos, bool has_pay_or_offs) {
  size_t hl = 0;
  const int32_t version = check_header(tm, len, "block_tree_terms_dict", 0, 3, &hl);
  check_footer(tm, len);
  const uint8_t* const end = tm + len - kFooterLen;
  uint64_t first_block = 0;
  {
    Cursor in(tm + hl, end, "term dictionary");
    if (version > 0 && in.v<uint32_t>()) throw not_supported(IRS_HIP_EUNSUPPORTED, "term dictionary: encrypted segment");
    size_t h2 = 0;   // postings_reader_base::prepare (formats_10.cpp:3404-3416)
    check_header(in.at(), in.left(), "iresearch_10_postings_terms", 0, 0, &h2);
    in.skip(h2);
    if (in.v<uint32_t>() != kBlockSize) throw index_error(IRS_HIP_ECORRUPT, "term dictionary: invalid postings block size");
    first_block = uint64_t(in.at() - tm);
  }
  std::vector<DictTerm> out;
  struct Todo { uint64_t start; std::string prefix; };
  // (an explicit stack: sub-block groups are expanded depth first, in entry order)
  std::function<void(uint64_t, const std::string&, unsigned)> group = [&](uint64_t start, const std::string& prefix, unsigned depth) {
    if (depth > 512) throw index_error(IRS_HIP_ECORRUPT, "term dictionary: block tree too deep");
    for (uint64_t at = start;;) {
      if (at < first_block || at >= uint64_t(end - tm)) throw index_error(IRS_HIP_ECORRUPT, "term dictionary: block pointer out of range");
      Cursor in(tm + at, end, "term dictionary");
      const uint32_t head = in.v<uint32_t>();
      const bool last = (head & 1u) != 0;
      const uint32_t n = head >> 1;
      const uint64_t sz = in.v<uint64_t>();
      const bool leaf = (sz & 1u) != 0;
      in.need(sz >> 1);
      Cursor sx(in.at(), in.at() + (sz >> 1), "term dictionary: suffix block");
      in.skip(sz >> 1);
      const uint64_t stats_bytes = in.v<uint64_t>();
      in.need(stats_bytes);
      const uint8_t* st = in.at();
      const uint8_t* const stend = st + stats_bytes;
      in.skip(stats_bytes);
      irs_hip_term_meta state{};   // (a block's first record is coded against zeros)
      state.pos_end = ~uint64_t(0);
      for (uint32_t i = 0; i < n; ++i) {
        const uint32_t v = sx.v<uint32_t>();
        const bool is_block = !leaf && (v & 1u);
        const uint32_t slen = leaf ? v : (v >> 1);
        sx.need(slen);
        std::string full = prefix;
        full.append(reinterpret_cast<const char*>(sx.at()), slen);
        sx.skip(slen);
        if (is_block) {
          const uint64_t back = sx.v<uint64_t>();
          if (back == 0 || back > at) throw index_error(IRS_HIP_ECORRUPT, "term dictionary: sub-block pointer out of range");
          group(at - back, full, depth + 1);
        } else {
          // (a stats record is at most 5 + 5 + 4 * 10 bytes of varints: decode from a copy that
          // cannot run past the block)
          uint8_t buf[64] = {0};
          const size_t have = size_t(std::min<uint64_t>(uint64_t(stend - st), sizeof buf));
          if (!have) throw index_error(IRS_HIP_ECORRUPT, "term dictionary: stats block overrun");
          std::memcpy(buf, st, have);
          const size_t used = decode_term_meta(buf, has_freq, has_pos, has_pay_or_offs, state);
          if (used > have) throw index_error(IRS_HIP_ECORRUPT, "term dictionary: stats block overrun");
          st += used;
          out.push_back(DictTerm{std::move(full), state});
        }
      }
      if (sx.left() || st != stend) throw index_error(IRS_HIP_ECORRUPT, "term dictionary: block sizes do not add up");
      if (last) break;
      at = uint64_t(in.at() - tm);   // the group's next floor block follows
    }
  };
  group(root_start, std::string(), 0);
  for (size_t i = 1; i < out.size(); ++i)
    if (!(out[i - 1].term < out[i].term)) throw index_error(IRS_HIP_ECORRUPT, "term dictionary: terms out of order");
  return out;
}

// ---- columnstore2: the fixed-length column a Norm2 feature lives in ---------------------------
// reader::prepare_index (columnstore2.cpp:1746-1830): per column — in name order — the
// compression id, the column header (read_header :79-88), the payload (for Norm2: its
// Norm2Header, norm.hpp:83-125), the name unless anonymous, the bitmap index of a column that
// lacks docs, then what its type needs: fixed_length_column (:792-1011) the value length and
// one data offset per 65536-doc block, dense_fixed_length_column (:650-789) the value length
// and ONE offset — value of doc d at data + len * (d - header.min) (:736-740).  The feature
// columns of a field are anonymous; field_meta::features maps the feature to the column id.
struct FixedColumn {
  uint32_t id = 0, min_doc = 0, docs_count = 0;
  uint32_t value_bytes = 0;
  std::vector<uint8_t> payload;
  std::vector<uint8_t> values;   // dense: value of doc d at value_bytes * (d - min_doc)
};

inline FixedColumn read_fixed_column(const uint8_t* csi, uint64_t csi_len, const uint8_t* csd,
                                     uint64_t csd_len, uint32_t column_id) {
  size_t hl = 0;
  check_header(csi, csi_len, "iresearch_11_columnstore_index", 0, 0, &hl);
  check_footer(csi, csi_len);
  size_t dl = 0;
  check_header(csd, csd_len, "iresearch_11_columnstore_data", 0, 0, &dl);
  check_footer(csd, csd_len, /*verify_checksum=*/false);   // (the reader only validates the data footer)
  const uint8_t* p = csi + hl;
  const uint8_t* const end = csi + csi_len - kFooterLen;
  auto need = [&](uint64_t n) {
    if (uint64_t(end - p) < n) throw index_error(IRS_HIP_ECORRUPT, "columnstore index: truncated");
  };
  auto skip_string = [&](const uint8_t** at = nullptr) -> uint32_t {
    need(1);
    const uint32_t n = vread<uint32_t>(p);
    need(n);
    if (at) *at = p;
    p += n;
    return n;
  };
  constexpr uint32_t kColumnBlock = 1u << 16;   // column::kBlockSize = sparse_bitmap_writer::kBlockSize
  need(1);
  const uint32_t count = vread<uint32_t>(p);
  for (uint32_t c = 0; c < count; ++c) {
    const uint8_t* comp = nullptr;
    const uint32_t comp_len = skip_string(&comp);
    need(24);
    const uint64_t docs_index = be64(p);
    const uint32_t id = be32(p + 8), min_doc = be32(p + 12), docs = be32(p + 16);
    const uint32_t type = (uint32_t(p[20]) << 8) | p[21], props = (uint32_t(p[22]) << 8) | p[23];
    p += 24;
    if (id >= count) throw index_error(IRS_HIP_ECORRUPT, "columnstore index: invalid ordinal position");
    const uint8_t* payload = nullptr;
    const uint32_t payload_len = skip_string(&payload);
    if (!(props & 2u)) skip_string();                       // ColumnProperty::kNoName
    if (docs_index) {                                       // read_bitmap_index (:109-133)
      need(4);
      const uint32_t nb = be32(p);
      p += 4;
      if (nb > 0xFFFFu) throw index_error(IRS_HIP_ECORRUPT, "columnstore index: invalid number of blocks");
      if (nb > 2) {
        need(uint64_t(nb) * 8);
        p += uint64_t(nb) * 8;
      }
    }
    const uint32_t nblocks = (docs + kColumnBlock - 1) / kColumnBlock;
    const bool mine = id == column_id;
    if (mine && (type != 2 && type != 3))
      throw not_supported(IRS_HIP_EUNSUPPORTED, "columnstore: the column is not a fixed-length column");
    if (mine && (props & 1u))
      throw not_supported(IRS_HIP_EUNSUPPORTED, "columnstore: encrypted column");
    if (mine && docs_index)
      throw not_supported(IRS_HIP_EUNSUPPORTED, "columnstore: the column lacks docs (not dense)");
    if (mine && !(comp_len == 28 && !std::memcmp(comp, "iresearch::compression::none", 28)))
      throw not_supported(IRS_HIP_EUNSUPPORTED, "columnstore: compressed column");
    switch (type) {
      case 0:   // kSparse: addr, avg, bits, data, last_size per block (write_blocks_sparse :135-145)
        need(uint64_t(nblocks) * 33);
        p += uint64_t(nblocks) * 33;
        break;
      case 1:   // kMask: nothing
        break;
      case 2:   // kFixed: value length, one data offset per block
      case 3: { // kDenseFixed: value length, ONE offset
        need(8);
        const uint64_t len = be64(p);
        p += 8;
        const uint32_t offsets = type == 3 ? 1u : nblocks;
        need(uint64_t(offsets) * 8);
        if (mine) {
          if (len != 1 && len != 2 && len != 4)
            throw not_supported(IRS_HIP_EUNSUPPORTED, "columnstore: value length is not 1, 2 or 4");
          FixedColumn col;
          col.id = id;
          col.min_doc = min_doc;
          col.docs_count = docs;
          col.value_bytes = uint32_t(len);
          col.payload.assign(payload, payload + payload_len);
          col.values.resize(uint64_t(docs) * len);
          for (uint32_t b = 0; b < nblocks; ++b) {
            // (offsets come from the file: no sum of them may wrap)
            const uint64_t limit = csd_len - kFooterLen;
            const uint64_t first = type == 3 ? be64(p) : be64(p + 8ull * b);
            const uint64_t shift = type == 3 ? uint64_t(b) * kColumnBlock * len : 0;
            const uint64_t n = std::min<uint64_t>(kColumnBlock, docs - uint64_t(b) * kColumnBlock) * len;
            if (first < dl || first > limit || shift > limit - first || n > limit - first - shift)
              throw index_error(IRS_HIP_ECORRUPT, "columnstore: block outside the data file");
            const uint64_t at = first + shift;
            std::memcpy(col.values.data() + uint64_t(b) * kColumnBlock * len, csd + at, n);
          }
          return col;
        }
        p += uint64_t(offsets) * 8;
        break;
      }
      default:
        throw index_error(IRS_HIP_ECORRUPT, "columnstore index: invalid column type");
    }
  }
  throw index_error(IRS_HIP_ECORRUPT, "columnstore: no column with that id");
}

// The files of one field of one segment, as they lie on disk -> a SegmentReader: the term
// dictionary walk yields the term table (terms ascending: ordinal i = i-th term), the
// columnstore the dense Norm2 column (norm_column < 0: the field has no norms) — the host half
// of a real-index ingestion adapter: what `irs_hip_segment_open` needs, from
// nothing but file bytes.  `terms_out` receives the term bytes per ordinal.
struct FieldFiles {
  const uint8_t* doc = nullptr;  uint64_t doc_len = 0;
  const uint8_t* pos = nullptr;  uint64_t pos_len = 0;     // null: not staged (by_phrase unavailable)
  const uint8_t* tm = nullptr;   uint64_t tm_len = 0;      // term dictionary (all fields)
  const uint8_t* ti = nullptr;   uint64_t ti_len = 0;      // term index: the fields' records
  const uint8_t* sm = nullptr;   uint64_t sm_len = 0;      // segment meta: the doc count
  const uint8_t* csi = nullptr;  uint64_t csi_len = 0;     // columnstore index + data
  const uint8_t* csd = nullptr;  uint64_t csd_len = 0;
  const uint8_t* doc_mask = nullptr;  uint64_t doc_mask_len = 0;   // `.doc_mask`: deleted docs (optional file)
  std::string field;             // which field of the segment
  // What the files do not say: which Scorer::WandType the field's wand data was written by
  // (it is a property of the scorer objects the index was created with, scorer.hpp:196-201).
  uint32_t wand_type = IRS_HIP_WAND_NONE;
  // WITHOUT a term index / segment meta (a single-field dictionary, e.g. one written by a test):
  // what they would have said
  int64_t norm_column = -1;
  uint32_t num_docs = 0;
  bool has_freq = true;
  uint32_t wand_count = 0;
};
struct OpenedField {
  std::vector<std::string> terms;            // ordinal -> term bytes
  std::vector<irs_hip_term_meta> metas;
  FixedColumn norms;
  Norm2Header norm_header;
  FieldRecord record;                         // (from the term index)
  uint64_t docs_with_field = 0;
  uint64_t total_term_freq = 0;
  uint32_t num_docs = 0;
  std::vector<uint32_t> doc_mask;            // the segment's deleted docs (DocumentMask)
};
// Everything irs_hip_segment_open needs for one field of a segment, from file bytes alone:
// with `ti` and `sm` given, the field's features (FREQ / POS / OFFS / PAY), its Norm2 column id,
// docs_with_field, the summed term frequency, the wand mask and the root of its term blocks
// come from the term index, the doc count from the segment meta, the block layout (scalar /
// simd) from the `.doc` header's version (formats_10.cpp:283-313: odd = simd) — what
// field_reader::prepare + term_reader_base::prepare + SegmentMetaReader::read establish.
inline irs_hip_segment_desc describe_field(const FieldFiles& f, int32_t device, OpenedField& o) {
  std::vector<DictTerm> dict;
  bool has_freq = f.has_freq, has_pos = f.pos != nullptr, has_pay = false;
  int64_t norm_column = f.norm_column;
  uint32_t wand_count = f.wand_count;
  o.num_docs = f.num_docs;
  o.record = FieldRecord{};
  if (f.ti) {
    const TermIndex index = read_term_index(f.ti, f.ti_len);
    const FieldRecord* rec = index.find(f.field);
    if (!rec) throw index_error(IRS_HIP_EINVAL, "segment has no field '" + f.field + "'");
    o.record = *rec;
    has_freq = rec->has_freq();
    has_pos = rec->has_pos();
    has_pay = rec->has_offs_or_pay();
    norm_column = rec->norm_column;
    wand_count = rec->wand_count();
    dict = walk_field(f.tm, f.tm_len, rec->root_start, has_freq, has_pos, has_pay);
    if (dict.size() != rec->terms_count)
      throw index_error(IRS_HIP_ECORRUPT, "term dictionary: the field's term count disagrees with the term index");
  } else {
    dict = walk_term_dictionary(f.tm, f.tm_len, f.has_freq, f.pos != nullptr, false);
  }
  if (f.sm) {
    const SegmentMetaFile meta = read_segment_meta(f.sm, f.sm_len);
    if (meta.docs_count == 0 || meta.docs_count > 0x7FFF0000ull)
      throw index_error(IRS_HIP_ECORRUPT, "segment meta: doc count out of range");
    o.num_docs = uint32_t(meta.docs_count);
  }
  o.terms.clear();
  o.metas.clear();
  o.total_term_freq = 0;
  uint64_t doc_freq = 0;
  for (const DictTerm& t : dict) {
    o.terms.push_back(t.term);
    o.metas.push_back(t.meta);
    o.total_term_freq += t.meta.freq;
    doc_freq += t.meta.docs_count;
  }
  o.docs_with_field = f.ti ? o.record.docs_with_field : o.num_docs;
  if (f.ti && (doc_freq != o.record.total_doc_freq || (has_freq && o.total_term_freq != o.record.total_term_freq)))
    throw index_error(IRS_HIP_ECORRUPT, "term dictionary: the field's totals disagree with the term index");
  int32_t doc_version = 0;
  {
    size_t hl = 0;
    doc_version = check_header(f.doc, f.doc_len, kDocFormatName, 0, 5, &hl);
  }
  irs_hip_segment_desc d{};
  d.device = device;
  d.layout = (doc_version & 1) ? IRS_HIP_LAYOUT_SIMD4 : IRS_HIP_LAYOUT_SCALAR;
  d.doc_file = f.doc;
  d.doc_file_len = f.doc_len;
  d.num_docs = o.num_docs;
  d.has_freq = has_freq ? 1u : 0u;
  if (norm_column >= 0) {
    o.norms = read_fixed_column(f.csi, f.csi_len, f.csd, f.csd_len, uint32_t(norm_column));
    if (!read_norm2_header(o.norms.payload.data(), o.norms.payload.size(), o.norm_header) ||
        o.norm_header.num_bytes != o.norms.value_bytes)
      throw index_error(IRS_HIP_ECORRUPT, "norm column: invalid Norm2 header");
    d.norms = o.norms.values.data();
    d.norm_width = o.norms.value_bytes;
    d.norm_min_doc = o.norms.min_doc;
    d.norm_count = o.norms.docs_count;
  }
  d.terms = o.metas.data();
  d.num_terms = uint32_t(o.metas.size());
  d.wand_count = wand_count;
  d.wand_type = f.wand_type;
  if (has_pos && f.pos) {
    d.pos_file = f.pos;
    d.pos_file_len = f.pos_len;
    d.pos_features = (o.record.index_features & 4u ? IRS_HIP_POS_OFFSETS : 0u) |
                     (o.record.index_features & 8u ? IRS_HIP_POS_PAYLOADS : 0u);
  }
  // SegmentReaderImpl::Open (segment_reader_impl.cpp:167-172): the optional `.doc_mask`; every
  // batch on the segment then behaves like an iterator behind SegmentReaderImpl::mask
  o.doc_mask.clear();
  if (f.doc_mask) {
    o.doc_mask = read_document_mask(f.doc_mask, f.doc_mask_len);
    if (f.sm) {   // (SegmentMeta::live_docs_count = docs_count - the mask's size: index_writer)
      const SegmentMetaFile meta = read_segment_meta(f.sm, f.sm_len);
      if (meta.live_docs_count + o.doc_mask.size() != meta.docs_count)
        throw index_error(IRS_HIP_ECORRUPT, "document mask: size disagrees with the segment meta's live docs");
    }
    d.doc_mask = o.doc_mask.data();
    d.doc_mask_count = o.doc_mask.size();
  }
  return d;
}

}  // namespace format10

// ---- multi-term expansion filters without scorers ------------------------------
// by_prefix / by_range / by_terms with an EMPTY order: every term the filter visits is an
// "unscored" one and the segment's iterator is lazy_bitset_iterator — one postings_reader::bit_union
// over the visited terms' cookies (multiterm_query.cpp:64-101, 162-167; prefix_filter.cpp:37-60;
// range_filter.cpp:62-108).  Here the visit is a range of ordinals of the field's sorted term
// table (OpenedField::terms: what term_reader::iterator() enumerates) and the union is
// irs_hip_bit_union on the device.
enum class BoundType { UNBOUNDED, INCLUSIVE, EXCLUSIVE };   // search/search_range.hpp
struct by_prefix { std::string prefix; };
struct by_range {
  std::string min, max;
  BoundType min_type = BoundType::UNBOUNDED, max_type = BoundType::UNBOUNDED;
};
struct by_terms { std::vector<std::string> terms; };

// the ordinals a filter visits in a field's term table (ascending byte order)
inline std::vector<uint32_t> visit(const std::vector<std::string>& terms, const by_prefix& f) {
  std::vector<uint32_t> out;   // seek_ge(prefix), then while the term starts with it (prefix_filter.cpp:44-58)
  auto it = std::lower_bound(terms.begin(), terms.end(), f.prefix);
  for (; it != terms.end() && it->compare(0, f.prefix.size(), f.prefix) == 0; ++it)
    out.push_back(uint32_t(it - terms.begin()));
  return out;
}
inline std::vector<uint32_t> visit(const std::vector<std::string>& terms, const by_range& f) {
  std::vector<uint32_t> out;
  if (f.min_type != BoundType::UNBOUNDED && f.max_type != BoundType::UNBOUNDED && f.min == f.max &&
      !(f.min_type == BoundType::INCLUSIVE && f.max_type == BoundType::INCLUSIVE))
    return out;   // "can't satisfy condition" (range_filter.cpp:122-131)
  auto it = terms.begin();
  if (f.min_type == BoundType::INCLUSIVE) it = std::lower_bound(terms.begin(), terms.end(), f.min);
  else if (f.min_type == BoundType::EXCLUSIVE) it = std::upper_bound(terms.begin(), terms.end(), f.min);
  for (; it != terms.end(); ++it) {
    if (f.max_type == BoundType::INCLUSIVE && !(*it <= f.max)) break;
    if (f.max_type == BoundType::EXCLUSIVE && !(*it < f.max)) break;
    out.push_back(uint32_t(it - terms.begin()));
  }
  return out;
}
inline std::vector<uint32_t> visit(const std::vector<std::string>& terms, const by_terms& f) {
  std::vector<uint32_t> out;   // terms the segment does not hold are skipped (terms_filter.cpp: seek fails)
  for (const std::string& t : f.terms) {
    const auto it = std::lower_bound(terms.begin(), terms.end(), t);
    if (it != terms.end() && *it == t) out.push_back(uint32_t(it - terms.begin()));
  }
  std::sort(out.begin(), out.end());
  out.erase(std::unique(out.begin(), out.end()), out.end());
  return out;
}

// The docs of a segment a filter without scorers accepts: bit `doc` of `words` (bit 0 —
// doc_limits::invalid() — never set), as lazy_bitset_iterator::refill leaves them.
struct DocSet {
  std::vector<uint64_t> words;
  uint64_t postings = 0;   // bit_union's return value: the sum of the visited terms' docs_count
  bool contains(uint32_t doc) const { return doc / 64 < words.size() && ((words[doc / 64] >> (doc % 64)) & 1u); }
  uint64_t count() const {
    uint64_t n = 0;
    for (uint64_t w : words) n += uint64_t(__builtin_popcountll(w));
    return n;
  }
};
template<typename Filter>
DocSet execute_unscored(const SegmentReader& segment, const std::vector<std::string>& field_terms,
                        uint32_t num_docs, const Filter& flt) {
  DocSet out;
  out.words.assign((uint64_t(num_docs) + 1 + 63) / 64, 0);   // docs_count + doc_limits::min() bits
  const std::vector<uint32_t> ordinals = visit(field_terms, flt);
  if (!ordinals.empty()) out.postings = segment.bit_union(ordinals, out.words);
  return out;
}

// ---- scored multi-term filters: by_prefix / by_wildcard / by_range with scored_terms_limit ----
// MultiTermQuery (core/search/multiterm_query.cpp:77-145) over the states that
// limited_sample_collector<term_frequency> keeps (core/search/limited_sample_collector.hpp
// :67-120, 126-170): the `scored_terms_limit` visited terms with the largest (docs_count, visit
// offset) keys are scored, each with the statistics of the segments where it IS scored; every other
// visited term only contributes documents (score 0).
struct PreparedExpansion {
  PreparedQuery scored;                            // Or of the scored terms, segment_terms filled
  std::vector<std::vector<uint32_t>> scored_in;    // [segment]: ordinals scored there (ascending)
  std::vector<std::vector<uint32_t>> unscored_in;  // [segment]: visited, not scored (visit order)
};

// limited_sample_collector::collect over the segments in order -> (segment, visit offset) of the
// states left scored.  The container algorithm is the reference's: an index min-heap moved by
// std::push_heap / std::pop_heap, a new key replaces the root only when strictly larger.
inline std::vector<std::pair<uint32_t, uint32_t>> scored_states(
    const std::vector<std::vector<uint32_t>>& visits_docs_counts, size_t limit) {
  struct State {
    uint32_t frequency, offset, segment;
    bool less(uint32_t f, uint32_t o) const { return frequency < f || (frequency == f && offset < o); }
  };
  std::vector<State> states;
  std::vector<size_t> heap;
  auto comp = [&](size_t l, size_t r) { return states[r].less(states[l].frequency, states[l].offset); };
  for (uint32_t s = 0; limit && s < visits_docs_counts.size(); ++s) {
    const auto& counts = visits_docs_counts[s];
    for (uint32_t off = 0; off < counts.size(); ++off) {
      if (states.size() < limit) {
        heap.push_back(states.size());
        states.push_back(State{counts[off], off, s});
        std::push_heap(heap.begin(), heap.end(), comp);
      } else if (const size_t root = heap.front(); states[root].less(counts[off], off)) {
        std::pop_heap(heap.begin(), heap.end(), comp);
        states[root] = State{counts[off], off, s};
        std::push_heap(heap.begin(), heap.end(), comp);
      }
    }
  }
  std::vector<std::pair<uint32_t, uint32_t>> out;
  for (const State& st : states) out.emplace_back(st.segment, st.offset);
  std::sort(out.begin(), out.end());
  return out;
}

// filter::prepare of ONE scored multi-term filter: visits[s] = the ordinals the filter's visitor
// yields in segment s, in term order (visit(field_terms, flt) per segment).
template<typename Scorer>
PreparedExpansion prepare_expansion(const std::vector<std::vector<uint32_t>>& visits,
                                    size_t scored_terms_limit, const Scorer& scorer,
                                    const std::vector<SegmentStats>& index, float boost = 1.f) {
  if (visits.size() != index.size())
    throw illegal_argument(IRS_HIP_EINVAL, "prepare_expansion: one visit per segment");
  uint64_t dwf = 0, ttf = 0;
  for (const auto& s : index) {
    dwf += s.docs_with_field;
    ttf += s.total_term_freq;
  }
  std::vector<std::vector<uint32_t>> counts(visits.size());
  for (size_t s = 0; s < visits.size(); ++s)
    for (uint32_t t : visits[s]) counts[s].push_back(uint32_t(index[s].docs_count(t)));
  PreparedExpansion out;
  out.scored_in.resize(visits.size());
  out.unscored_in.resize(visits.size());
  std::vector<std::vector<bool>> is_scored(visits.size());
  for (size_t s = 0; s < visits.size(); ++s) is_scored[s].assign(visits[s].size(), false);
  for (const auto& [s, off] : scored_states(counts, scored_terms_limit)) is_scored[s][off] = true;
  std::vector<uint32_t> slots;
  for (size_t s = 0; s < visits.size(); ++s) {
    for (size_t off = 0; off < visits[s].size(); ++off)
      (is_scored[s][off] ? out.scored_in[s] : out.unscored_in[s]).push_back(visits[s][off]);
    std::sort(out.scored_in[s].begin(), out.scored_in[s].end());
    slots.insert(slots.end(), out.scored_in[s].begin(), out.scored_in[s].end());
  }
  std::sort(slots.begin(), slots.end());
  slots.erase(std::unique(slots.begin(), slots.end()), slots.end());
  PreparedQuery& q = out.scored;
  q.op = IRS_HIP_OP_OR;
  q.segment_terms.assign(visits.size(), std::vector<uint32_t>(std::max<size_t>(1, slots.size()),
                                                              IRS_HIP_NO_TERM));
  for (size_t j = 0; j < slots.size(); ++j) {
    uint64_t dwt = 0;   // (collector::score: the term's statistics where its state is scored)
    for (size_t s = 0; s < visits.size(); ++s)
      if (std::binary_search(out.scored_in[s].begin(), out.scored_in[s].end(), slots[j])) {
        dwt += index[s].docs_count(slots[j]);
        q.segment_terms[s][j] = slots[j];
      }
    TermStats st;
    scorer.collect(st, dwf, dwt, ttf);
    irs_hip_term_scorer e = scorer.term_scorer(st, boost);
    e.term = slots[j];
    q.terms.push_back(e);
  }
  if (slots.empty()) {   // nothing scored: a one-slot query of an absent term (matches nothing)
    TermStats st;
    scorer.collect(st, dwf ? dwf : 1, 1, ttf);
    irs_hip_term_scorer e = scorer.term_scorer(st, boost);
    e.term = IRS_HIP_NO_TERM;
    q.terms.push_back(e);
  }
  return out;
}

// MultiTermQuery::execute on every (filter, segment) + the harness's top k per segment.  The
// scored disjunctions of all filters are ONE batch; total(s, q) is the population of the union of
// ALL visited terms' postings (what the disjunction of the scored iterators and the
// lazy_bitset_iterator of the unscored ones yields); where fewer than k docs score, the list is
// filled with the docs only unscored terms hold (score 0, ascending doc id).
inline QueryBatch::Results execute_expansions(const std::vector<const SegmentReader*>& segments,
                                              const std::vector<uint32_t>& segment_docs,
                                              const std::vector<PreparedExpansion>& prepared,
                                              uint32_t k) {
  std::vector<PreparedQuery> queries;
  for (const auto& p : prepared) queries.push_back(p.scored);
  QueryBatch batch(segments, queries, k);
  QueryBatch::Results r = batch.run().results();
  for (uint32_t s = 0; s < r.n_segments; ++s) {
    std::vector<uint32_t> need;
    std::vector<std::vector<uint32_t>> visited;
    for (uint32_t q = 0; q < r.n_queries; ++q) {
      const auto& p = prepared[q];
      if (p.unscored_in[s].empty()) continue;
      need.push_back(q);
      visited.push_back(p.unscored_in[s]);
      visited.back().insert(visited.back().end(), p.scored_in[s].begin(), p.scored_in[s].end());
    }
    const std::vector<uint64_t> totals = segments[s]->bit_union_counts(visited);
    const size_t words = (uint64_t(segment_docs[s]) + 1 + 63) / 64;
    for (size_t i = 0; i < need.size(); ++i) {
      const size_t unit = size_t(s) * r.n_queries + need[i];
      r.total_hits[unit] = totals[i];
      const uint32_t have = r.counts[unit];
      if (have >= k || totals[i] <= have) continue;
      std::vector<uint64_t> all(words, 0), sc(words, 0);
      segments[s]->bit_union(visited[i], all);
      if (!prepared[need[i]].scored_in[s].empty())
        segments[s]->bit_union(prepared[need[i]].scored_in[s], sc);
      uint32_t n = have;
      for (size_t w = 0; w < words && n < k; ++w)
        for (uint64_t bits = all[w] & ~sc[w]; bits && n < k; bits &= bits - 1)
          r.hits[unit * k + n++] = irs_hip_hit{0.f, uint32_t(w * 64 + __builtin_ctzll(bits))};
      r.counts[unit] = n;
    }
  }
  return r;
}

// ---- several GPUs: one process per GPU, segments sharded, ONE all-gather per batch -----------
// (what the harness loop over `reader`'s segments becomes when the segments
// live on different devices)
class DeviceBuffer {
 public:
  DeviceBuffer(int32_t device, uint64_t bytes) : device_{device}, bytes_{bytes} {
    check(irs_hip_device_alloc(device, bytes ? bytes : 1, &p_), "irs_hip_device_alloc");
  }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  ~DeviceBuffer() { irs_hip_device_free(device_, p_); }
  void* get() const noexcept { return p_; }
  char* at(uint64_t off) const noexcept { return static_cast<char*>(p_) + off; }
  uint64_t bytes() const noexcept { return bytes_; }

 private:
  int32_t device_;
  uint64_t bytes_;
  void* p_ = nullptr;
};

class Communicator {
 public:
  using Id = std::array<uint8_t, IRS_HIP_COMM_ID_BYTES>;
  // on ONE rank; the caller carries the bytes to the others (MPI_Bcast, a file, a socket)
  static Id unique_id() {
    Id id{};
    check(irs_hip_comm_unique_id(id.data()), "irs_hip_comm_unique_id");
    return id;
  }
  Communicator(int32_t device, const Id& id, int n_ranks, int rank)
    : device_{device}, n_ranks_{n_ranks}, rank_{rank} {
    check(irs_hip_comm_init_rank(device, id.data(), n_ranks, rank, &h_), "irs_hip_comm_init_rank");
  }
  Communicator(const Communicator&) = delete;
  Communicator& operator=(const Communicator&) = delete;
  ~Communicator() { irs_hip_comm_destroy(h_); }
  int n_ranks() const noexcept { return n_ranks_; }
  int rank() const noexcept { return rank_; }
  int32_t device() const noexcept { return device_; }
  void all_gather(const void* d_send, void* d_recv, uint64_t bytes_per_rank, void* stream = nullptr) {
    check(irs_hip_topk_allgather(h_, d_send, d_recv, bytes_per_rank, stream), "irs_hip_topk_allgather");
  }
  irs_hip_comm* handle() const noexcept { return h_; }

 private:
  int32_t device_;
  int n_ranks_, rank_;
  irs_hip_comm* h_ = nullptr;
};

// Every rank calls this with ITS segments (ranks hold consecutive blocks of `per_rank` segments
// of the index's `n_segments`; the last block may be short, a rank may hold none — 5 segments on
// 8 ranks) and the statistics of ALL segments of the index (they are index-global:
// term_filter.cpp:102-125).  Returns, on every rank, the global top-k per query:
// ScoredDoc::segment is the GLOBAL ordinal.  The queries must be all boolean or all by_phrase.
// Every rank reaches the collective whatever happens locally: a rank whose own execution
// fails sends a failure mark with its (empty) block and all ranks throw AFTER the all-gather —
// nobody is left waiting in it.
template<typename Scorer>
std::vector<std::vector<ScoredDoc>> search_sharded(Communicator& comm,
                                                   const std::vector<const SegmentReader*>& mine,
                                                   uint32_t per_rank, uint32_t n_segments,
                                                   const std::vector<SegmentStats>& index,
                                                   const std::vector<filter>& filters,
                                                   const Scorer& scorer, uint32_t k, bool wand = false) {
  const uint32_t nq = uint32_t(filters.size());
  const int32_t dev = comm.device();
  if (mine.size() > per_rank) throw illegal_argument(IRS_HIP_EINVAL, "more local segments than per_rank");
  // one block per rank: per_rank hit tables [nq][k], then per_rank count tables [nq] (padded
  // to 8 bytes) — irs_hip_batch_results_to_device writes the local lists straight into it —
  // then 8 bytes of status (0 = this rank's part is good)
  const uint64_t hit_bytes = uint64_t(per_rank) * nq * k * sizeof(irs_hip_hit);
  const uint64_t cnt_bytes = (uint64_t(per_rank) * nq * 4 + 7) & ~uint64_t(7);
  const uint64_t block = hit_bytes + cnt_bytes + 8;
  DeviceBuffer send(dev, block), recv(dev, block * uint64_t(comm.n_ranks()));
  std::vector<char> zero(block, 0);   // slots of segments this rank does not have: count 0
  check(irs_hip_device_upload(dev, send.get(), zero.data(), block), "irs_hip_device_upload");
  std::string local_error;
  std::unique_ptr<QueryBatch> batch;
  try {
    if (!mine.empty()) {   // (a rank without segments has no batch to build)
      batch = std::make_unique<QueryBatch>(mine, prepare(filters, scorer, index), k);
      if (wand) batch->set_wand(true);
      batch->set_shared_threshold(true);   // (the lists are merged below: one threshold per query)
    }
  } catch (const std::exception& e) {
    local_error = e.what();
    batch.reset();
  }
  // ONE threshold per query over ALL ranks' segments (irs_hip_batch_set_comm) needs every rank
  // inside the batch's collectives: only when every rank holds segments and built its batch —
  // agreed on with one 8-byte all-gather; otherwise each rank keeps the threshold of its own
  // segments (same merged result, more candidates)
  {
    DeviceBuffer flag(dev, 8), flags(dev, 8 * uint64_t(comm.n_ranks()));
    const uint64_t ready = batch ? 1 : 0;
    check(irs_hip_device_upload(dev, flag.get(), &ready, 8), "irs_hip_device_upload");
    check(irs_hip_device_sync(dev, nullptr), "irs_hip_device_sync");
    comm.all_gather(flag.get(), flags.get(), 8);
    std::vector<uint64_t> all(size_t(comm.n_ranks()));
    check(irs_hip_device_download(dev, all.data(), flags.get(), 8 * all.size()), "irs_hip_device_download");
    bool everyone = true;
    for (uint64_t f : all) everyone = everyone && f != 0;
    if (everyone && comm.n_ranks() > 1) batch->set_comm(comm.handle());
  }
  try {
    if (batch) {
      batch->run();
      check(irs_hip_batch_results_to_device(batch->single_part(), send.at(0), send.at(hit_bytes), nullptr),
            "irs_hip_batch_results_to_device");
      check(irs_hip_device_sync(dev, nullptr), "irs_hip_device_sync");
    } else if (!local_error.empty()) {
      throw error(IRS_HIP_EHIP, local_error);
    }
  } catch (const std::exception& e) {
    local_error = e.what();
    const uint64_t failed = 1;     // the block goes out empty, marked
    std::memcpy(zero.data() + hit_bytes + cnt_bytes, &failed, 8);
    check(irs_hip_device_upload(dev, send.get(), zero.data(), block), "irs_hip_device_upload");
  }
  check(irs_hip_device_sync(dev, nullptr), "irs_hip_device_sync");
  comm.all_gather(send.get(), recv.get(), block);
  for (int r = 0; r < comm.n_ranks(); ++r) {
    uint64_t st = 0;
    check(irs_hip_device_download(dev, &st, recv.at(block * uint64_t(r) + hit_bytes + cnt_bytes), 8),
          "irs_hip_device_download");
    if (st) {
      throw error(IRS_HIP_EHIP, r == comm.rank() ? "search_sharded: " + local_error
                                                 : "search_sharded: rank " + std::to_string(r) + " failed");
    }
  }
  std::vector<const void*> lists, counts;
  std::vector<uint32_t> ids;
  for (uint32_t s = 0; s < n_segments; ++s) {
    const uint32_t r = s / per_rank, j = s % per_rank;
    lists.push_back(recv.at(block * r + uint64_t(j) * nq * k * sizeof(irs_hip_hit)));
    counts.push_back(recv.at(block * r + hit_bytes + uint64_t(j) * nq * 4));
    ids.push_back(s);
  }
  DeviceBuffer out_h(dev, uint64_t(nq) * k * sizeof(irs_hip_hit)), out_s(dev, uint64_t(nq) * k * 4),
    out_c(dev, uint64_t(nq) * 4);
  check(irs_hip_merge_topk(dev, lists.data(), counts.data(), ids.data(), n_segments, nq, k,
                           out_h.get(), out_s.get(), out_c.get(), nullptr),
        "irs_hip_merge_topk");
  std::vector<irs_hip_hit> h(size_t(nq) * k);
  std::vector<uint32_t> sg(size_t(nq) * k), c(nq);
  check(irs_hip_device_download(dev, h.data(), out_h.get(), h.size() * sizeof(irs_hip_hit)), "download");
  check(irs_hip_device_download(dev, sg.data(), out_s.get(), sg.size() * 4), "download");
  check(irs_hip_device_download(dev, c.data(), out_c.get(), c.size() * 4), "download");
  std::vector<std::vector<ScoredDoc>> out(nq);
  for (uint32_t q = 0; q < nq; ++q)
    for (uint32_t i = 0; i < c[q]; ++i)
      out[q].push_back(ScoredDoc{h[size_t(q) * k + i].score, sg[size_t(q) * k + i], h[size_t(q) * k + i].doc});
  return out;
}

}  // namespace irs_hip_host
