// tests/cpp/test_host.cpp — TEST: the C++ host layer (iresearch_amd/cpp/irs_hip.hpp) above the
// C ABI, written the way the reference's own search tests read (build an index, prepare
// filters against it with a scorer, execute per segment, compare with expectations), checked
// against the oracle's C API.  Linked against libirs_hip.so on a GPU box, or against the CPU
// emulator build of the same sources (tests/sim) in the CPU tier.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "irs_hip.hpp"
#include "oracle.h"
#include "synth_index.h"

using namespace irs_hip_host;

#define REQUIRE(c)                                                          \
  do {                                                                      \
    if (!(c)) {                                                             \
      std::fprintf(stderr, "%s:%d: REQUIRE(%s) failed\n", __FILE__, __LINE__, #c); \
      return 1;                                                             \
    }                                                                       \
  } while (0)

namespace {

struct Segment {  // a synthetic segment and everything the two sides need of it
  irs_synth_index* idx = nullptr;
  const uint8_t *doc = nullptr, *pos = nullptr, *norms = nullptr;
  uint64_t doc_len = 0, pos_len = 0, norm_count = 0;
  const irs_hip_term_meta* metas = nullptr;
  uint32_t num_terms = 0, num_docs = 0;
  std::unique_ptr<SegmentReader> reader;

  Segment(uint32_t docs, uint64_t first_doc, uint32_t max_rank) : num_docs{docs} {
    irs_synth_params p{};
    p.seed = 20260926;
    p.first_doc = first_doc;
    p.num_docs = docs;
    p.vocab_log2 = 20;
    p.max_rank = max_rank;
    p.layout = IRS_SYNTH_LAYOUT_SIMD4;
    p.mean_len = 100;
    p.stddev_len = 30;
    p.with_positions = 1;
    if (irs_synth_build(&p, &idx) != 0) throw std::runtime_error("irs_synth_build");
    doc = irs_synth_doc_bytes(idx, &doc_len);
    pos = irs_synth_pos_bytes(idx, &pos_len);
    norms = irs_synth_norms(idx, &norm_count);
    static_assert(sizeof(irs_synth_term_meta) == sizeof(irs_hip_term_meta), "same layout");
    static_assert(sizeof(orc_term_meta) == sizeof(irs_hip_term_meta), "same layout");
    metas = reinterpret_cast<const irs_hip_term_meta*>(irs_synth_term_metas(idx, &num_terms));
    reader = std::make_unique<SegmentReader>(desc());
  }
  ~Segment() {
    reader.reset();
    irs_synth_free(idx);
  }
  irs_hip_segment_desc desc() const {
    irs_hip_segment_desc d{};
    d.device = 0;
    d.layout = IRS_HIP_LAYOUT_SIMD4;
    d.doc_file = doc;
    d.doc_file_len = doc_len;
    d.num_docs = num_docs;
    d.has_freq = 1;
    d.norms = norms;
    d.norm_width = 1;
    d.norm_min_doc = 1;
    d.norm_count = norm_count;
    d.terms = metas;
    d.num_terms = num_terms;
    d.pos_file = pos;
    d.pos_file_len = pos_len;
    return d;
  }
  SegmentStats stats() const {
    return SegmentStats{irs_synth_docs_with_field(idx), irs_synth_total_term_freq(idx), metas,
                        num_terms};
  }
  orc_segment oracle_view() const {
    orc_segment s{};
    s.doc_file = doc;
    s.doc_file_len = doc_len;
    s.layout = ORC_LAYOUT_SIMD4;
    s.num_docs = num_docs;
    s.norms = norms;
    s.norm_width = 1;
    s.pos_file = pos;
    s.pos_file_len = pos_len;
    return s;
  }
};

bool close_rel(float a, float b) { return std::fabs(a - b) <= 1e-5f * std::fabs(b); }

}  // namespace

int main() {
  constexpr uint32_t kMaxRank = 128, kTop = 50;
  Segment a(20000, 0, kMaxRank), b(9000, 20000, kMaxRank);
  const Segment* segs[2] = {&a, &b};

  // filters, as a caller of the reference would write them
  std::vector<filter> filters;
  uint32_t ranks[4 * 8];
  REQUIRE(irs_synth_queries(20260928, 4, 8, 2, kMaxRank, ranks) == 0);
  for (int q = 0; q < 4; ++q) {
    Or f;
    for (int t = 0; t < 8; ++t) f.subs.push_back(by_term{ranks[q * 8 + t] - 1});
    filters.push_back(f);
  }
  filters.push_back(by_term{7});
  filters.push_back(And{{by_term{1}, by_term{20}, by_term{3, 2.5f}}});
  filters.push_back(Or{{by_term{2}, by_term{9}, by_term{30}, by_term{5}}, 2});
  filters.push_back(by_phrase{}.push_back(0).push_back(1));
  filters.push_back(by_phrase{}.push_back(2).push_back(0, 1).push_back(1));  // a gap of one word
  filters.push_back(by_term{kMaxRank + 500});                                  // no such term
  {  // boolean_filter::merge_type(): kMax on a disjunction, kMin on a conjunction
    Or mx{{by_term{4}, by_term{11}, by_term{40}}};
    mx.merge_type = IRS_HIP_MERGE_MAX;
    filters.push_back(mx);
    And mn{{by_term{1}, by_term{6}}};
    mn.merge_type = IRS_HIP_MERGE_MIN;
    filters.push_back(mn);
  }

  const BM25 scorer;  // k = 1.2, b = 0.75
  const auto prepared = prepare(filters, scorer, {a.stats(), b.stats()});
  QueryBatch batch({a.reader.get(), b.reader.get()}, prepared, kTop);
  const auto res = batch.run().results();
  REQUIRE(res.n_segments == 2 && res.n_queries == filters.size());
  {  // the serving-loop form: the copy queued behind the run, read later — the same arrays
    const auto again = batch.results_to_host().host_results();
    REQUIRE(again.counts == res.counts && again.total_hits == res.total_hits);
    for (size_t i = 0; i < res.hits.size(); ++i)
      REQUIRE(again.hits[i].doc == res.hits[i].doc && again.hits[i].score == res.hits[i].score);
  }
  const auto top = merge(res);  // the harness heap over both segments

  // the oracle's harness loop over both segments (utils/index-search.cpp:719-787)
  const orc_segment views[2] = {a.oracle_view(), b.oracle_view()};
  const uint64_t dwf[2] = {a.stats().docs_with_field, b.stats().docs_with_field};
  const uint64_t ttf[2] = {a.stats().total_term_freq, b.stats().total_term_freq};
  const orc_scorer osc{ORC_SCORER_BM25, scorer.k(), scorer.b(), 0};
  for (size_t q = 0; q < filters.size(); ++q) {
    const PreparedQuery& p = prepared[q];
    const uint32_t n = uint32_t(p.terms.size());
    std::vector<orc_term_meta> metas(2 * n);
    std::vector<float> boosts(n, 1.f);
    std::vector<uint32_t> offsets(n);
    for (uint32_t s = 0; s < 2; ++s)
      for (uint32_t t = 0; t < n; ++t) {
        const uint32_t term = p.terms[t].term;
        if (term < segs[s]->num_terms)
          std::memcpy(&metas[s * n + t], &segs[s]->metas[term], sizeof(orc_term_meta));
        offsets[t] = p.terms[t].phrase_offset;
      }
    std::vector<orc_hit> want(kTop);
    uint64_t want_total = 0;
    int64_t got_n;
    if (p.op == IRS_HIP_OP_PHRASE) {
      got_n = orc_search_phrase(views, 2, metas.data(), n, offsets.data(), &osc,
                                std::get<by_phrase>(filters[q]).boost, dwf, ttf, kTop, want.data(),
                                &want_total);
    } else {
      if (const auto* f = std::get_if<And>(&filters[q]))
        for (uint32_t t = 0; t < n; ++t) boosts[t] = f->subs[t].boost;
      const int32_t op = (p.op == IRS_HIP_OP_MINMATCH ? (ORC_OP_MINMATCH | int32_t(p.min_match << 8))
                                                      : p.op) |
                         int32_t(p.merge << 24);   // ORC_MERGE_* == IRS_HIP_MERGE_*
      got_n = orc_search(views, 2, metas.data(), n, op, &osc, boosts.data(), dwf, ttf, kTop,
                         want.data(), &want_total);
    }
    REQUIRE(got_n >= 0);
    want.resize(size_t(got_n));
    const std::vector<ScoredDoc>& mine = top[q];
    REQUIRE(res.total(0, uint32_t(q)) + res.total(1, uint32_t(q)) == want_total);
    REQUIRE(mine.size() == want.size());
    std::sort(want.begin(), want.end(),
              [](const orc_hit& x, const orc_hit& y) { return x.score > y.score; });
    for (size_t i = 0; i < want.size(); ++i) REQUIRE(close_rel(mine[i].score, want[i].score));
  }
  REQUIRE(batch.reruns() == 0);
  {
    // planning queued ahead (irs_hip_batch_plan) changes nothing; irs::score::Min at every
    // query's k-th score gives the same lists again
    const auto planned = merge(batch.plan().run().results());
    std::vector<float> kth(filters.size(), 0.f);
    for (size_t q = 0; q < top.size(); ++q) {
      REQUIRE(planned[q].size() == top[q].size());
      for (size_t i = 0; i < top[q].size(); ++i)
        REQUIRE(planned[q][i].score == top[q][i].score && planned[q][i].doc == top[q][i].doc);
      if (!top[q].empty()) kth[q] = top[q].back().score;
    }
    // (... with the run kept on the caller's thread: irs_hip_batch_set_async)
    const auto pushed = merge(batch.set_min_scores(kth).set_async(false).run().results());
    batch.set_async(true);
    for (size_t q = 0; q < top.size(); ++q) {
      REQUIRE(pushed[q].size() == top[q].size());
      for (size_t i = 0; i < top[q].size(); ++i)
        REQUIRE(pushed[q][i].score == top[q][i].score && pushed[q][i].doc == top[q][i].doc);
    }
    batch.set_min_scores({});
    // the joined plain disjunctions on 32-bit tiles instead of paired tiles
    // (irs_hip_batch_set_paired_tiles): the same lists, bit for bit
    const auto unpaired = merge(batch.set_paired_tiles(false).run().results());
    REQUIRE(!batch.paired_tiles());
    batch.set_paired_tiles(true);
    for (size_t q = 0; q < top.size(); ++q) {
      REQUIRE(unpaired[q].size() == top[q].size());
      for (size_t i = 0; i < top[q].size(); ++i)
        REQUIRE(unpaired[q][i].score == top[q][i].score && unpaired[q][i].doc == top[q][i].doc);
    }
  }
  // the one-call form gives the same lists
  {
    const auto again = search({a.reader.get(), b.reader.get()}, {a.stats(), b.stats()}, filters,
                              scorer, kTop);
    REQUIRE(again.size() == top.size());
    for (size_t q = 0; q < top.size(); ++q) {
      REQUIRE(again[q].size() == top[q].size());
      for (size_t i = 0; i < top[q].size(); ++i)
        REQUIRE(again[q][i].score == top[q][i].score && again[q][i].doc == top[q][i].doc &&
                again[q][i].segment == top[q][i].segment);
    }
  }

  // postings / positions / bit_union of one term against the oracle's iterators
  {
    const uint32_t term = 11;
    std::vector<uint32_t> docs, freqs, pos;
    a.reader->postings(term, docs, &freqs, a.metas[term].docs_count);
    std::vector<uint32_t> odocs(docs.size()), ofreqs(docs.size());
    REQUIRE(orc_decode_term(a.doc, a.doc_len, ORC_LAYOUT_SIMD4,
                            reinterpret_cast<const orc_term_meta*>(&a.metas[term]), odocs.data(),
                            ofreqs.data(), odocs.size()) == int64_t(docs.size()));
    REQUIRE(docs == odocs && freqs == ofreqs);
    a.reader->positions(term, pos, a.metas[term].freq);
    std::vector<uint32_t> opos(pos.size());
    REQUIRE(orc_decode_positions(a.doc, a.doc_len, a.pos, a.pos_len, ORC_LAYOUT_SIMD4, 0,
                                 reinterpret_cast<const orc_term_meta*>(&a.metas[term]), 1,
                                 opos.data(), opos.size()) == int64_t(pos.size()));
    REQUIRE(pos == opos);
    std::vector<uint64_t> set((a.num_docs + 64) / 64, 0), oset(set.size(), 0);
    const std::vector<uint32_t> terms{3, 11, 90};
    orc_term_meta om[3];
    for (int i = 0; i < 3; ++i) std::memcpy(&om[i], &a.metas[terms[i]], sizeof om[i]);
    const uint64_t cnt = a.reader->bit_union(terms, set);
    REQUIRE(int64_t(cnt) == orc_bit_union(a.doc, a.doc_len, ORC_LAYOUT_SIMD4, 1, om, 3, oset.data(),
                                          oset.size()));
    REQUIRE(set == oset);
  }

  // scored multi-term filters (by_prefix / by_wildcard with scored_terms_limit): the collector's
  // choice, the statistics of a term where it is scored, totals over ALL visited terms, the fill
  {
    auto range = [](uint32_t lo, uint32_t hi) {
      std::vector<uint32_t> v;
      for (uint32_t t = lo; t < hi; ++t) v.push_back(t);
      return v;
    };
    struct Case { std::vector<std::vector<uint32_t>> visits; size_t limit; };
    const std::vector<Case> cases{
      {{range(10, 60), range(20, 70)}, 8},          // more visited terms than scored ones
      {{range(100, 128), range(90, 128)}, 3},       // rare terms: fewer than k docs score somewhere
      {{range(5, 9), {}}, 16},                      // everything scored; nothing visited in b
      {{range(40, 50), range(40, 50)}, 0},          // nothing scored at all
      {{range(120, 128), range(120, 128)}, 1},      // one state scored: the other segment only fills
    };
    uint64_t filled_anywhere = 0;
    std::vector<PreparedExpansion> prepared_x;
    for (const Case& c : cases)
      prepared_x.push_back(prepare_expansion(c.visits, c.limit, scorer, {a.stats(), b.stats()}));
    const auto rx = execute_expansions({a.reader.get(), b.reader.get()}, {a.num_docs, b.num_docs},
                                       prepared_x, kTop);
    const auto topx = merge(rx);
    for (size_t q = 0; q < cases.size(); ++q) {
      const PreparedExpansion& p = prepared_x[q];
      size_t n_scored = 0;
      for (int s = 0; s < 2; ++s) {
        n_scored += p.scored_in[s].size();
        REQUIRE(p.scored_in[s].size() + p.unscored_in[s].size() == cases[q].visits[s].size());
      }
      size_t n_visited = cases[q].visits[0].size() + cases[q].visits[1].size();
      REQUIRE(n_scored == std::min(cases[q].limit, n_visited));
      // the oracle: a disjunction of the scored slots, a term's meta zero where it is unscored
      const uint32_t n = uint32_t(p.scored.terms.size());
      std::vector<orc_term_meta> metas(2 * n);
      std::vector<float> boosts(n, 1.f);
      for (uint32_t s = 0; s < 2; ++s)
        for (uint32_t t = 0; t < n; ++t)
          if (p.scored.segment_terms[s][t] != IRS_HIP_NO_TERM)
            std::memcpy(&metas[s * n + t], &segs[s]->metas[p.scored.segment_terms[s][t]],
                        sizeof(orc_term_meta));
      std::vector<orc_hit> want(kTop);
      uint64_t scored_total = 0;
      const int64_t got_n = orc_search(views, 2, metas.data(), n, ORC_OP_OR, &osc, boosts.data(), dwf,
                                       ttf, kTop, want.data(), &scored_total);
      REQUIRE(got_n >= 0);
      std::sort(want.begin(), want.begin() + got_n,
                [](const orc_hit& x, const orc_hit& y) { return x.score > y.score; });
      uint64_t visited_total = 0, filled = 0;
      for (uint32_t s = 0; s < 2; ++s) {
        std::vector<orc_term_meta> om;
        for (uint32_t t : cases[q].visits[s]) {
          om.emplace_back();
          std::memcpy(&om.back(), &segs[s]->metas[t], sizeof(orc_term_meta));
        }
        std::vector<uint64_t> all((segs[s]->num_docs + 64) / 64, 0), sc(all.size(), 0);
        if (!om.empty())
          (void)orc_bit_union(segs[s]->doc, segs[s]->doc_len, ORC_LAYOUT_SIMD4, 1, om.data(),
                              uint32_t(om.size()), all.data(), all.size());
        om.clear();
        for (uint32_t t : p.scored_in[s]) {
          om.emplace_back();
          std::memcpy(&om.back(), &segs[s]->metas[t], sizeof(orc_term_meta));
        }
        if (!om.empty())
          (void)orc_bit_union(segs[s]->doc, segs[s]->doc_len, ORC_LAYOUT_SIMD4, 1, om.data(),
                              uint32_t(om.size()), sc.data(), sc.size());
        uint64_t pop = 0, pop_scored = 0;
        for (size_t w = 0; w < all.size(); ++w) {
          pop += uint64_t(__builtin_popcountll(all[w]));
          pop_scored += uint64_t(__builtin_popcountll(sc[w]));
        }
        REQUIRE(rx.total(s, uint32_t(q)) == pop);
        visited_total += pop;
        // this segment's list: the scored docs first, then docs only unscored terms hold
        const irs_hip_hit* h = rx.of(s, uint32_t(q));
        const uint32_t cnt = rx.count(s, uint32_t(q));
        REQUIRE(cnt == std::min<uint64_t>(kTop, pop));
        uint32_t prev = 0;
        for (uint32_t i = 0; i < cnt; ++i) {
          const bool in_scored = (sc[h[i].doc / 64] >> (h[i].doc % 64)) & 1;
          REQUIRE((all[h[i].doc / 64] >> (h[i].doc % 64)) & 1);
          if (i < std::min<uint64_t>(kTop, pop_scored)) {
            REQUIRE(in_scored && h[i].score > 0.f);
          } else {
            REQUIRE(!in_scored && h[i].score == 0.f && h[i].doc > prev);
            prev = h[i].doc;
            ++filled;
          }
        }
      }
      REQUIRE(visited_total >= scored_total);
      filled_anywhere += filled;
      // merged over both segments: the scored docs' scores are the oracle's
      const size_t n_cmp = std::min<size_t>(size_t(got_n), topx[q].size());
      REQUIRE(topx[q].size() == std::min<uint64_t>(kTop, std::min<uint64_t>(visited_total, uint64_t(got_n) + filled)));
      for (size_t i = 0; i < n_cmp; ++i) REQUIRE(close_rel(topx[q][i].score, want[i].score));
    }
    REQUIRE(filled_anywhere > 0);
    // only a strictly larger (docs_count, offset) key replaces the heap's root: segment 1's
    // (5, 0) leaves segment 0's (5, 0) alone, its (5, 1) replaces it (tests/test_oracle.py pins
    // the heap's own order on random ties against tests/cpp/collector_heap.cpp)
    const auto st = scored_states({{5, 5}, {5, 5}}, 2);
    REQUIRE((st == std::vector<std::pair<uint32_t, uint32_t>>{{0, 1}, {1, 1}}));
  }

  // adapter groundwork (format10::): the files' headers and footers, the term dictionary's
  // term_meta entries, the Norm2 column header (norm.cpp:107-115); tests/cpp/test_files.cpp
  // opens a whole segment from file bytes
  {
    size_t hdr = 0;
    REQUIRE(format10::check_header(a.doc, a.doc_len, format10::kDocFormatName, 0, 5, &hdr) == 5);
    REQUIRE(hdr < a.metas[0].doc_start + 1);
    REQUIRE(format10::check_header(a.pos, a.pos_len, format10::kPosFormatName, 0, 5) == 5);
    format10::check_footer(a.doc, a.doc_len);
    format10::check_footer(a.pos, a.pos_len);
    std::vector<uint8_t> bad(a.doc, a.doc + a.doc_len);
    bad[bad.size() / 2] ^= 1;   // a flipped bit in the postings: the checksum catches it
    bool threw = false;
    try {
      format10::check_footer(bad.data(), bad.size());
    } catch (const index_error&) {
      threw = true;
    }
    REQUIRE(threw);
    // the stats records of one dictionary block: written by the emitter from the writer
    // (synth_dict.cpp <- postings_writer_base::encode, formats_10.cpp:576-604), read back by
    // format10::decode_term_meta AND by the oracle's twin of the reader (dict_oracle.cpp <-
    // postings_reader_base::decode :3421-3456); the oracle's own encoder must produce the
    // same bytes.  (A real dictionary has no entry for a term without docs.)
    std::vector<irs_synth_term_meta> present;
    for (uint32_t t = 0; t < a.num_terms; ++t)
      if (a.metas[t].docs_count)
        present.push_back(reinterpret_cast<const irs_synth_term_meta&>(a.metas[t]));
    std::vector<uint8_t> dict(64 * present.size() + 64);
    const int64_t dict_len = irs_synth_term_meta_stream(present.data(), uint32_t(present.size()), 1, 1, 0,
                                                        dict.data(), dict.size());
    REQUIRE(dict_len > 0);
    dict.resize(size_t(dict_len));
    {
      std::vector<uint8_t> again(dict.size() + 64);
      orc_term_meta last;
      std::memset(&last, 0, sizeof last);
      size_t at = 0;
      for (const auto& m : present) {
        const int64_t n = orc_encode_term_meta(reinterpret_cast<const orc_term_meta*>(&m), &last, 1, 0,
                                               again.data() + at, again.size() - at);
        REQUIRE(n > 0);
        at += size_t(n);
      }
      REQUIRE(at == dict.size() && !std::memcmp(again.data(), dict.data(), at));
    }
    const uint8_t* p = dict.data();
    irs_hip_term_meta state{};
    orc_term_meta ostate;
    std::memset(&ostate, 0, sizeof ostate);
    ostate.pos_end = ~uint64_t(0);
    state.pos_end = ~uint64_t(0);
    for (const auto& m : present) {
      const int64_t used = orc_decode_term_meta(p, uint64_t(dict.data() + dict.size() - p), 1, 1, 0, &ostate);
      REQUIRE(used > 0);
      REQUIRE(format10::decode_term_meta(p, true, true, false, state) == size_t(used));
      p += used;
      REQUIRE(!std::memcmp(&state, &ostate, sizeof state));
      REQUIRE(state.docs_count == m.docs_count && state.freq == m.freq);
      REQUIRE(state.doc_start == m.doc_start && state.pos_start == m.pos_start);
      REQUIRE(m.freq <= 128 || state.pos_end == m.pos_end);
      REQUIRE((m.docs_count != 1 && m.docs_count <= 128) || state.e_skip_start == m.e_skip_start);
    }
    REQUIRE(p == dict.data() + dict.size());
    const uint8_t n2[10] = {0, 1, 0, 0, 0, 3, 0, 0, 0, 0xF0};
    format10::Norm2Header nh;
    REQUIRE(format10::read_norm2_header(n2, sizeof n2, nh));
    REQUIRE(nh.num_bytes == 1 && nh.min == 3 && nh.max == 0xF0 && nh.max_num_bytes() == 1);
    const uint8_t n4[10] = {0, 4, 0, 0, 0, 1, 0, 1, 0x86, 0xA0};
    REQUIRE(format10::read_norm2_header(n4, sizeof n4, nh) && nh.num_bytes == 4 && nh.max_num_bytes() == 4);
    const uint8_t nbad[10] = {0, 3, 0, 0, 0, 1, 0, 0, 0, 2};
    REQUIRE(!format10::read_norm2_header(nbad, sizeof nbad, nh));
  }

  // error behaviour: exceptions where the reference throws
  {
    bool threw = false;
    try {
      QueryBatch bad({a.reader.get()}, prepared, 0);  // top-0
    } catch (const illegal_argument&) {
      threw = true;
    }
    REQUIRE(threw);
    std::vector<uint8_t> corrupt(a.doc, a.doc + a.doc_len);
    corrupt[0] ^= 0xFF;  // format magic
    irs_hip_segment_desc d = a.desc();
    d.doc_file = corrupt.data();
    threw = false;
    try {
      SegmentReader r(d);
    } catch (const index_error&) {
      threw = true;
    }
    REQUIRE(threw);
    d = a.desc();
    d.device = 99;  // no such device
    threw = false;
    try {
      SegmentReader r(d);
    } catch (const io_error&) {
      threw = true;
    }
    REQUIRE(threw);
  }
  std::printf("test_host OK: %zu queries over 2 segments, %zu phrase\n", filters.size(), size_t(2));
  return 0;
}
