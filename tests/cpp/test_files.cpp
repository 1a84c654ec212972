// tests/cpp/test_files.cpp — TEST: a segment opened from NOTHING BUT FILE BYTES (SURVEY.md §8 f3,
// a7, a18): `.doc` (+ `.pos`), the term dictionary `.tm` and the columnstore pair `.csi` /
// `.csd`, all written by the emitter from the reference's WRITER code
// (iresearch_amd/index/synth_index.cpp, synth_dict.cpp).  The product's host readers
// (format10::walk_term_dictionary — without the term index —, read_fixed_column,
// describe_field; iresearch_amd/cpp/irs_hip.hpp) recover the term table and the Norm2 column;
// the oracle's readers (oracle/dict_oracle.cpp: the reference iterator's traversal from the root
// block) must yield the same; the segment then runs BASELINE config 2 — OR-of-2 BM25 top-100 —
// against the oracle's harness loop.  argv[1] = docs (1 M on the GPU, less on the emulator).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
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

static bool same_meta(const irs_hip_term_meta& a, const irs_synth_term_meta& b, bool has_pos) {
  if (a.docs_count != b.docs_count || a.freq != b.freq || a.doc_start != b.doc_start) return false;
  if (has_pos && (a.pos_start != b.pos_start || (b.freq > 128 && a.pos_end != b.pos_end))) return false;
  if (b.docs_count == 1) return uint32_t(a.e_skip_start) == uint32_t(b.e_skip_start);
  return b.docs_count <= 128 || a.e_skip_start == b.e_skip_start;
}

int main(int argc, char** argv) {
  const uint32_t docs = argc > 1 ? uint32_t(std::atoi(argv[1])) : 60000;
  const bool with_pos = argc > 2 && std::atoi(argv[2]) != 0;
  constexpr uint32_t kMaxRank = 4096, kTop = 100, kQueries = 40;
  irs_synth_params p{};
  p.seed = 20260926;
  p.num_docs = docs;
  p.vocab_log2 = 20;
  p.max_rank = kMaxRank;
  p.layout = IRS_SYNTH_LAYOUT_SIMD4;
  p.mean_len = 100;
  p.stddev_len = 30;
  p.with_positions = with_pos ? 1 : 0;
  irs_synth_index* idx = nullptr;
  REQUIRE(irs_synth_build(&p, &idx) == 0);
  uint64_t doc_len = 0, pos_len = 0, norm_count = 0;
  const uint8_t* doc = irs_synth_doc_bytes(idx, &doc_len);
  const uint8_t* pos = with_pos ? irs_synth_pos_bytes(idx, &pos_len) : nullptr;
  const uint8_t* norms = irs_synth_norms(idx, &norm_count);
  uint32_t num_terms = 0;
  const irs_synth_term_meta* metas = irs_synth_term_metas(idx, &num_terms);
  REQUIRE(num_terms == kMaxRank && norm_count == docs);

  // ---- the files, from the writer side -------------------------------------------------
  // terms: the rank's 4-byte big-endian ordinal (ascending like the ordinals; long shared
  // prefixes make a deep block tree with floor blocks)
  std::vector<uint8_t> term_bytes;
  std::vector<uint32_t> term_lens(num_terms, 4);
  for (uint32_t t = 0; t < num_terms; ++t)
    for (int s = 24; s >= 0; s -= 8) term_bytes.push_back(uint8_t(t >> s));
  std::vector<uint8_t> tm(64 * num_terms + 8192);
  uint64_t root = 0;
  const int64_t tm_len = irs_synth_term_dictionary(term_bytes.data(), term_lens.data(), metas, num_terms,
                                                   1, with_pos, 0, 25, 48, tm.data(), tm.size(), &root);
  REQUIRE(tm_len > 0);
  uint32_t lo = 255, hi = 0;
  for (uint64_t i = 0; i < norm_count; ++i) {
    lo = std::min<uint32_t>(lo, norms[i]);
    hi = std::max<uint32_t>(hi, norms[i]);
  }
  const uint8_t n2[10] = {0, 1, uint8_t(lo >> 24), uint8_t(lo >> 16), uint8_t(lo >> 8), uint8_t(lo),
                          uint8_t(hi >> 24), uint8_t(hi >> 16), uint8_t(hi >> 8), uint8_t(hi)};
  for (int dense = 0; dense < 2; ++dense) {   // a fresh segment's blocks / a consolidated one's single piece
    std::vector<uint8_t> csd(docs + 8192), csi(8192);
    uint64_t csd_len = 0, csi_len = 0;
    uint32_t column = 0;
    REQUIRE(irs_synth_columnstore(norms, 1, docs, 1, n2, sizeof n2, dense, 2, csd.data(), csd.size(),
                                  &csd_len, csi.data(), csi.size(), &csi_len, &column) == 0);

    // ---- the product's readers: nothing but the bytes -------------------------------------
    format10::FieldFiles files;
    files.doc = doc;  files.doc_len = doc_len;
    files.pos = pos;  files.pos_len = pos_len;
    files.tm = tm.data();  files.tm_len = uint64_t(tm_len);
    files.csi = csi.data();  files.csi_len = csi_len;
    files.csd = csd.data();  files.csd_len = csd_len;
    files.norm_column = column;
    files.num_docs = docs;
    format10::OpenedField field;
    const irs_hip_segment_desc desc = format10::describe_field(files, 0, field);
    // every rank with postings is there, in order, with the meta the writer recorded
    uint32_t present = 0;
    for (uint32_t t = 0; t < num_terms; ++t) present += metas[t].docs_count ? 1u : 0u;
    REQUIRE(field.terms.size() == present && desc.num_terms == present);
    std::vector<uint32_t> ordinal_of(num_terms, IRS_HIP_NO_TERM);   // rank - 1 -> ordinal in the dictionary
    for (uint32_t t = 0, o = 0; t < num_terms; ++t) {
      if (!metas[t].docs_count) continue;
      REQUIRE(field.terms[o].size() == 4 && !std::memcmp(field.terms[o].data(), &term_bytes[4 * t], 4));
      REQUIRE(same_meta(field.metas[o], metas[t], with_pos));
      ordinal_of[t] = o++;
    }
    REQUIRE(field.norms.value_bytes == 1 && field.norms.min_doc == 1 && field.norms.docs_count == docs);
    REQUIRE(!std::memcmp(field.norms.values.data(), norms, docs));
    REQUIRE(field.norm_header.min == lo && field.norm_header.max == hi && field.norm_header.max_num_bytes() == 1);
    REQUIRE(field.total_term_freq <= irs_synth_total_term_freq(idx));   // (ranks beyond max_rank are not indexed)

    // ---- the oracle's readers agree (the reference iterator's walk from the root block) ----
    if (!dense) {
      uint32_t on = 0;
      uint64_t obytes = 0;
      REQUIRE(orc_walk_term_dictionary(tm.data(), uint64_t(tm_len), root, 1, with_pos, 0, &on, &obytes,
                                       nullptr, nullptr, nullptr) == 0);
      REQUIRE(on == present && obytes == 4ull * present);
      std::vector<uint32_t> olens(on);
      std::vector<uint8_t> oterms(obytes);
      std::vector<orc_term_meta> ometas(on);
      REQUIRE(orc_walk_term_dictionary(tm.data(), uint64_t(tm_len), root, 1, with_pos, 0, &on, &obytes,
                                       olens.data(), oterms.data(), ometas.data()) == 0);
      for (uint32_t o = 0; o < on; ++o) {   // (in the iterator's order: ascending)
        REQUIRE(olens[o] == 4 && !std::memcmp(&oterms[4 * o], field.terms[o].data(), 4));
        REQUIRE(!std::memcmp(&ometas[o], &field.metas[o], sizeof(orc_term_meta)));
      }
    }
    {
      uint32_t vb = 0, mn = 0, dc = 0, pl = 0;
      uint8_t payload[16];
      std::vector<uint8_t> ovals(docs);
      REQUIRE(orc_read_fixed_column(csi.data(), csi_len, csd.data(), csd_len, column, &vb, &mn, &dc,
                                    payload, sizeof payload, &pl, ovals.data(), ovals.size()) == 0);
      REQUIRE(vb == 1 && mn == 1 && dc == docs && pl == 10 && !std::memcmp(payload, n2, 10));
      REQUIRE(ovals == field.norms.values);
    }
    // a flipped bit anywhere in the dictionary is refused (footer checksum)
    {
      std::vector<uint8_t> bad(tm.begin(), tm.begin() + tm_len);
      bad[bad.size() / 2] ^= 4;
      bool threw = false;
      try {
        format10::walk_term_dictionary(bad.data(), bad.size(), true, with_pos, false);
      } catch (const index_error&) {
        threw = true;
      }
      REQUIRE(threw);
    }
    if (dense) continue;   // (the query run once)

    // ---- BASELINE config 2 on the segment opened from files ---------------------------------
    SegmentReader reader(desc);
    std::vector<filter> filters;
    std::vector<uint32_t> ranks(kQueries * 2);
    REQUIRE(irs_synth_queries(20260926 + 1, kQueries, 2, 16, kMaxRank, ranks.data()) == 0);
    for (uint32_t q = 0; q < kQueries; ++q) {
      Or f;
      for (int t = 0; t < 2; ++t) f.subs.push_back(by_term{ordinal_of[ranks[q * 2 + t] - 1]});
      filters.push_back(f);
    }
    const SegmentStats stats{docs, field.total_term_freq, field.metas.data(), uint32_t(field.metas.size())};
    const BM25 scorer;
    const auto prepared = prepare(filters, scorer, {stats});
    QueryBatch batch({&reader}, prepared, kTop);
    const auto res = batch.run().results();
    orc_segment view{};
    view.doc_file = doc;
    view.doc_file_len = doc_len;
    view.layout = ORC_LAYOUT_SIMD4;
    view.num_docs = docs;
    view.norms = norms;
    view.norm_width = 1;
    view.pos_file = pos;
    view.pos_file_len = pos_len;
    const uint64_t dwf = docs, ttf = field.total_term_freq;
    const orc_scorer osc{ORC_SCORER_BM25, scorer.k(), scorer.b(), 0};
    for (uint32_t q = 0; q < kQueries; ++q) {
      orc_term_meta om[2];
      float boosts[2] = {1.f, 1.f};
      for (int t = 0; t < 2; ++t) std::memcpy(&om[t], &metas[ranks[q * 2 + t] - 1], sizeof om[t]);
      std::vector<orc_hit> want(kTop);
      uint64_t want_total = 0;
      const int64_t n = orc_search(&view, 1, om, 2, ORC_OP_OR, &osc, boosts, &dwf, &ttf, kTop,
                                   want.data(), &want_total);
      REQUIRE(n >= 0 && res.total(0, q) == want_total && res.count(0, q) == uint32_t(n));
      std::sort(want.begin(), want.begin() + n, [](const orc_hit& x, const orc_hit& y) { return x.score > y.score; });
      for (int64_t i = 0; i < n; ++i)
        REQUIRE(std::fabs(res.of(0, q)[i].score - want[size_t(i)].score) <= 1e-5f * std::fabs(want[size_t(i)].score));
    }
  }
  // ---- a dictionary of variable-length terms, some of them prefixes of others ("1", "10",
  // "100", "1000", "1001", ...): entries whose suffix is EMPTY inside the block of their own
  // prefix, blocks below blocks, min/max block sizes other than the defaults
  for (const auto& geometry : {std::pair<uint32_t, uint32_t>{25, 48}, {2, 2}, {3, 7}, {60, 200}}) {
    std::vector<std::string> words;
    for (uint32_t i = 0; i < 3000; ++i) words.push_back(std::to_string(i * 7u));
    std::sort(words.begin(), words.end());
    std::vector<uint8_t> blob;
    std::vector<uint32_t> lens;
    std::vector<irs_synth_term_meta> wm;
    for (size_t i = 0; i < words.size(); ++i) {
      blob.insert(blob.end(), words[i].begin(), words[i].end());
      lens.push_back(uint32_t(words[i].size()));
      irs_synth_term_meta m = metas[(i * 37u) % num_terms];
      if (!m.docs_count) m = metas[16];
      wm.push_back(m);
    }
    std::vector<uint8_t> wtm(64 * words.size() + 8192);
    uint64_t wroot = 0;
    const int64_t wlen = irs_synth_term_dictionary(blob.data(), lens.data(), wm.data(), uint32_t(words.size()),
                                                   1, with_pos, 0, geometry.first, geometry.second,
                                                   wtm.data(), wtm.size(), &wroot);
    REQUIRE(wlen > 0);
    const auto got = format10::walk_term_dictionary(wtm.data(), uint64_t(wlen), true, with_pos, false);
    REQUIRE(got.size() == words.size());
    uint32_t on = 0;
    uint64_t obytes = 0;
    REQUIRE(orc_walk_term_dictionary(wtm.data(), uint64_t(wlen), wroot, 1, with_pos, 0, &on, &obytes,
                                     nullptr, nullptr, nullptr) == 0 && on == words.size());
    std::vector<uint32_t> olens(on);
    std::vector<uint8_t> oterms(obytes);
    std::vector<orc_term_meta> ometas(on);
    REQUIRE(orc_walk_term_dictionary(wtm.data(), uint64_t(wlen), wroot, 1, with_pos, 0, &on, &obytes,
                                     olens.data(), oterms.data(), ometas.data()) == 0);
    size_t at = 0;
    for (size_t i = 0; i < words.size(); ++i) {
      REQUIRE(got[i].term == words[i] && same_meta(got[i].meta, wm[i], with_pos));
      REQUIRE(olens[i] == words[i].size() && !std::memcmp(&oterms[at], words[i].data(), olens[i]));
      REQUIRE(!std::memcmp(&ometas[i], &got[i].meta, sizeof(orc_term_meta)));
      at += olens[i];
    }
  }
  irs_synth_free(idx);
  std::printf("test_files OK: %u docs, %u terms from the dictionary, positions %d\n", docs, num_terms,
              int(with_pos));
  return 0;
}
