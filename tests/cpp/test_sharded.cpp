// tests/cpp/test_sharded.cpp — TEST: irs_hip_host::search_sharded (iresearch_amd/cpp/irs_hip.hpp):
// one process per rank, segments sharded, ONE all-gather through the C ABI's communicator
// (irs_hip_comm_* / irs_hip_topk_allgather), merge on every rank.  Run as N processes:
//   test_sharded <n_ranks> <rank> <id file>
// Rank 0 makes the communicator id and leaves it in the id file (what MPI_Bcast would carry);
// every rank checks the sharded result against the single-process search() over all segments.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <thread>
#include <vector>

#include "irs_hip.hpp"
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

struct Segment {
  irs_synth_index* idx = nullptr;
  irs_hip_segment_desc d{};
  std::unique_ptr<SegmentReader> reader;
  Segment(uint32_t docs, uint64_t first_doc, uint32_t max_rank) {
    irs_synth_params p{};
    p.seed = 20260926;
    p.first_doc = first_doc;
    p.num_docs = docs;
    p.vocab_log2 = 20;
    p.max_rank = max_rank;
    p.layout = IRS_SYNTH_LAYOUT_SIMD4;
    p.mean_len = 100;
    p.stddev_len = 30;
    if (irs_synth_build(&p, &idx) != 0) throw std::runtime_error("irs_synth_build");
    uint64_t n = 0;
    uint32_t m = 0;
    d.layout = IRS_HIP_LAYOUT_SIMD4;
    d.doc_file = irs_synth_doc_bytes(idx, &n);
    d.doc_file_len = n;
    d.num_docs = docs;
    d.has_freq = 1;
    d.norms = irs_synth_norms(idx, &n);
    d.norm_width = 1;
    d.norm_min_doc = 1;
    d.norm_count = n;
    d.terms = reinterpret_cast<const irs_hip_term_meta*>(irs_synth_term_metas(idx, &m));
    d.num_terms = m;
  }
  ~Segment() {
    reader.reset();
    irs_synth_free(idx);
  }
  void open() { reader = std::make_unique<SegmentReader>(d); }
  SegmentStats stats() const {
    return SegmentStats{irs_synth_docs_with_field(idx), irs_synth_total_term_freq(idx), d.terms,
                        d.num_terms};
  }
};

}  // namespace

int main(int argc, char** argv) {
  if (argc != 4) return 2;
  const int n_ranks = std::atoi(argv[1]), rank = std::atoi(argv[2]);
  const std::string id_file = argv[3];
  constexpr uint32_t kMaxRank = 128, kTop = 40, kSegs = 3, kPerRank = 2;
  REQUIRE(n_ranks == 2 && (rank == 0 || rank == 1));
  std::vector<std::unique_ptr<Segment>> segs;
  const uint32_t sizes[kSegs] = {12000, 5000, 9000};
  uint64_t first = 0;
  for (uint32_t s = 0; s < kSegs; ++s) {
    segs.push_back(std::make_unique<Segment>(sizes[s], first, kMaxRank));
    first += sizes[s];
  }
  // the communicator id: made on rank 0, carried by a file
  Communicator::Id id{};
  if (rank == 0) {
    id = Communicator::unique_id();
    std::ofstream(id_file + ".tmp", std::ios::binary).write(reinterpret_cast<const char*>(id.data()), id.size());
    std::rename((id_file + ".tmp").c_str(), id_file.c_str());
  } else {
    for (int tries = 0; tries < 20000; ++tries) {
      std::ifstream in(id_file, std::ios::binary);
      if (in && in.read(reinterpret_cast<char*>(id.data()), id.size())) break;
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
  }
  Communicator comm(0, id, n_ranks, rank);

  std::vector<filter> filters;
  uint32_t ranks[3 * 8];
  REQUIRE(irs_synth_queries(20260928, 3, 8, 2, kMaxRank, ranks) == 0);
  for (int q = 0; q < 3; ++q) {
    Or f;
    for (int t = 0; t < 8; ++t) f.subs.push_back(by_term{ranks[q * 8 + t] - 1});
    filters.push_back(f);
  }
  filters.push_back(by_term{7});
  filters.push_back(And{{by_term{1}, by_term{20}, by_term{3, 2.5f}}});
  filters.push_back(Or{{by_term{2}, by_term{9}, by_term{30}, by_term{5}}, 2});
  const BM25 scorer;
  std::vector<SegmentStats> index;
  for (const auto& s : segs) index.push_back(s->stats());

  // this rank's block of segments
  std::vector<const SegmentReader*> mine;
  for (uint32_t s = rank * kPerRank; s < std::min(kSegs, (rank + 1) * kPerRank); ++s) {
    segs[s]->open();
    mine.push_back(segs[s]->reader.get());
  }
  const auto sharded = search_sharded(comm, mine, kPerRank, kSegs, index, filters, scorer, kTop);
  const auto sharded_wand = search_sharded(comm, mine, kPerRank, kSegs, index, filters, scorer, kTop, true);

  // the single-process answer over ALL segments (harness loop, index-search.cpp:719-787)
  std::vector<const SegmentReader*> all;
  for (auto& s : segs) {
    if (!s->reader) s->open();
    all.push_back(s->reader.get());
  }
  const auto want = search(all, index, filters, scorer, kTop);
  REQUIRE(sharded.size() == want.size() && sharded_wand.size() == want.size());
  for (size_t q = 0; q < want.size(); ++q) {
    REQUIRE(sharded[q].size() == want[q].size());
    REQUIRE(sharded_wand[q].size() == want[q].size());
    for (size_t i = 0; i < want[q].size(); ++i) {
      REQUIRE(sharded[q][i].score == want[q][i].score && sharded[q][i].doc == want[q][i].doc &&
              sharded[q][i].segment == want[q][i].segment);
      // (block-max pruning runs on the work-item / block-driven kernels; without it the And and
      // the min-match query take the joined streams, whose match counts round each posting's
      // fixed-point contribution: same docs, scores within the parity tolerance)
      REQUIRE(std::fabs(sharded_wand[q][i].score - want[q][i].score) <= 1e-5f * want[q][i].score &&
              sharded_wand[q][i].doc == want[q][i].doc &&
              sharded_wand[q][i].segment == want[q][i].segment);
    }
  }
  // a rank WITHOUT segments (3 per rank: rank 0 holds all three, rank 1 none) builds no batch and
  // still takes part in the exchange
  {
    std::vector<const SegmentReader*> lot;
    if (rank == 0) lot = all;
    const auto lopsided = search_sharded(comm, lot, kSegs, kSegs, index, filters, scorer, kTop);
    REQUIRE(lopsided.size() == want.size());
    for (size_t q = 0; q < want.size(); ++q) {
      REQUIRE(lopsided[q].size() == want[q].size());
      for (size_t i = 0; i < want[q].size(); ++i)
        REQUIRE(lopsided[q][i].score == want[q][i].score && lopsided[q][i].doc == want[q][i].doc &&
                lopsided[q][i].segment == want[q][i].segment);
    }
  }
  std::printf("test_sharded OK: rank %d of %d, %zu queries over %u segments\n", rank, n_ranks,
              filters.size(), kSegs);
  return 0;
}
