// tests/cpp/collector_heap.cpp — TEST INFRASTRUCTURE: the container algorithm of
// limited_sample_collector<term_frequency>::collect (core/search/limited_sample_collector.hpp
// :67-120, 226-240) on the standard library's own std::push_heap / std::pop_heap, so that the
// oracle's and the product's restatements of "which of two equal keys is replaced" are pinned
// against what the reference's compiled code does (libstdc++).  stdin: limit, n_segments, then per
// segment n and n docs_count values; stdout: the (segment, offset) pairs left scored, sorted.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <utility>
#include <vector>

struct Key {
  uint32_t offset, frequency;
  bool operator<(const Key& r) const {
    return frequency < r.frequency || (frequency == r.frequency && offset < r.offset);
  }
};
struct State {
  Key key;
  uint32_t segment;
};

int main() {
  unsigned long limit = 0, n_segs = 0;
  if (std::scanf("%lu %lu", &limit, &n_segs) != 2) return 1;
  std::vector<State> states;
  std::vector<size_t> heap;
  auto comp = [&](size_t lhs, size_t rhs) { return states[rhs].key < states[lhs].key; };
  for (uint32_t s = 0; s < n_segs; ++s) {
    unsigned long n = 0;
    if (std::scanf("%lu", &n) != 1) return 1;
    for (uint32_t off = 0; off < n; ++off) {
      unsigned long f = 0;
      if (std::scanf("%lu", &f) != 1) return 1;
      const Key key{off, uint32_t(f)};
      if (!limit) continue;
      if (states.size() < limit) {
        heap.emplace_back(states.size());
        states.push_back(State{key, s});
        std::push_heap(heap.begin(), heap.end(), comp);
        continue;
      }
      const size_t min_idx = heap.front();
      if (states[min_idx].key < key) {
        std::pop_heap(heap.begin(), heap.end(), comp);
        states[min_idx] = State{key, s};
        std::push_heap(heap.begin(), heap.end(), comp);
      }
    }
  }
  std::vector<std::pair<uint32_t, uint32_t>> out;
  for (const State& st : states) out.emplace_back(st.segment, st.key.offset);
  std::sort(out.begin(), out.end());
  for (auto& p : out) std::printf("%u %u\n", p.first, p.second);
  return 0;
}
