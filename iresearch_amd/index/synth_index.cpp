// Synthetic IResearch segment builder — see synth_index.h.
//
// This file is an independent emitter of the reference's on-disk posting
// format (the "writer side" of SURVEY.md §8 a1/a6/a7); every routine cites the
// reference lines whose byte layout it reproduces.  It shares no code with
// oracle/ (which restates the *reader* side) so that "writer -> oracle reader"
// round trips are a real cross-check.
#include "synth_index.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

namespace {

constexpr uint32_t kBlock = 128;        // formats_10.cpp:90 block_size()
constexpr uint32_t kSkipN = 8;          // formats_10.cpp:323
constexpr uint32_t kMaxSkipLevels = 9;  // formats_10.cpp:322
constexpr uint32_t kDocMin = 1;         // doc_limits::min(), type_limits.hpp:45

using Bytes = std::vector<uint8_t>;

// LEB128, LSB first — bytes_utils.hpp:125-134 (vwrite<uint32_t>)
inline void put_vint(Bytes& o, uint32_t v) {
  while (v >= 0x80) {
    o.push_back(static_cast<uint8_t>(v | 0x80));
    v >>= 7;
  }
  o.push_back(static_cast<uint8_t>(v));
}
inline void put_vlong(Bytes& o, uint64_t v) {
  while (v >= 0x80) {
    o.push_back(static_cast<uint8_t>(v | 0x80));
    v >>= 7;
  }
  o.push_back(static_cast<uint8_t>(v));
}
// big-endian fixed ints — bytes_utils.hpp:137-146
inline void put_be32(Bytes& o, uint32_t v) {
  o.push_back(uint8_t(v >> 24));
  o.push_back(uint8_t(v >> 16));
  o.push_back(uint8_t(v >> 8));
  o.push_back(uint8_t(v));
}
inline void put_be64(Bytes& o, uint64_t v) {
  put_be32(o, uint32_t(v >> 32));
  put_be32(o, uint32_t(v));
}

inline uint32_t bits_of(uint32_t v) { return v ? 32u - __builtin_clz(v) : 0u; }

// Scalar layout ("1_0".."1_5"): format_traits::pack_block packs four
// 32-value sub-blocks of `bits` words each, back to back
// (formats_10.cpp:96-105); within a sub-block value i occupies bits
// [i*b, i*b+b) LSB-first (bit_packing.cpp fastpack<N>), so the whole block is
// one little-endian bitstream with value j at bit j*b.
void pack_scalar(const uint32_t* v, uint32_t b, uint32_t* out /*4*b words*/) {
  std::memset(out, 0, 16u * b);
  for (uint32_t j = 0; j < kBlock; ++j) {
    const uint64_t bit = uint64_t(j) * b;
    const uint32_t w = uint32_t(bit >> 5), s = uint32_t(bit & 31);
    const uint64_t x = uint64_t(v[j]) << s;
    out[w] |= uint32_t(x);
    if (s + b > 32) out[w + 1] |= uint32_t(x >> 32);
  }
}

// simdcomp vertical layout ("1_2simd".."1_5simd"): simdpackwithoutmask
// (external/simdcomp/src/simdbitpacking.c) treats the 128 values as 32 rows
// of one __m128i; SSE lane l of row r is value 4r+l and is OR-ed into lane l's
// own bitstream at bit r*b.  Stream word k of lane l is u32 index 4k+l.
void pack_simd4(const uint32_t* v, uint32_t b, uint32_t* out /*4*b words*/) {
  std::memset(out, 0, 16u * b);
  for (uint32_t j = 0; j < kBlock; ++j) {
    const uint32_t r = j >> 2, l = j & 3;
    const uint32_t bit = r * b;
    const uint32_t k = bit >> 5, s = bit & 31;
    const uint64_t x = uint64_t(v[j]) << s;
    out[4 * k + l] |= uint32_t(x);
    if (s + b > 32) out[4 * (k + 1) + l] |= uint32_t(x >> 32);
  }
}

// bitpack::write_block32 — bitpack.hpp:75-108
void write_block(Bytes& o, const uint32_t* v, uint32_t layout) {
  bool all_equal = true;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < kBlock; ++i) {
    all_equal &= (v[i] == v[0]);
    acc |= v[i];
  }
  if (all_equal) {
    o.push_back(0);  // ALL_EQUAL
    put_vint(o, v[0]);
    return;
  }
  const uint32_t b = bits_of(acc);
  uint32_t buf[kBlock];
  if (layout == IRS_SYNTH_LAYOUT_SIMD4) {
    pack_simd4(v, b, buf);
  } else {
    pack_scalar(v, b, buf);
  }
  o.push_back(uint8_t(b));
  const size_t at = o.size();
  o.resize(at + 16u * b);
  std::memcpy(o.data() + at, buf, 16u * b);  // host is little endian
}

// math::log (math_utils.hpp:109-116) and CountMaxLevels (skip_list.cpp:38-41)
uint32_t ilog(uint64_t x, uint64_t base) {
  uint32_t r = 0;
  while (x >= base) {
    x /= base;
    ++r;
  }
  return r;
}
uint32_t count_max_levels(uint32_t skip0, uint32_t skipn, uint64_t count) {
  return skip0 < count ? 1 + ilog(count / skip0, skipn) : 0;
}

// ---- wand data (formats 1_4 / 1_5 written with scorers) ------------------
// FreqNormProducer<Tag> — wand_writer.hpp:152-342.  One accumulator per skip level
// (+1 for the whole list), per scorer; the payload of an entry is
// vint(freq) [+ vint(norm - freq) when the tag carries a norm and norm != freq].
struct WandEntry {
  uint32_t freq = 1;
  uint32_t norm = 0xFFFFFFFFu;
};
struct WandSpec {
  const uint8_t* norms;     // 1-byte Norm2 column, norms[doc - 1]; needed by MinNorm / DivNorm
  uint32_t count;           // scorers the field was indexed with
  const uint32_t* kinds;    // IRS_SYNTH_WAND_* per scorer
};
// Produce(const Entry& from, Entry& to) — :171-195; also Produce(Entry& to) from a doc :258-291
void wand_produce(uint32_t kind, uint32_t freq, uint32_t norm, WandEntry& to) {
  if (kind == IRS_SYNTH_WAND_DIV_NORM) {
    if (uint64_t(freq) * to.norm > uint64_t(to.freq) * norm) {
      to.freq = freq;
      to.norm = norm;
    }
    return;
  }
  if (freq > to.freq) to.freq = freq;
  if (kind == IRS_SYNTH_WAND_MIN_NORM) {
    if (norm < to.norm) to.norm = norm;
    if (to.norm < to.freq) to.norm = to.freq;
  }
}
uint32_t vint_size(uint32_t v) {
  uint32_t n = 1;
  while (v >= 0x80) { v >>= 7; ++n; }
  return n;
}
// Size(Entry) :210-220 and Write(Entry, out) :197-208
uint8_t wand_size(uint32_t kind, const WandEntry& e) {
  uint32_t n = vint_size(e.freq);
  if (kind != IRS_SYNTH_WAND_MAX_FREQ && e.norm != e.freq) n += vint_size(e.norm - e.freq);
  return uint8_t(n);
}
void wand_write(uint32_t kind, const WandEntry& e, Bytes& o) {
  put_vint(o, e.freq);
  if (kind != IRS_SYNTH_WAND_MAX_FREQ && e.norm != e.freq) put_vint(o, e.norm - e.freq);
}

// One term: postings_writer<>::write + BeginDocument + EndDocument + EndTerm
// (formats_10.cpp:942-1025, 865-891, 639-657, 662-798) for a FREQ-only field
// written with no scorers (no wand bytes: valid_writers_ empty, :453-458).
// With `positions` (the field has IndexFeatures::POS; Σ freqs entries, ascending and >= 1
// within a doc) the `.pos` stream of the term is appended to `*pos`: AddPosition :894-933
// (delta to the previous position of the doc, first one relative to pos_min() == 0 in the
// zero-based formats 1_3+, packed in blocks of 128 across docs), EndTerm :713-760 (vint tail,
// pos_end), and every skip entry also carries vint(pend_pos) + vlong(Δpos_ptr) (:511-518).
void encode_term(const uint32_t* docs, const uint32_t* freqs, uint32_t count,
                 uint32_t segment_docs, uint32_t layout, Bytes& o,
                 irs_synth_term_meta& meta, const WandSpec* wand = nullptr,
                 const uint32_t* positions = nullptr, Bytes* pos = nullptr,
                 bool one_based = false) {
  const size_t start = o.size();
  meta = irs_synth_term_meta{};
  meta.pos_end = UINT64_MAX;
  meta.docs_count = count;
  if (count == 0) return;
  const bool has_pos = positions && pos && freqs;
  const uint64_t pos_start = has_pos ? pos->size() : 0;  // BeginTerm :626
  uint64_t pos_skip_ptr[kMaxSkipLevels];
  std::fill_n(pos_skip_ptr, kMaxSkipLevels, pos_start);   // :627
  uint32_t pos_buf[kBlock];
  uint32_t pos_n = 0;           // pos_.size
  uint32_t pos_block_last = 0;  // pos_.block_last: positions pending at the last doc-block end
  uint64_t pos_at = 0;          // cursor in `positions`

  // SkipWriter::Prepare — skip_list.cpp:47-48
  const uint32_t max_levels = std::min(
    kMaxSkipLevels, count_max_levels(kBlock, kSkipN, segment_docs));
  std::vector<Bytes> levels(max_levels);
  uint64_t skip_ptr[kMaxSkipLevels];
  std::fill_n(skip_ptr, kMaxSkipLevels, uint64_t(start));  // BeginTerm :623

  uint32_t block_last = kDocMin;  // BeginTerm :636
  uint32_t buf_docs[kBlock], buf_freqs[kBlock];
  uint32_t n = 0;          // entries buffered
  uint32_t last = 0;       // doc_.last (invalid)
  uint64_t total_freq = 0;

  // WandWriterImpl::levels_ — wand_writer.hpp:40-98: [scorer][level], level max_levels = root
  const uint32_t wn = wand ? wand->count : 0;
  std::vector<std::vector<WandEntry>> wl(wn, std::vector<WandEntry>(max_levels + 1));

  // postings_writer_base::WriteSkip — :501-533 (no POS features), then, in formats with
  // wand data, one size byte per scorer followed by the payloads (:990-999);
  // WandWriterImpl::Write folds the entry into the level above and clears it (:61-67)
  auto write_skip = [&](uint32_t level, Bytes& out) {
    const uint64_t doc_ptr = o.size();
    put_vint(out, block_last);
    put_vlong(out, doc_ptr - skip_ptr[level]);
    skip_ptr[level] = doc_ptr;
    if (has_pos) {  // :511-518
      const uint64_t pos_ptr = pos->size();
      put_vint(out, pos_block_last);
      put_vlong(out, pos_ptr - pos_skip_ptr[level]);
      pos_skip_ptr[level] = pos_ptr;
    }
    for (uint32_t w = 0; w < wn; ++w) out.push_back(wand_size(wand->kinds[w], wl[w][level]));
    for (uint32_t w = 0; w < wn; ++w) {
      WandEntry& e = wl[w][level];
      wand_produce(wand->kinds[w], e.freq, e.norm, wl[w][level + 1]);
      wand_write(wand->kinds[w], e, out);
      e = WandEntry{};
    }
  };
  // EndTerm's write_max_score(level) — :669-675; SizeRoot cascades the levels below (:80-88)
  auto write_max_score = [&](uint32_t level) {
    for (uint32_t w = 0; w < wn; ++w) {
      for (uint32_t l = 0; l < level; ++l)
        wand_produce(wand->kinds[w], wl[w][l].freq, wl[w][l].norm, wl[w][l + 1]);
      o.push_back(wand_size(wand->kinds[w], wl[w][level]));
    }
    for (uint32_t w = 0; w < wn; ++w) wand_write(wand->kinds[w], wl[w][level], o);
  };

  for (uint32_t i = 0; i < count; ++i) {
    // :987-1002 — skip entry is emitted when the NEXT doc arrives
    if (last != 0 && n == 0) {
      uint32_t c = i;  // docs_count so far
      // SkipWriter::Skip — skip_list.hpp:91-117
      if (c % kBlock == 0 && max_levels) {
        write_skip(0, levels[0]);
        c /= kBlock;
        uint64_t child = levels[0].size();
        for (uint32_t l = 1; c % kSkipN == 0 && l < max_levels;
             ++l, c /= kSkipN) {
          write_skip(l, levels[l]);
          const uint64_t next_child = levels[l].size();
          put_vlong(levels[l], child);
          child = next_child;
        }
      }
    }
    // BeginDocument :865-878
    buf_docs[n] = docs[i];
    buf_freqs[n] = freqs ? freqs[i] : 1u;
    last = docs[i];
    ++n;
    total_freq += freqs ? freqs[i] : 0u;
    for (uint32_t w = 0; w < wn; ++w)  // writer.Update() :1005-1006
      wand_produce(wand->kinds[w], freqs ? freqs[i] : 1u,
                   wand->norms ? wand->norms[docs[i] - 1] : 0xFFFFFFFFu, wl[w][0]);
    const bool doc_block_full = n == kBlock;
    if (n == kBlock) {
      // simd::delta_encode<128>(docs, block_last) — simd_utils.hpp:200-249
      uint32_t prev = block_last;
      for (uint32_t k = 0; k < kBlock; ++k) {
        const uint32_t cur = buf_docs[k];
        buf_docs[k] = cur - prev;
        prev = cur;
      }
      write_block(o, buf_docs, layout);
      if (freqs) write_block(o, buf_freqs, layout);  // only fields with IndexFeatures::FREQ (:875-877)
      // EndDocument :639-657
      block_last = last;
      n = 0;
    }
    if (has_pos) {
      // FormatTraits::pos_min() (:883): 0 in the zero-based formats 1_3+ (:4196), pos_limits::min()
      // in 1_0..1_2 (:4161), whose reader adds it back before the first delta (:1623-1625)
      uint32_t pos_last = one_based ? 1u : 0u;
      for (uint32_t k = 0; k < freqs[i]; ++k) {  // AddPosition :894-913
        const uint32_t p = positions[pos_at++];
        pos_buf[pos_n++] = p - pos_last;
        pos_last = p;
        if (pos_n == kBlock) {
          write_block(*pos, pos_buf, layout);
          pos_n = 0;
        }
      }
      if (doc_block_full) pos_block_last = pos_n;  // EndDocument :645-649
    }
  }
  meta.freq = uint32_t(total_freq);
  if (has_pos) {  // EndTerm :713-760
    if (meta.freq > kBlock) meta.pos_end = pos->size() - pos_start;
    for (uint32_t k = 0; k < pos_n; ++k) put_vint(*pos, pos_buf[k]);
    meta.pos_start = pos_start;
  }

  // EndTerm :662-798
  if (count == 1) {
    meta.e_skip_start = uint64_t(buf_docs[0] - kDocMin);  // e_single_doc :677
  } else {
    if (!(kBlock < count)) write_max_score(0);  // no skip list: wand root before the tail :686-688
    uint32_t prev = block_last;
    for (uint32_t k = 0; k < n; ++k) {  // tail :689-704
      const uint32_t delta = buf_docs[k] - prev;
      if (!freqs) {
        put_vint(o, delta);  // field without FREQ (:705-709)
      } else if (buf_freqs[k] == 1) {
        put_vint(o, (delta << 1) | 1u);
      } else {
        put_vint(o, delta << 1);
        put_vint(o, buf_freqs[k]);
      }
      prev = buf_docs[k];
    }
    if (kBlock < count) {  // has_skip_list :667, :776-781
      meta.e_skip_start = o.size() - start;
      // SkipWriter::CountLevels — skip_list.cpp:62-74
      uint32_t num_levels = max_levels;
      while (num_levels && levels[num_levels - 1].empty()) --num_levels;
      write_max_score(num_levels);  // wand root of the whole list :778
      // FlushLevels — skip_list.cpp:76-92
      put_vint(o, num_levels);
      for (uint32_t l = num_levels; l-- > 0;) {
        put_vlong(o, levels[l].size());
        o.insert(o.end(), levels[l].begin(), levels[l].end());
      }
    }
  }
  meta.doc_start = start;
}

// CRC-32C (Castagnoli), as absl::ExtendCrc32c from 0 — utils/crc.hpp:38-41
uint32_t crc32c(const uint8_t* p, size_t n) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1)));
      table[i] = c;
    }
    init = true;
  }
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

constexpr char kDocFormatName[] = "iresearch_10_postings_documents";  // :325
constexpr char kPosFormatName[] = "iresearch_10_postings_positions";  // :328

void make_header(Bytes& o, uint32_t layout, const char* name = kDocFormatName,
                 bool one_based = false) {
  const size_t len = std::strlen(name);
  put_be32(o, 0x3fd76c17u);  // kFormatMagic format_utils.hpp:36
  put_vint(o, uint32_t(len));
  o.insert(o.end(), name, name + len);
  // PostingsFormat::WAND_SSE (5) / WAND (4) — formats_10.cpp:305-311
  // ... or POSITIONS_ONEBASED_SSE (1) / POSITIONS_ONEBASED (0), formats 1_2simd / 1_0 (:283-296)
  put_be32(o, (one_based ? 0u : 4u) + (layout == IRS_SYNTH_LAYOUT_SIMD4 ? 1u : 0u));
}

// ---------------------------------------------------------------- corpus --

inline uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// Irwin-Hall(12) approximation of Normal(mean, sd), integer only.
uint32_t doc_length(uint64_t seed, uint64_t gdoc, uint32_t mean, uint32_t sd) {
  uint64_t s = 0;
  for (uint32_t i = 0; i < 3; ++i) {
    uint64_t h = mix64(seed ^ mix64(gdoc * 4 + i + 0x5bd1e995ull));
    s += (h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48);
  }
  // z = (s - 6*65536)/65536 ~ N(0,1); L = round(mean + sd*z)
  const int64_t num = int64_t(mean) * 65536 +
                      int64_t(sd) * (int64_t(s) - 6 * 65536) + 32768;
  int64_t len = num >> 16;
  if (len < 1) len = 1;
  if (len > 255) len = 255;
  return uint32_t(len);
}

inline uint64_t token_hash(uint64_t seed, uint64_t gdoc, uint32_t pos) {
  return mix64((seed * 0xD6E8FEB86659FD93ull) ^ mix64((gdoc << 8) | pos));
}

struct Zipf {
  // thresholds[r-1] = floor(H_r/H_V * 2^64); rank = first r with h < thr[r-1]
  std::vector<uint64_t> thr;
  std::vector<uint32_t> guide;  // by top 20 bits of h: first candidate index
  uint32_t limit;               // max_rank

  Zipf(uint32_t vocab_log2, uint32_t max_rank) : limit(max_rank) {
    const uint64_t V = 1ull << vocab_log2;
    double hv = 0.0;
    for (uint64_t r = 1; r <= V; ++r) hv += 1.0 / double(r);
    thr.resize(max_rank);
    double h = 0.0;
    for (uint32_t r = 1; r <= max_rank; ++r) {
      h += 1.0 / double(r);
      const double f = h / hv;
      thr[r - 1] = (uint64_t(r) == V || f >= 1.0)
                     ? UINT64_MAX
                     : uint64_t(f * 18446744073709551616.0);
    }
    guide.resize(1u << 20);
    uint32_t idx = 0;
    for (uint32_t g = 0; g < (1u << 20); ++g) {
      const uint64_t lo = uint64_t(g) << 44;
      while (idx < max_rank && thr[idx] <= lo) ++idx;
      guide[g] = idx;
    }
  }
  // returns rank in [1,limit] or 0 when the token's rank is > limit
  inline uint32_t sample(uint64_t h) const {
    if (h >= thr[limit - 1]) return 0;
    uint32_t i = guide[h >> 44];
    while (thr[i] <= h) ++i;
    return i + 1;
  }
};

struct ThreadPostings {
  std::vector<std::vector<uint32_t>> docs;
  std::vector<std::vector<uint8_t>> tfs;
  std::vector<std::vector<uint8_t>> poss;  // with_positions: 1-based token positions (<= 255)
};

}  // namespace

struct irs_synth_index {
  Bytes doc_file;
  Bytes pos_file;  // with_positions
  std::vector<uint8_t> norms;
  std::vector<irs_synth_term_meta> metas;
  uint64_t docs_with_field = 0;
  uint64_t total_term_freq = 0;
  // keep_postings
  std::vector<std::vector<uint32_t>> docs;
  std::vector<std::vector<uint32_t>> freqs;
  std::vector<std::vector<uint32_t>> positions;
};

extern "C" {

uint32_t irs_synth_doc_length(uint64_t seed, uint64_t global_doc,
                              uint32_t mean_len, uint32_t stddev_len) {
  return doc_length(seed, global_doc, mean_len, stddev_len);
}

int64_t irs_synth_encode_term(const uint32_t* docs, const uint32_t* freqs,
                              uint32_t count, uint32_t segment_docs,
                              uint32_t layout, uint8_t* out, uint64_t out_cap,
                              irs_synth_term_meta* meta) {
  if (!meta || (count && !docs)) return -1;  // freqs == NULL: a field without FREQ
  for (uint32_t i = 0; i < count; ++i) {
    // formats_10.cpp:866, 885-889: docs must be strictly ascending and valid
    if (docs[i] < kDocMin || (i && docs[i] <= docs[i - 1]) || (freqs && freqs[i] == 0))
      return -1;
  }
  Bytes o;
  encode_term(docs, freqs, count, segment_docs, layout, o, *meta);
  if (o.size() > out_cap) return -2;
  if (!o.empty()) std::memcpy(out, o.data(), o.size());
  return int64_t(o.size());
}

int64_t irs_synth_encode_term_wand(const uint32_t* docs, const uint32_t* freqs,
                                   uint32_t count, uint32_t segment_docs,
                                   uint32_t layout, const uint8_t* norms,
                                   const uint32_t* wand_kinds, uint32_t wand_count,
                                   uint8_t* out, uint64_t out_cap,
                                   irs_synth_term_meta* meta) {
  if (!meta || (count && !docs) || wand_count > 8 || (wand_count && !wand_kinds)) return -1;
  bool uses_norms = false;
  for (uint32_t w = 0; w < wand_count; ++w) {
    if (wand_kinds[w] > IRS_SYNTH_WAND_DIV_NORM) return -1;
    uses_norms |= wand_kinds[w] != IRS_SYNTH_WAND_MAX_FREQ;
  }
  if (uses_norms && !norms) return -1;
  for (uint32_t i = 0; i < count; ++i) {
    if (docs[i] < kDocMin || docs[i] > segment_docs || (i && docs[i] <= docs[i - 1]) ||
        (freqs && freqs[i] == 0))
      return -1;
    // a doc's field length is at least the term's frequency in it (wand_writer.hpp:201)
    if (uses_norms && freqs && norms[docs[i] - 1] < freqs[i]) return -1;
  }
  const WandSpec spec{norms, wand_count, wand_kinds};
  Bytes o;
  encode_term(docs, freqs, count, segment_docs, layout, o, *meta, wand_count ? &spec : nullptr);
  if (o.size() > out_cap) return -2;
  if (!o.empty()) std::memcpy(out, o.data(), o.size());
  return int64_t(o.size());
}

int64_t irs_synth_wrap_doc_file(const uint8_t* body, uint64_t body_len,
                                uint32_t layout, uint8_t* out,
                                uint64_t out_cap, uint64_t* body_offset) {
  Bytes o;
  make_header(o, layout);
  const uint64_t hdr = o.size();
  o.insert(o.end(), body, body + body_len);
  // write_footer — format_utils.cpp:63-67
  put_be32(o, uint32_t(-int32_t(0x3fd76c17)));
  put_be32(o, 0);
  put_be64(o, crc32c(o.data(), o.size()));
  if (o.size() > out_cap) return -2;
  std::memcpy(out, o.data(), o.size());
  if (body_offset) *body_offset = hdr;
  return int64_t(o.size());
}

int64_t irs_synth_wrap_file(const uint8_t* body, uint64_t body_len, uint32_t layout,
                            uint32_t is_pos, uint32_t one_based, uint8_t* out, uint64_t out_cap,
                            uint64_t* body_offset) {
  Bytes o;
  make_header(o, layout, is_pos ? kPosFormatName : kDocFormatName, one_based != 0);
  const uint64_t hdr = o.size();
  o.insert(o.end(), body, body + body_len);
  put_be32(o, uint32_t(-int32_t(0x3fd76c17)));
  put_be32(o, 0);
  put_be64(o, crc32c(o.data(), o.size()));
  if (o.size() > out_cap) return -2;
  std::memcpy(out, o.data(), o.size());
  if (body_offset) *body_offset = hdr;
  return int64_t(o.size());
}

int64_t irs_synth_wrap_pos_file(const uint8_t* body, uint64_t body_len,
                                uint32_t layout, uint8_t* out,
                                uint64_t out_cap, uint64_t* body_offset) {
  Bytes o;
  make_header(o, layout, kPosFormatName);
  const uint64_t hdr = o.size();
  o.insert(o.end(), body, body + body_len);
  put_be32(o, uint32_t(-int32_t(0x3fd76c17)));
  put_be32(o, 0);
  put_be64(o, crc32c(o.data(), o.size()));
  if (o.size() > out_cap) return -2;
  std::memcpy(out, o.data(), o.size());
  if (body_offset) *body_offset = hdr;
  return int64_t(o.size());
}

int64_t irs_synth_encode_term_pos(const uint32_t* docs, const uint32_t* freqs,
                                  const uint32_t* positions, uint32_t count,
                                  uint32_t segment_docs, uint32_t layout,
                                  const uint8_t* norms, const uint32_t* wand_kinds,
                                  uint32_t wand_count, uint8_t* out, uint64_t out_cap,
                                  uint8_t* pos_out, uint64_t pos_cap, uint64_t* pos_len,
                                  irs_synth_term_meta* meta) {
  return irs_synth_encode_term_pos_v(docs, freqs, positions, count, segment_docs, layout, norms,
                                     wand_kinds, wand_count, 0, out, out_cap, pos_out, pos_cap,
                                     pos_len, meta);
}

int64_t irs_synth_encode_term_pos_v(const uint32_t* docs, const uint32_t* freqs,
                                    const uint32_t* positions, uint32_t count,
                                    uint32_t segment_docs, uint32_t layout,
                                    const uint8_t* norms, const uint32_t* wand_kinds,
                                    uint32_t wand_count, uint32_t one_based, uint8_t* out,
                                    uint64_t out_cap, uint8_t* pos_out, uint64_t pos_cap,
                                    uint64_t* pos_len, irs_synth_term_meta* meta) {
  if (one_based && wand_count) return -1;  // the one-based formats predate wand data
  if (!meta || !freqs || !positions || !pos_len || (count && !docs) || wand_count > 8 ||
      (wand_count && !wand_kinds))
    return -1;
  bool uses_norms = false;
  for (uint32_t w = 0; w < wand_count; ++w) {
    if (wand_kinds[w] > IRS_SYNTH_WAND_DIV_NORM) return -1;
    uses_norms |= wand_kinds[w] != IRS_SYNTH_WAND_MAX_FREQ;
  }
  if (uses_norms && !norms) return -1;
  uint64_t at = 0;
  for (uint32_t i = 0; i < count; ++i) {
    if (docs[i] < kDocMin || docs[i] > segment_docs || (i && docs[i] <= docs[i - 1]) ||
        freqs[i] == 0)
      return -1;
    if (uses_norms && norms[docs[i] - 1] < freqs[i]) return -1;
    for (uint32_t k = 0; k < freqs[i]; ++k, ++at) {  // pos_limits::valid, ascending (:1012)
      if (positions[at] == 0 || positions[at] == UINT32_MAX ||
          (k && positions[at] <= positions[at - 1]))
        return -1;
    }
  }
  const WandSpec spec{norms, wand_count, wand_kinds};
  Bytes o, po;
  encode_term(docs, freqs, count, segment_docs, layout, o, *meta,
              wand_count ? &spec : nullptr, positions, &po, one_based != 0);
  if (o.size() > out_cap || po.size() > pos_cap) return -2;
  if (!o.empty()) std::memcpy(out, o.data(), o.size());
  if (!po.empty()) std::memcpy(pos_out, po.data(), po.size());
  *pos_len = po.size();
  return int64_t(o.size());
}

int irs_synth_build(const irs_synth_params* p, irs_synth_index** out) {
  if (!p || !out || p->num_docs == 0 || p->max_rank == 0 ||
      p->vocab_log2 == 0 || p->vocab_log2 > 24 ||
      p->max_rank > (1u << p->vocab_log2) || p->num_docs >= 0x7FFFFFF0u ||
      p->wand_count > 8 || p->wand_kind > IRS_SYNTH_WAND_DIV_NORM ||
      (p->one_based_positions && p->wand_count) || p->topic_percent > 100)
    return -1;
  auto idx = std::make_unique<irs_synth_index>();
  const uint32_t N = p->num_docs;
  const uint32_t R = p->max_rank;
  uint32_t T = p->threads ? p->threads : std::thread::hardware_concurrency();
  if (T == 0) T = 1;
  if (T > 256) T = 256;
  if (uint64_t(T) * 1024 > N) T = std::max<uint32_t>(1, N / 1024);

  const Zipf zipf(p->vocab_log2, R);
  idx->norms.resize(N);

  // ---- pass 1: per-thread doc ranges -> per-rank (doc, tf) runs ----------
  std::vector<ThreadPostings> tp(T);
  std::vector<uint64_t> ttf(T, 0);
  {
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < T; ++t) {
      pool.emplace_back([&, t] {
        auto& mine = tp[t];
        mine.docs.resize(R);
        mine.tfs.resize(R);
        if (p->with_positions) mine.poss.resize(R);
        const uint32_t lo = uint32_t(uint64_t(N) * t / T);
        const uint32_t hi = uint32_t(uint64_t(N) * (t + 1) / T);
        uint32_t ranks[256];  // (rank << 8) | 1-based token position
        uint64_t local_ttf = 0;
        for (uint32_t d = lo; d < hi; ++d) {
          const uint64_t g = p->first_doc + d;
          const uint32_t len =
            doc_length(p->seed, g, p->mean_len, p->stddev_len);
          idx->norms[d] = uint8_t(len);
          local_ttf += len;
          uint32_t m = 0;
          for (uint32_t i = 0; i < len; ++i) {
            const uint64_t h = token_hash(p->seed, g, i);
            uint32_t r;
            if (p->topic_docs && mix64(h ^ 0x7091C5ull) % 100u < p->topic_percent) {
              // one of the topic's own ranks (clustered corpus)
              const uint64_t topic = g / p->topic_docs;
              const uint64_t slot = mix64(h ^ 0x51A7ull) % std::max<uint32_t>(1, p->topic_terms);
              r = 1u + uint32_t(mix64(p->seed ^ mix64(topic * 0x9E3779B97F4A7C15ull + slot)) % R);
            } else {
              r = zipf.sample(h);
            }
            if (r) ranks[m++] = (r << 8) | (i + 1);
          }
          std::sort(ranks, ranks + m);
          for (uint32_t i = 0; i < m;) {
            uint32_t j = i + 1;
            const uint32_t r = ranks[i] >> 8;
            while (j < m && (ranks[j] >> 8) == r) ++j;
            mine.docs[r - 1].push_back(d + kDocMin);
            mine.tfs[r - 1].push_back(uint8_t(j - i));
            if (p->with_positions)
              for (uint32_t k = i; k < j; ++k) mine.poss[r - 1].push_back(uint8_t(ranks[k]));
            i = j;
          }
        }
        ttf[t] = local_ttf;
      });
    }
    for (auto& th : pool) th.join();
  }
  idx->docs_with_field = N;
  for (uint64_t v : ttf) idx->total_term_freq += v;

  // scorers the field is indexed with (0 = none: no wand bytes, as index-put writes)
  uint32_t wand_kinds[8];
  for (uint32_t w = 0; w < 8; ++w) wand_kinds[w] = p->wand_kind;
  const WandSpec wand{idx->norms.data(), p->wand_count, wand_kinds};

  // ---- pass 2: encode terms in parallel (rank order = LPT order) ---------
  std::vector<Bytes> term_bytes(R);
  std::vector<Bytes> term_pos(p->with_positions ? R : 0);
  idx->metas.resize(R);
  if (p->keep_postings) {
    idx->docs.resize(R);
    idx->freqs.resize(R);
    if (p->with_positions) idx->positions.resize(R);
  }
  {
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < T; ++t) {
      pool.emplace_back([&] {
        std::vector<uint32_t> d, f, ps;
        for (;;) {
          const uint32_t r = next.fetch_add(1);
          if (r >= R) break;
          size_t total = 0;
          for (uint32_t k = 0; k < T; ++k) total += tp[k].docs[r].size();
          d.resize(total);
          f.resize(total);
          size_t at = 0;
          for (uint32_t k = 0; k < T; ++k) {
            auto& dv = tp[k].docs[r];
            auto& fv = tp[k].tfs[r];
            std::copy(dv.begin(), dv.end(), d.begin() + at);
            for (size_t i = 0; i < fv.size(); ++i) f[at + i] = fv[i];
            at += dv.size();
            std::vector<uint32_t>().swap(dv);
            std::vector<uint8_t>().swap(fv);
          }
          if (p->with_positions) {
            ps.clear();
            for (uint32_t k = 0; k < T; ++k) {
              auto& pv = tp[k].poss[r];
              ps.insert(ps.end(), pv.begin(), pv.end());
              std::vector<uint8_t>().swap(pv);
            }
          }
          encode_term(d.data(), f.data(), uint32_t(total), N, p->layout,
                      term_bytes[r], idx->metas[r], wand.count ? &wand : nullptr,
                      p->with_positions ? ps.data() : nullptr,
                      p->with_positions ? &term_pos[r] : nullptr, p->one_based_positions != 0);
          if (p->keep_postings) {
            idx->docs[r] = d;
            idx->freqs[r] = f;
            if (p->with_positions) idx->positions[r] = ps;
          }
        }
      });
    }
    for (auto& th : pool) th.join();
  }

  // ---- assemble the `.doc` image -----------------------------------------
  Bytes& file = idx->doc_file;
  make_header(file, p->layout, kDocFormatName, p->one_based_positions != 0);
  uint64_t total = file.size();
  for (uint32_t r = 0; r < R; ++r) {
    idx->metas[r].doc_start += total;  // encode_term left it at 0
    total += term_bytes[r].size();
  }
  file.reserve(total + 16);
  for (uint32_t r = 0; r < R; ++r) {
    file.insert(file.end(), term_bytes[r].begin(), term_bytes[r].end());
    Bytes().swap(term_bytes[r]);
  }
  put_be32(file, uint32_t(-int32_t(0x3fd76c17)));
  put_be32(file, 0);
  put_be64(file, crc32c(file.data(), file.size()));

  if (p->with_positions) {  // the `.pos` image, same framing (:544-547)
    Bytes& pf = idx->pos_file;
    make_header(pf, p->layout, kPosFormatName, p->one_based_positions != 0);
    uint64_t at = pf.size();
    for (uint32_t r = 0; r < R; ++r) {
      idx->metas[r].pos_start += at;
      at += term_pos[r].size();
    }
    pf.reserve(at + 16);
    for (uint32_t r = 0; r < R; ++r) {
      pf.insert(pf.end(), term_pos[r].begin(), term_pos[r].end());
      Bytes().swap(term_pos[r]);
    }
    put_be32(pf, uint32_t(-int32_t(0x3fd76c17)));
    put_be32(pf, 0);
    put_be64(pf, crc32c(pf.data(), pf.size()));
  }

  *out = idx.release();
  return 0;
}

void irs_synth_free(irs_synth_index* idx) { delete idx; }

const uint8_t* irs_synth_doc_bytes(const irs_synth_index* idx, uint64_t* len) {
  if (len) *len = idx->doc_file.size();
  return idx->doc_file.data();
}
const uint8_t* irs_synth_pos_bytes(const irs_synth_index* idx, uint64_t* len) {
  if (len) *len = idx->pos_file.size();
  return idx->pos_file.empty() ? nullptr : idx->pos_file.data();
}
int irs_synth_positions(const irs_synth_index* idx, uint32_t rank,
                        const uint32_t** positions, uint64_t* count) {
  if (rank == 0 || rank > idx->positions.size()) return -1;
  *positions = idx->positions[rank - 1].data();
  *count = idx->positions[rank - 1].size();
  return 0;
}
const uint8_t* irs_synth_norms(const irs_synth_index* idx, uint64_t* count) {
  if (count) *count = idx->norms.size();
  return idx->norms.data();
}
const irs_synth_term_meta* irs_synth_term_metas(const irs_synth_index* idx,
                                                uint32_t* count) {
  if (count) *count = uint32_t(idx->metas.size());
  return idx->metas.data();
}
uint64_t irs_synth_docs_with_field(const irs_synth_index* idx) {
  return idx->docs_with_field;
}
uint64_t irs_synth_total_term_freq(const irs_synth_index* idx) {
  return idx->total_term_freq;
}
int irs_synth_postings(const irs_synth_index* idx, uint32_t rank,
                       const uint32_t** docs, const uint32_t** freqs,
                       uint32_t* count) {
  if (rank == 0 || rank > idx->docs.size()) return -1;
  *docs = idx->docs[rank - 1].data();
  *freqs = idx->freqs[rank - 1].data();
  *count = uint32_t(idx->docs[rank - 1].size());
  return 0;
}

int irs_synth_queries(uint64_t seed, uint32_t n_queries, uint32_t n_terms,
                      uint32_t lo_rank, uint32_t hi_rank, uint32_t* ranks_out) {
  if (!ranks_out || lo_rank == 0 || hi_rank < lo_rank ||
      hi_rank - lo_rank + 1 < n_terms)
    return -1;
  const double a = std::log(double(lo_rank));
  const double b = std::log(double(hi_rank) + 1.0);
  for (uint32_t q = 0; q < n_queries; ++q) {
    uint32_t* row = ranks_out + size_t(q) * n_terms;
    uint64_t ctr = 0;
    for (uint32_t t = 0; t < n_terms;) {
      const uint64_t h = mix64(seed ^ mix64((uint64_t(q) << 20) | ctr++));
      const double u = double(h >> 11) * (1.0 / 9007199254740992.0);
      uint32_t r = uint32_t(std::exp(a + u * (b - a)));
      r = std::min(std::max(r, lo_rank), hi_rank);
      bool dup = false;
      for (uint32_t k = 0; k < t; ++k) dup |= (row[k] == r);
      if (!dup) row[t++] = r;
    }
  }
  return 0;
}

}  // extern "C"
