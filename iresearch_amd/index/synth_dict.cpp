// synth_dict.cpp — the two other files a segment's hot path is fed from, written the way the
// reference's WRITERS lay them out (an emitter for tests and the ingestion path; host only,
// no GPU, no oracle):
//
//   `.tm`   term dictionary of one field: blocks of (suffix, term_meta) entries
//           field_writer::Push / WriteBlocks / WriteBlock  core/formats/formats_burst_trie.cpp:1023-1196
//           postings_writer_base::encode (the per-term stats) core/formats/formats_10.cpp:576-604
//           file framing: field_writer::prepare :1235-1290 + postings_writer_base::prepare
//           formats_10.cpp:563-565, footer field_writer::end :1434-1455
//   `.csd` / `.csi`   columnstore2 data + index holding the dense Norm2 column
//           writer::prepare / commit  core/formats/columnstore2.cpp:1561-1697
//           column::flush_block / finish  :1340-1551,  write_header :69-77
//
//   `.ti`   term index: the segment's feature list, then per field (in name order) its record —
//           name, index features, feature -> column ids, counts, min / max term, wand mask —
//           and the FST over its block prefixes
//           field_writer::prepare :1263-1287, write_segment_features :662-681, EndField
//           :1347-1432, write_field_features :684-708, MergeBlocks :856-911 (what the FST
//           holds for a block), ImmutableFst::Write utils/fstext/immutable_fst.hpp:269-335
//           Here the FST of a field has ONE state whose final weight is the root block's
//           record (the output of the empty prefix — which is where every reader starts);
//           the prefix arcs that only accelerate seeks are not built.
//   `.sm`   segment meta: SegmentMetaWriter::write  core/formats/formats_10.cpp:3102-3140
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "synth_index.h"

namespace {

using Bytes = std::vector<uint8_t>;

inline void put_vint(Bytes& o, uint32_t v) {
  while (v >= 0x80u) {
    o.push_back(uint8_t(v | 0x80u));
    v >>= 7;
  }
  o.push_back(uint8_t(v));
}
inline void put_vlong(Bytes& o, uint64_t v) {
  while (v >= 0x80u) {
    o.push_back(uint8_t(v | 0x80u));
    v >>= 7;
  }
  o.push_back(uint8_t(v));
}
inline void put_be16(Bytes& o, uint16_t v) {
  o.push_back(uint8_t(v >> 8));
  o.push_back(uint8_t(v));
}
inline void put_be32(Bytes& o, uint32_t v) {
  for (int s = 24; s >= 0; s -= 8) o.push_back(uint8_t(v >> s));
}
inline void put_be64(Bytes& o, uint64_t v) {
  put_be32(o, uint32_t(v >> 32));
  put_be32(o, uint32_t(v));
}
inline void put_string(Bytes& o, const void* p, size_t n) {   // write_string: vint size + bytes
  put_vint(o, uint32_t(n));
  const uint8_t* b = static_cast<const uint8_t*>(p);
  o.insert(o.end(), b, b + n);
}

uint32_t crc32c(const uint8_t* p, size_t n) {   // Castagnoli, reflected (utils/crc.hpp)
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      table[i] = c;
    }
    ready = true;
  }
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

// format_utils::write_header / write_footer (format_utils.cpp:57-67)
void put_header(Bytes& o, const char* format, int32_t version) {
  put_be32(o, 0x3fd76c17u);
  put_string(o, format, std::strlen(format));
  put_be32(o, uint32_t(version));
}
void put_footer(Bytes& o) {
  put_be32(o, uint32_t(-int32_t(0x3fd76c17)));
  put_be32(o, 0);
  put_be64(o, crc32c(o.data(), o.size()));
}

constexpr uint64_t kNoAddress = ~uint64_t(0);   // address_limits::invalid()
constexpr uint32_t kPostingsBlock = 128;        // skip_.Skip0()

struct Features {
  bool freq, pos, pay;
};

// postings_writer_base::encode: one term's stats, delta-coded against the previous term of
// the block (`last`; begin_block() zeroes it).
void encode_meta(Bytes& o, const irs_synth_term_meta& m, irs_synth_term_meta& last,
                 const Features& f) {
  put_vint(o, m.docs_count);
  if (m.freq) put_vint(o, m.freq - m.docs_count);   // (a field without FREQ keeps freq == 0)
  put_vlong(o, m.doc_start - last.doc_start);
  if (f.pos) {
    put_vlong(o, m.pos_start - last.pos_start);
    if (m.pos_end != kNoAddress) put_vlong(o, m.pos_end);
    if (f.pay) put_vlong(o, m.pay_start - last.pay_start);
  }
  if (m.docs_count == 1) {
    put_vint(o, uint32_t(m.e_skip_start));            // e_single_doc
  } else if (m.docs_count > kPostingsBlock) {
    put_vlong(o, m.e_skip_start);
  }
  last = m;
}

// ---- the block tree -----------------------------------------------------------------------
// An entry on the writer's stack: a term with its stats, or a finished block of longer terms
// (the group's FIRST floor block stands for all of them).
struct Entry {
  std::string data;          // term, or the prefix the block's terms share
  bool is_block = false;
  irs_synth_term_meta meta{};
  uint64_t block_start = 0;  // file offset of the (first floor) block
};

struct DictWriter {
  Bytes& out;
  Features feat;
  uint32_t min_block, max_block;
  std::vector<Entry> stack;
  std::vector<size_t> prefixes;   // prefixes[i]: first stack slot sharing last_term[0..i]
  std::string last_term;
  uint64_t root_start = 0;
  // the last group written: per (floor) block its start, lead label (-1: none) and meta bits
  // (1 has terms, 2 has sub-blocks, 4 is one of several floor blocks) — MergeBlocks' input
  struct Floor { uint64_t start; int label; uint8_t meta; };
  std::vector<Floor> group;

  // One block of entries stack[begin, end) whose terms share `prefix` bytes.
  uint64_t write_block(size_t prefix, size_t begin, size_t end, bool leaf, bool last_of_group) {
    const uint64_t start = out.size();
    put_vint(out, (uint32_t(end - begin) << 1) | (last_of_group ? 1u : 0u));
    Bytes suffix, stats;
    irs_synth_term_meta last{};   // begin_block()
    for (size_t i = begin; i < end; ++i) {
      const Entry& e = stack[i];
      const uint32_t suf = uint32_t(e.data.size() - prefix);
      put_vint(suffix, leaf ? suf : ((suf << 1) | (e.is_block ? 1u : 0u)));
      suffix.insert(suffix.end(), e.data.begin() + prefix, e.data.end());
      if (!e.is_block) {
        encode_meta(stats, e.meta, last, feat);
      } else {
        put_vlong(suffix, start - e.block_start);
      }
    }
    put_vlong(out, (uint64_t(suffix.size()) << 1) | (leaf ? 1u : 0u));
    out.insert(out.end(), suffix.begin(), suffix.end());
    put_vlong(out, stats.size());
    out.insert(out.end(), stats.begin(), stats.end());
    return start;
  }

  // The top `count` entries of the stack share `prefix` bytes: they become one block, or —
  // more than max_block of them — several FLOOR blocks cut where the byte behind the prefix
  // changes; one block entry replaces them on the stack.
  void write_blocks(size_t prefix, size_t count) {
    const size_t end = stack.size(), begin = end - count;
    size_t block_start = begin;
    uint64_t first = 0;
    bool have_first = false;
    int last_label = -2;
    bool has_blocks = false, has_terms = false;
    int lead_label = -1;   // of the block being collected (next_label)
    group.clear();
    auto flush = [&](size_t from, size_t to) {
      const uint64_t at = write_block(prefix, from, to, !has_blocks, to == end);
      if (!have_first) {
        first = at;
        have_first = true;
      }
      group.push_back(Floor{at, lead_label,
                            uint8_t((has_terms ? 1u : 0u) | (has_blocks ? 2u : 0u) | (to - from < count ? 4u : 0u))});
      has_blocks = has_terms = false;
    };
    for (size_t i = begin; i < end; ++i) {
      const Entry& e = stack[i];
      const int label = e.data.size() == prefix ? -1 : int(uint8_t(e.data[prefix]));
      if (label != last_label) {
        const size_t size = i - block_start;
        if (size >= min_block && end - block_start > max_block) {
          flush(block_start, i);
          lead_label = label;
          block_start = i;
        }
        last_label = label;
      }
      has_blocks = has_blocks || e.is_block;
      has_terms = has_terms || !e.is_block;
    }
    if (block_start < end) flush(block_start, end);
    Entry blk;
    blk.data = last_term.substr(0, prefix);
    blk.is_block = true;
    blk.block_start = first;
    stack.erase(stack.begin() + begin, stack.end());
    stack.push_back(std::move(blk));
  }

  // A new term arrives (ascending order): every prefix group of the previous term that the
  // new one leaves is closed; a group of more than min_block entries becomes a block.
  void push(const std::string& term) {
    size_t pos = 0;
    const size_t limit = std::min(last_term.size(), term.size());
    while (pos < limit && term[pos] == last_term[pos]) ++pos;
    for (size_t i = last_term.empty() ? 0 : last_term.size() - 1; i > pos;) {
      --i;
      const size_t top = stack.size() - prefixes[i];
      if (top > min_block) {
        write_blocks(i + 1, top);
        prefixes[i] -= (top - 1);
      }
    }
    prefixes.resize(term.size());
    std::fill(prefixes.begin() + pos, prefixes.end(), stack.size());
    last_term = term;
  }

  void add(const std::string& term, const irs_synth_term_meta& meta) {
    push(term);
    Entry e;
    e.data = term;
    e.meta = meta;
    stack.push_back(std::move(e));
  }

  void finish() {
    push(std::string());
    write_blocks(0, stack.size());
    root_start = stack.front().block_start;
  }
};

}  // namespace

extern "C" {

int64_t irs_synth_term_meta_stream(const irs_synth_term_meta* metas, uint32_t n, uint32_t has_freq,
                                   uint32_t has_pos, uint32_t has_pay, uint8_t* out,
                                   uint64_t out_cap) {
  if (!metas && n) return -1;
  Bytes o;
  irs_synth_term_meta last{};
  const Features f{has_freq != 0, has_pos != 0, has_pay != 0};
  for (uint32_t i = 0; i < n; ++i) encode_meta(o, metas[i], last, f);
  if (o.size() > out_cap) return -2;
  if (!o.empty()) std::memcpy(out, o.data(), o.size());
  return int64_t(o.size());
}

int64_t irs_synth_term_dictionary(const uint8_t* terms, const uint32_t* term_lens,
                                  const irs_synth_term_meta* metas, uint32_t n,
                                  uint32_t has_freq, uint32_t has_pos, uint32_t has_pay,
                                  uint32_t min_block, uint32_t max_block, uint8_t* out,
                                  uint64_t out_cap, uint64_t* root_start) {
  if ((!terms || !term_lens || !metas) && n) return -1;
  if (min_block < 2 || max_block < min_block || 2 * (min_block - 1) > max_block) return -1;
  Bytes o;
  put_header(o, "block_tree_terms_dict", 3);          // burst_trie::Version::WAND
  put_vint(o, 0);                                       // irs::encrypt: no cipher = empty header
  put_header(o, "iresearch_10_postings_terms", 0);     // postings_writer_base::prepare
  put_vint(o, kPostingsBlock);
  DictWriter w{o, Features{has_freq != 0, has_pos != 0, has_pay != 0}, min_block, max_block,
               {}, {}, {}, 0};
  const uint8_t* p = terms;
  std::string prev;
  for (uint32_t i = 0; i < n; ++i) {
    std::string t(reinterpret_cast<const char*>(p), term_lens[i]);
    p += term_lens[i];
    if (i && !(prev < t)) return -3;   // terms must ascend (bytewise)
    if (metas[i].docs_count) w.add(t, metas[i]);   // (field_writer::write drops empty terms)
    prev = std::move(t);
  }
  if (!w.stack.empty()) w.finish();
  put_footer(o);
  if (o.size() > out_cap) return -2;
  std::memcpy(out, o.data(), o.size());
  if (root_start) *root_start = w.root_start;
  return int64_t(o.size());
}

// One anonymous fixed-length column (the Norm2 feature column of a field) behind `lead`
// other columns (mask columns: no data) — enough structure for a reader to have to FIND it.
int64_t irs_synth_columnstore(const uint8_t* values, uint32_t value_bytes, uint32_t n_docs,
                              uint32_t min_doc, const uint8_t* payload, uint32_t payload_len,
                              uint32_t dense_fixed, uint32_t lead_columns, uint8_t* csd_out,
                              uint64_t csd_cap, uint64_t* csd_len, uint8_t* csi_out,
                              uint64_t csi_cap, uint64_t* csi_len, uint32_t* column_id) {
  if (!values || !n_docs || (value_bytes != 1 && value_bytes != 2 && value_bytes != 4)) return -1;
  constexpr uint32_t kBlock = 1u << 16;   // column::kBlockSize
  Bytes data, index;
  put_header(data, "iresearch_11_columnstore_data", 0);
  put_vint(data, 0);                                    // irs::encrypt: empty header
  // the column's blocks: fixed-length values back to back (no address table: all equal)
  std::vector<uint64_t> block_at;
  for (uint64_t d = 0; d < n_docs; d += kBlock) {
    // a non-consolidated writer interleaves blocks of different columns; a few filler bytes
    // between ours stand for that (a dense-fixed column is written in one piece)
    if (!dense_fixed && d) data.insert(data.end(), 7, uint8_t(0xEE));
    block_at.push_back(data.size());
    const uint64_t cnt = std::min<uint64_t>(kBlock, n_docs - d);
    data.insert(data.end(), values + d * value_bytes, values + (d + cnt) * value_bytes);
  }
  put_footer(data);

  put_header(index, "iresearch_11_columnstore_index", 0);
  const uint32_t count = lead_columns + 1;
  put_vint(index, count);
  const char* comp = "iresearch::compression::none";
  auto header = [&](uint32_t id, uint32_t docs, uint16_t type, uint16_t props) {
    put_string(index, comp, std::strlen(comp));
    put_be64(index, 0);        // docs_index: 0 = every doc has a value
    put_be32(index, id);
    put_be32(index, min_doc);
    put_be32(index, docs);
    put_be16(index, type);
    put_be16(index, props);
  };
  // columns are stored sorted by name, the anonymous ones (null name) first
  const uint32_t id = 0;       // the norm column was pushed first
  header(id, n_docs, dense_fixed ? 3 /*kDenseFixed*/ : 2 /*kFixed*/, 2 /*kNoName*/);
  put_string(index, payload, payload_len);
  put_be64(index, value_bytes);                         // avg = the value length
  if (dense_fixed) {
    put_be64(index, block_at.front());
  } else {
    for (uint64_t at : block_at) put_be64(index, at);
  }
  for (uint32_t c = 0; c < lead_columns; ++c) {
    const std::string name = "mask" + std::to_string(c);
    header(1 + c, n_docs, 1 /*kMask*/, 0);
    put_string(index, "", 0);
    put_string(index, name.data(), name.size());
  }
  put_footer(index);
  if (data.size() > csd_cap || index.size() > csi_cap) return -2;
  std::memcpy(csd_out, data.data(), data.size());
  std::memcpy(csi_out, index.data(), index.size());
  *csd_len = data.size();
  *csi_len = index.size();
  if (column_id) *column_id = id;
  return 0;
}

// ---- a segment's `.tm` + `.ti` (every field) and its `.sm` ------------------------------------

int64_t irs_synth_segment_dictionary(const irs_synth_field* fields, uint32_t n_fields,
                                     uint32_t min_block, uint32_t max_block, uint8_t* tm_out,
                                     uint64_t tm_cap, uint64_t* tm_len, uint8_t* ti_out,
                                     uint64_t ti_cap, uint64_t* ti_len) {
  if (!fields || !n_fields || !tm_len || !ti_len) return -1;
  if (min_block < 2 || max_block < min_block || 2 * (min_block - 1) > max_block) return -1;
  Bytes tm, ti;
  // field_writer::prepare: both files, then the segment's features into the term index
  put_header(tm, "block_tree_terms_dict", 3);          // burst_trie::Version::WAND
  put_vint(tm, 0);                                      // irs::encrypt: no cipher = empty header
  put_header(tm, "iresearch_10_postings_terms", 0);    // postings_writer_base::prepare
  put_vint(tm, kPostingsBlock);
  put_header(ti, "block_tree_terms_index", 3);
  put_vint(ti, 0);
  uint32_t seg_features = 0;
  bool any_norm = false;
  for (uint32_t f = 0; f < n_fields; ++f) {
    seg_features |= fields[f].index_features;
    any_norm = any_norm || fields[f].norm_column >= 0;
    if (f && !(std::string(fields[f - 1].name, fields[f - 1].name_len) <
               std::string(fields[f].name, fields[f].name_len)))
      return -3;   // fields are written in name order (field_reader::prepare checks it)
  }
  put_be32(ti, seg_features);                           // write_segment_features
  const char* kNorm2 = "iresearch::norm2";              // irs::Norm2::type_name()
  put_vlong(ti, any_norm ? 1 : 0);
  if (any_norm) put_string(ti, kNorm2, std::strlen(kNorm2));
  uint64_t n_written = 0;
  for (uint32_t f = 0; f < n_fields; ++f) {
    const irs_synth_field& fd = fields[f];
    if ((!fd.terms || !fd.term_lens || !fd.metas) && fd.n_terms) return -1;
    const bool has_freq = (fd.index_features & 1u) != 0, has_pos = (fd.index_features & 2u) != 0;
    const bool has_pay = (fd.index_features & 12u) != 0;   // OFFS | PAY
    DictWriter w{tm, Features{has_freq, has_pos, has_pay}, min_block, max_block, {}, {}, {}, 0, {}};
    const uint8_t* p = fd.terms;
    std::string prev, min_term, max_term;
    uint64_t term_count = 0, sum_df = 0, sum_tf = 0;
    for (uint32_t i = 0; i < fd.n_terms; ++i) {
      std::string t(reinterpret_cast<const char*>(p), fd.term_lens[i]);
      p += fd.term_lens[i];
      if (i && !(prev < t)) return -3;
      if (fd.metas[i].docs_count) {   // (field_writer::write drops empty terms)
        w.add(t, fd.metas[i]);
        if (!term_count) min_term = t;
        max_term = t;
        ++term_count;
        sum_df += fd.metas[i].docs_count;
        sum_tf += fd.metas[i].freq;
      }
      prev = std::move(t);
    }
    if (!term_count) continue;   // EndField: nothing to write
    w.finish();
    // EndField: the field's record ...
    put_string(ti, fd.name, fd.name_len);
    put_be32(ti, fd.index_features);                    // write_field_features
    put_vlong(ti, fd.norm_column >= 0 ? 1 : 0);
    if (fd.norm_column >= 0) {
      put_vlong(ti, 0);                                 // the feature's id in the segment's list
      put_vlong(ti, uint64_t(fd.norm_column) + 1);
    }
    put_vlong(ti, term_count);
    put_vlong(ti, fd.docs_with_field);
    put_vlong(ti, sum_df);
    put_string(ti, min_term.data(), min_term.size());
    put_string(ti, max_term.data(), max_term.size());
    if (has_freq) put_vlong(ti, sum_tf);
    put_be64(ti, fd.wand_mask);
    // ... and its FST: one state (the start), no arcs, final weight = the root block's record
    Bytes root;                                         // MergeBlocks
    const auto& g = w.group;
    root.push_back(g.front().meta);
    put_vlong(root, g.front().start);
    if (g.front().meta & 4u) {
      put_vint(root, uint32_t(g.size() - 1));
      for (size_t i = 1; i < g.size(); ++i) {
        root.push_back(uint8_t(g[i].label & 0xFF));
        put_vlong(root, g[i].start - g.front().start);
        root.push_back(g[i].meta);
      }
    }
    ti.push_back(0);                                    // ImmutableFst::Write: Version::MIN
    put_be64(ti, 1);                                    // properties (kExpanded)
    put_be64(ti, root.size());                          // total weight size
    put_be32(ti, 1);                                    // states
    put_vint(ti, 1);                                    // states - start
    put_vlong(ti, 1);                                   // zig-zag(arcs - states = -1)
    put_vlong(ti, (uint64_t(root.size()) << 1) | 1u);   // final weight size, no arcs
    ti.insert(ti.end(), root.begin(), root.end());
    ++n_written;
  }
  put_footer(tm);
  put_be64(ti, n_written);                              // field_writer::end
  put_footer(ti);
  if (tm.size() > tm_cap || ti.size() > ti_cap) return -2;
  std::memcpy(tm_out, tm.data(), tm.size());
  std::memcpy(ti_out, ti.data(), ti.size());
  *tm_len = tm.size();
  *ti_len = ti.size();
  return int64_t(n_written);
}

int64_t irs_synth_segment_meta(const char* name, uint32_t name_len, uint64_t version,
                               uint64_t docs_count, uint64_t live_docs_count, uint64_t byte_size,
                               uint32_t has_column_store, const char* const* files,
                               const uint32_t* file_lens, uint32_t n_files, uint8_t* out,
                               uint64_t out_cap) {
  if (!name || docs_count < live_docs_count || (n_files && (!files || !file_lens))) return -1;
  Bytes o;
  put_header(o, "iresearch_10_segment_meta", 1);
  put_string(o, name, name_len);
  put_vlong(o, version);
  put_vlong(o, live_docs_count);
  put_vlong(o, docs_count - live_docs_count);
  put_vlong(o, byte_size);
  o.push_back(has_column_store ? 1 : 0);               // flags: HAS_COLUMN_STORE, not SORTED
  put_vlong(o, 0);                                      // 1 + sort, sort = field_limits::invalid()
  put_vlong(o, n_files);                                // write_strings
  for (uint32_t i = 0; i < n_files; ++i) put_string(o, files[i], file_lens[i]);
  put_footer(o);
  if (o.size() > out_cap) return -2;
  std::memcpy(out, o.data(), o.size());
  return int64_t(o.size());
}

// `.doc_mask` — DocumentMaskWriter::write (formats_10.cpp:3245-3268): header
// ("iresearch_10_doc_mask", version 0), vint count, one vint per deleted doc id in the mask's
// iteration order (a hash set's: any order), footer.
int64_t irs_synth_document_mask(const uint32_t* docs, uint64_t n, uint8_t* out, uint64_t out_cap) {
  if ((n && !docs) || n > 0xFFFFFFFFull) return -1;
  Bytes o;
  put_header(o, "iresearch_10_doc_mask", 0);
  put_vint(o, uint32_t(n));
  for (uint64_t i = 0; i < n; ++i) put_vint(o, docs[i]);
  put_footer(o);
  if (o.size() > out_cap) return -2;
  std::memcpy(out, o.data(), o.size());
  return int64_t(o.size());
}

}  // extern "C"
