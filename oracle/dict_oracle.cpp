// oracle/dict_oracle.cpp — TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// CPU restatement of the READER side of the two files a real index keeps next to `.doc`:
//   orc_encode_term_meta / orc_decode_term_meta
//       postings_writer_base::encode  core/formats/formats_10.cpp:576-604
//       postings_reader_base::decode  core/formats/formats_10.cpp:3421-3456
//   orc_walk_term_dictionary — the term iterator's traversal of `.tm`, from the ROOT block the
//       term index (FST) hands out: block_iterator::load / read_entry_* / next floor block
//       core/formats/formats_burst_trie.cpp:1765-1848, 1874-1930, 2020-2060, term_iterator::next
//       :2364-2420 (depth first, sub-blocks pushed when met)
//   orc_read_term_index — field_reader::prepare's `.ti` half: read_segment_features :711-733, per
//       field term_reader_base::prepare :1509-1545 + read_field_features :741-766, then the
//       field's FST (ImmutableFstImpl::Read, utils/fstext/immutable_fst.hpp:136-203) whose start
//       state's final weight is the root block's header (block_iterator :1751-1764)
//   orc_read_segment_meta — SegmentMetaReader::read  core/formats/formats_10.cpp:3147-3218
//   orc_read_fixed_column — columnstore2 reader::prepare_index + (dense_)fixed_length_column
//       core/formats/columnstore2.cpp:1746-1830, 650-789 (value at data + len*(doc - min)), 792-1011
// Written from the reader code; the product's own readers (iresearch_amd/cpp/irs_hip.hpp) walk
// `.tm` WITHOUT the root pointer (parse every block, resolve prefixes backwards) — the two must
// agree on the emitter's files (iresearch_amd/index/synth_dict.cpp, written from the writers).
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "oracle.h"

namespace {

inline uint32_t be32(const uint8_t* p) {
  return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3];
}
inline uint64_t be64(const uint8_t* p) { return (uint64_t(be32(p)) << 32) | be32(p + 4); }

struct Cursor {
  const uint8_t* p;
  const uint8_t* end;
  bool bad = false;
  uint64_t vlong() {
    uint64_t v = 0;
    for (unsigned sh = 0; sh < 70; sh += 7) {
      if (p >= end) { bad = true; return 0; }
      const uint8_t b = *p++;
      v |= uint64_t(b & 0x7Fu) << sh;
      if (!(b & 0x80u)) return v;
    }
    bad = true;
    return v;
  }
  uint32_t vint() { return uint32_t(vlong()); }
  const uint8_t* take(uint64_t n) {
    if (uint64_t(end - p) < n) { bad = true; return nullptr; }
    const uint8_t* at = p;
    p += n;
    return at;
  }
};

inline void put_v(std::vector<uint8_t>& o, uint64_t v) {
  while (v >= 0x80u) {
    o.push_back(uint8_t(v | 0x80u));
    v >>= 7;
  }
  o.push_back(uint8_t(v));
}

// format_utils::check_header (format_utils.cpp:74-105); returns the header length or 0
size_t header_len(const uint8_t* f, uint64_t len, const char* name, int32_t max_ver) {
  const size_t n = std::strlen(name);
  if (len < 9 + n || be32(f) != 0x3fd76c17u || f[4] != n || std::memcmp(f + 5, name, n)) return 0;
  const int32_t v = int32_t(be32(f + 5 + n));
  if (v < 0 || v > max_ver) return 0;
  return 9 + n;
}

constexpr uint64_t kInvalid = ~uint64_t(0);

struct Walk {
  const uint8_t* tm;
  uint64_t body_end;
  int has_freq, has_pos, has_pay;
  std::vector<std::string>* terms;
  std::vector<orc_term_meta>* metas;
  bool bad = false;

  // one group of floor blocks starting at `start`, its terms' common prefix `prefix`
  void group(uint64_t start, const std::string& prefix, int depth) {
    if (depth > 4096) { bad = true; return; }
    uint64_t at = start;
    for (;;) {   // block_iterator::load, then the next floor block right behind (cur_end_)
      if (at >= body_end) { bad = true; return; }
      Cursor c{tm + at, tm + body_end};
      const uint32_t head = c.vint();
      const bool last = head & 1u;
      const uint32_t n = head >> 1;
      const uint64_t sz = c.vlong();
      const bool leaf = sz & 1u;
      const uint8_t* sfx = c.take(sz >> 1);
      const uint64_t stats_len = c.vlong();
      const uint8_t* sts = c.take(stats_len);
      if (c.bad) { bad = true; return; }
      Cursor s{sfx, sfx + (sz >> 1)};
      orc_term_meta state;
      std::memset(&state, 0, sizeof state);   // begin_block: a block's records start from zeros
      state.pos_end = kInvalid;
      const uint8_t* sp = sts;
      for (uint32_t i = 0; i < n; ++i) {
        const uint32_t v = s.vint();
        const bool is_block = !leaf && (v & 1u);
        const uint32_t slen = leaf ? v : (v >> 1);
        const uint8_t* bytes = s.take(slen);
        if (s.bad) { bad = true; return; }
        std::string full = prefix;
        full.append(reinterpret_cast<const char*>(bytes), slen);
        if (is_block) {
          const uint64_t back = s.vlong();
          if (s.bad || back == 0 || back > at) { bad = true; return; }
          group(at - back, full, depth + 1);   // (sub-blocks hold longer terms: order within the
          if (bad) return;                     //  output is restored by the caller's sort)
        } else {
          const int64_t used = orc_decode_term_meta(sp, uint64_t(sts + stats_len - sp), has_freq,
                                                    has_pos, has_pay, &state);
          if (used < 0) { bad = true; return; }
          sp += used;
          terms->push_back(std::move(full));
          metas->push_back(state);
        }
      }
      if (sp != sts + stats_len || s.p != s.end) { bad = true; return; }
      if (last) return;
      at = uint64_t(c.p - tm);
    }
  }
};

}  // namespace

extern "C" {

int64_t orc_encode_term_meta(const orc_term_meta* meta, orc_term_meta* last, int has_pos,
                             int has_pay, uint8_t* out, uint64_t cap) {
  // formats_10.cpp:576-604: docs_count; freq - docs_count when the field has frequencies
  // (meta.freq != 0); doc_start delta; with positions: pos_start delta, pos_end when valid,
  // pay_start delta with offsets / payloads; then e_single_doc (one doc) or e_skip_start
  // (more than one block of docs)
  std::vector<uint8_t> o;
  put_v(o, meta->docs_count);
  if (meta->freq) put_v(o, meta->freq - meta->docs_count);
  put_v(o, meta->doc_start - last->doc_start);
  if (has_pos) {
    put_v(o, meta->pos_start - last->pos_start);
    if (meta->pos_end != kInvalid) put_v(o, meta->pos_end);
    if (has_pay) put_v(o, meta->pay_start - last->pay_start);
  }
  if (meta->docs_count == 1) put_v(o, uint32_t(meta->e_skip_start));
  else if (meta->docs_count > 128) put_v(o, meta->e_skip_start);
  *last = *meta;
  if (o.size() > cap) return -1;
  std::memcpy(out, o.data(), o.size());
  return int64_t(o.size());
}

int64_t orc_decode_term_meta(const uint8_t* in, uint64_t len, int has_freq, int has_pos,
                             int has_pay, orc_term_meta* state) {
  // formats_10.cpp:3421-3456
  Cursor c{in, in + len};
  state->docs_count = c.vint();
  if (has_freq) state->freq = state->docs_count + c.vint();
  state->doc_start += c.vlong();
  if (has_freq && state->freq && has_pos) {
    state->pos_start += c.vlong();
    state->pos_end = state->freq > 128 ? c.vlong() : kInvalid;
    if (has_pay) state->pay_start += c.vlong();
  }
  if (state->docs_count == 1) state->e_skip_start = c.vint();
  else if (state->docs_count > 128) state->e_skip_start = c.vlong();
  return c.bad ? -1 : int64_t(c.p - in);
}

/* Terms of the field whose root block group starts at `root_start`, ascending.  Two calls:
 * with terms == NULL it only counts (*n_terms, *term_bytes); then the caller passes buffers:
 * term_lens[n], terms (concatenated), metas[n].  Returns 0, <0 on a malformed file. */
int orc_walk_term_dictionary(const uint8_t* tm, uint64_t len, uint64_t root_start, int has_freq,
                             int has_pos, int has_pay, uint32_t* n_terms, uint64_t* term_bytes,
                             uint32_t* term_lens, uint8_t* terms, orc_term_meta* metas) {
  const size_t h1 = header_len(tm, len, "block_tree_terms_dict", 3);
  if (!h1 || len < h1 + 16) return -1;
  Cursor c{tm + h1, tm + len - 16};
  if (int32_t(be32(tm + h1 - 4)) > 0 && c.vint() != 0) return -2;   // encrypted
  const size_t h2 = header_len(c.p, uint64_t(c.end - c.p), "iresearch_10_postings_terms", 0);
  if (!h2) return -1;
  c.p += h2;
  if (c.vint() != 128 || c.bad) return -1;
  std::vector<std::string> ts;
  std::vector<orc_term_meta> ms;
  Walk w{tm, len - 16, has_freq, has_pos, has_pay, &ts, &ms};
  if (root_start < uint64_t(c.p - tm)) return -1;
  w.group(root_start, std::string(), 0);
  if (w.bad) return -3;
  // the iterator yields terms in order; the recursion above emits a sub-block's terms at the
  // place its entry stands, which IS that order (entries of a block ascend)
  uint64_t bytes = 0;
  for (const auto& t : ts) bytes += t.size();
  *n_terms = uint32_t(ts.size());
  *term_bytes = bytes;
  if (!terms) return 0;
  uint8_t* o = terms;
  for (size_t i = 0; i < ts.size(); ++i) {
    term_lens[i] = uint32_t(ts[i].size());
    std::memcpy(o, ts[i].data(), ts[i].size());
    o += ts[i].size();
    metas[i] = ms[i];
  }
  return 0;
}

/* The dense fixed-length column `column_id`: *value_bytes, *min_doc, *docs_count, the payload
 * (<= payload_cap bytes, *payload_len) and — values != NULL — its values, value of doc d at
 * value_bytes * (d - min_doc).  Returns 0, <0 on error / unsupported column. */
int orc_read_fixed_column(const uint8_t* csi, uint64_t csi_len, const uint8_t* csd,
                          uint64_t csd_len, uint32_t column_id, uint32_t* value_bytes,
                          uint32_t* min_doc, uint32_t* docs_count, uint8_t* payload,
                          uint32_t payload_cap, uint32_t* payload_len, uint8_t* values,
                          uint64_t values_cap) {
  const size_t hi = header_len(csi, csi_len, "iresearch_11_columnstore_index", 0);
  const size_t hd = header_len(csd, csd_len, "iresearch_11_columnstore_data", 0);
  if (!hi || !hd || csi_len < hi + 16 || csd_len < hd + 16) return -1;
  Cursor c{csi + hi, csi + csi_len - 16};
  const uint32_t count = c.vint();
  for (uint32_t i = 0; i < count && !c.bad; ++i) {
    const uint32_t comp_len = c.vint();
    const uint8_t* comp = c.take(comp_len);
    const uint8_t* h = c.take(24);
    if (c.bad) break;
    const uint64_t docs_index = be64(h);
    const uint32_t id = be32(h + 8), mn = be32(h + 12), docs = be32(h + 16);
    const uint32_t type = (uint32_t(h[20]) << 8) | h[21], props = (uint32_t(h[22]) << 8) | h[23];
    const uint32_t pl = c.vint();
    const uint8_t* pb = c.take(pl);
    if (!(props & 2u)) c.take(c.vint());           // the name of a named column
    if (docs_index) {                               // read_bitmap_index
      const uint8_t* nbp = c.take(4);
      if (c.bad) break;
      const uint32_t nb = be32(nbp);
      if (nb > 2) c.take(uint64_t(nb) * 8);
    }
    const uint32_t nblocks = (docs + 65535u) >> 16;
    if (type == 0) { c.take(uint64_t(nblocks) * 33); continue; }
    if (type == 1) continue;
    if (type != 2 && type != 3) return -1;
    const uint8_t* lp = c.take(8);
    const uint8_t* offs = c.take(uint64_t(type == 3 ? 1 : nblocks) * 8);
    if (c.bad) break;
    if (id != column_id) continue;
    const uint64_t vlen = be64(lp);
    if (docs_index || (props & 1u) || (vlen != 1 && vlen != 2 && vlen != 4)) return -2;
    if (comp_len != 28 || std::memcmp(comp, "iresearch::compression::none", 28)) return -2;
    *value_bytes = uint32_t(vlen);
    *min_doc = mn;
    *docs_count = docs;
    *payload_len = pl;
    if (pl > payload_cap) return -4;
    std::memcpy(payload, pb, pl);
    if (!values) return 0;
    if (values_cap < uint64_t(docs) * vlen) return -4;
    for (uint32_t b = 0; b < nblocks; ++b) {
      const uint64_t at = type == 3 ? be64(offs) + (uint64_t(b) << 16) * vlen : be64(offs + 8ull * b);
      uint64_t n = docs - (uint64_t(b) << 16);
      if (n > 65536) n = 65536;
      n *= vlen;
      if (at < hd || at + n > csd_len - 16) return -3;
      std::memcpy(values + (uint64_t(b) << 16) * vlen, csd + at, n);
    }
    return 0;
  }
  return c.bad ? -1 : -5;
}


// crc32c of everything in front of the footer's checksum (format_utils::check_footer)
static uint32_t orc_crc32c(const uint8_t* p, size_t n) {
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
static bool footer_ok(const uint8_t* f, uint64_t len) {
  if (len < 16) return false;
  const uint8_t* t = f + len - 16;
  return be32(t) == uint32_t(-int32_t(0x3fd76c17)) && be32(t + 4) == 0 && be64(t + 8) == orc_crc32c(f, len - 8);
}
static bool read_str(Cursor& c, std::string& out) {
  const uint32_t n = c.vint();
  const uint8_t* at = c.take(n);
  if (c.bad) return false;
  out.assign(reinterpret_cast<const char*>(at), n);
  return true;
}

int orc_read_term_index(const uint8_t* ti, uint64_t len, orc_field_record* out, uint32_t cap,
                        uint32_t* count, uint32_t* segment_index_features) {
  const size_t hl = header_len(ti, len, "block_tree_terms_index", 3);
  if (!hl || !footer_ok(ti, len) || len < hl + 24) return -1;
  if (int32_t(be32(ti + hl - 4)) < 2) return -3;        // (mutable FSTs of formats < 1_3: not restated)
  const uint64_t fields = be64(ti + len - 24);          // the count sits in front of the footer
  Cursor c{ti + hl, ti + len - 24};
  if (c.vint() != 0) return -3;                         // encrypted
  const uint8_t* f4 = c.take(4);
  if (c.bad) return -1;
  const uint32_t seg_features = be32(f4);
  if (seg_features > 15) return -1;
  if (segment_index_features) *segment_index_features = seg_features;
  std::vector<std::string> feature_names;
  for (uint64_t n = c.vlong(); n && !c.bad; --n) {
    std::string s;
    if (!read_str(c, s)) return -1;
    feature_names.push_back(std::move(s));
  }
  std::string prev;
  for (uint64_t i = 0; i < fields; ++i) {
    orc_field_record r;
    std::memset(&r, 0, sizeof r);
    r.norm_column = -1;
    std::string name;
    if (!read_str(c, name)) return -1;
    if (i && !(prev < name)) return -1;                 // "Invalid field order in segment"
    prev = name;
    r.name_len = uint32_t(name.size());
    std::memcpy(r.name, name.data(), std::min<size_t>(name.size(), sizeof r.name));
    const uint8_t* ff = c.take(4);
    if (c.bad) return -1;
    r.index_features = be32(ff);
    if (r.index_features > 15) return -1;
    for (uint64_t n = c.vlong(); n && !c.bad; --n) {
      const uint64_t id = c.vlong();
      const uint64_t column = c.vlong() - 1;
      if (c.bad || id >= feature_names.size()) return -1;
      if (feature_names[id] == "iresearch::norm2") r.norm_column = int64_t(column);
    }
    r.terms_count = c.vlong();
    r.docs_count = c.vlong();
    r.total_doc_freq = c.vlong();
    std::string skip;
    if (!read_str(c, skip) || !read_str(c, skip)) return -1;   // min / max term
    if (r.index_features & 1u) r.total_term_freq = c.vlong();
    const uint8_t* wm = c.take(8);                      // (version >= WAND: checked by header_len's max)
    if (c.bad) return -1;
    r.wand_mask = be64(wm);
    // ImmutableFstImpl::Read
    const uint8_t* fh = c.take(1 + 8 + 8 + 4);
    if (c.bad || fh[0] != 0) return -1;
    const uint64_t total_weight = be64(fh + 9);
    const uint32_t nstates = be32(fh + 17);
    const uint32_t start = nstates - c.vint();
    (void)c.vlong();                                    // zig-zag(arcs - states)
    if (c.bad || start >= nstates) return -1;
    uint64_t at = 0, root_at = 0, root_len = 0;
    for (uint32_t s = 0; s < nstates && !c.bad; ++s) {
      const uint64_t packed = c.vlong();
      if (s == start) {
        root_at = at;
        root_len = packed >> 1;
      }
      at += packed >> 1;
      if (!(packed & 1u)) {
        const uint8_t* nb = c.take(1);
        if (c.bad) return -1;
        for (uint32_t a = uint32_t(*nb) + 1; a && !c.bad; --a) {
          (void)c.take(1);
          (void)c.vint();
          at += c.vlong();
        }
      }
    }
    const uint8_t* weights = c.take(total_weight);
    if (c.bad || at != total_weight || root_at + root_len > total_weight || root_len < 2) return -1;
    Cursor w{weights + root_at, weights + root_at + root_len};
    r.root_meta = *w.take(1);                           // block_iterator: meta, start, floor data
    r.root_start = w.vlong();
    if (r.root_meta & 4u) r.root_floor_blocks = w.vint();
    if (w.bad) return -1;
    if (i < cap && out) out[i] = r;
  }
  if (c.bad || c.p != c.end) return -1;
  if (count) *count = uint32_t(fields);
  return 0;
}

int orc_read_segment_meta(const uint8_t* sm, uint64_t len, uint64_t* docs_count,
                          uint64_t* live_docs_count, uint32_t* has_column_store, uint32_t* n_files) {
  const size_t hl = header_len(sm, len, "iresearch_10_segment_meta", 1);
  if (!hl || !footer_ok(sm, len)) return -1;
  const int32_t version = int32_t(be32(sm + hl - 4));
  Cursor c{sm + hl, sm + len - 16};
  std::string name;
  if (!read_str(c, name)) return -1;
  (void)c.vlong();                                      // segment version
  const uint64_t live = c.vlong();
  const uint64_t docs = c.vlong() + live;
  (void)c.vlong();                                      // byte size
  const uint8_t* fl = c.take(1);
  if (c.bad || (*fl & ~3u)) return -1;
  uint64_t sort = 0;
  if (version > 0) sort = c.vlong();
  if (((*fl & 2u) != 0) != (sort != 0)) return -1;
  const uint64_t files = c.vlong();
  for (uint64_t i = 0; i < files; ++i) {
    std::string s;
    if (!read_str(c, s)) return -1;
  }
  if (c.bad || c.p != c.end) return -1;
  if (docs_count) *docs_count = docs;
  if (live_docs_count) *live_docs_count = live;
  if (has_column_store) *has_column_store = *fl & 1u;
  if (n_files) *n_files = uint32_t(files);
  return 0;
}

// DocumentMaskReader::read (core/formats/formats_10.cpp:3275-3312): checksum, header
// ("iresearch_10_doc_mask", versions 0..0), vint count, `count` vint doc ids, footer.
int64_t orc_read_document_mask(const uint8_t* dm, uint64_t len, uint32_t* docs, uint64_t cap) {
  const size_t hl = header_len(dm, len, "iresearch_10_doc_mask", 0);
  if (!hl || !footer_ok(dm, len)) return -1;
  Cursor c{dm + hl, dm + len - 16};
  const uint64_t count = c.vint();
  for (uint64_t i = 0; i < count; ++i) {
    const uint32_t d = c.vint();
    if (c.bad) return -1;
    if (docs && i < cap) docs[i] = d;
  }
  if (c.bad || c.p != c.end) return -1;
  return int64_t(count);
}

}  // extern "C"
