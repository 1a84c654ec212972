/* oracle/postings_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Plain-C restatement of the reference's posting decode path:
 *   block codec   core/utils/bit_packing.cpp:48-843 (fastpack/fastunpack<N>),
 *                 external/simdcomp/src/simdbitpacking.c (simdpackwithoutmask /
 *                 simdunpack), core/formats/formats_10.cpp:96-116, 4131-4142
 *   framing       core/utils/bitpack.hpp:60-69, 149-177
 *   iterator      core/formats/formats_10.cpp:1741-1792 (refill/read_tail_block),
 *                 2089-2119 (next), 2238-2302 (prepare), 1803-1919 (single doc)
 *   skip data     core/formats/skip_list.cpp:111-156, formats_10.cpp:1063-1080
 */
#include <string.h>

#include "oracle.h"
#include "oracle_internal.h"



/* ------------------------------------------------------------------ codec */

/* fastpack<N> semantics: value i of a 32-value sub-block, masked to N bits, is
 * OR-ed at bit (N*i)%32 of word (N*i)/32 and spills into the next word
 * (bit_packing.cpp:48-60 pattern, repeated 32 times); four sub-blocks are laid
 * back to back `bits` words apart (formats_10.cpp:96-105). */
void orc_pack_scalar(const uint32_t* in, uint32_t bits, uint32_t* out) {
  uint32_t sb, i;
  memset(out, 0, 16u * bits);
  for (sb = 0; sb < 4; ++sb) {
    uint32_t* o = out + sb * bits;
    const uint32_t* v = in + sb * 32;
    if (bits == 32) { /* bit_packing.cpp: case 32 -> memcpy */
      memcpy(o, v, 128);
      continue;
    }
    for (i = 0; i < 32; ++i) {
      const uint32_t x = v[i] % (1u << bits);
      const uint32_t bit = bits * i;
      const uint32_t w = bit >> 5, s = bit & 31;
      o[w] |= x << s;
      if (s + bits > 32) o[w + 1] |= x >> (32 - s);
    }
  }
}

/* fastunpack<N> (bit_packing.cpp:644-843) */
void orc_unpack_scalar(const uint32_t* in, uint32_t bits, uint32_t* out) {
  uint32_t sb, i;
  for (sb = 0; sb < 4; ++sb) {
    const uint32_t* p = in + sb * bits;
    uint32_t* v = out + sb * 32;
    if (bits == 32) {
      memcpy(v, p, 128);
      continue;
    }
    for (i = 0; i < 32; ++i) {
      const uint32_t bit = bits * i;
      const uint32_t w = bit >> 5, s = bit & 31;
      uint32_t x = p[w] >> s;
      if (s + bits > 32) x |= p[w + 1] << (32 - s);
      v[i] = x & ((1u << bits) - 1u);
    }
  }
}

uint32_t orc_at_scalar(const uint32_t* in, uint32_t i, uint32_t bits) {
  /* packed::at -> fastpack_at(encoded + bit*(i/32), i%32, bit) */
  const uint32_t* p = in + bits * (i / 32);
  const uint32_t bit = bits * (i % 32);
  const uint32_t w = bit >> 5, s = bit & 31;
  uint32_t x;
  if (bits == 32) return p[i % 32];
  x = p[w] >> s;
  if (s + bits > 32) x |= p[w + 1] << (32 - s);
  return x & ((1u << bits) - 1u);
}

/* simdpackwithoutmask: row r (= one __m128i of 4 consecutive inputs) is
 * shifted left by (r*bits)%32 and OR-ed into output vector (r*bits)/32, the
 * spill going to the next vector — per 32-bit lane independently
 * (e.g. __SIMD_fastpackwithoutmask5_32, simdbitpacking.c:338-). No masking. */
void orc_pack_simd4(const uint32_t* in, uint32_t bits, uint32_t* out) {
  uint32_t r, l;
  memset(out, 0, 16u * bits);
  if (bits == 32) {
    memcpy(out, in, 512);
    return;
  }
  for (r = 0; r < 32; ++r) {
    const uint32_t bit = r * bits;
    const uint32_t k = bit >> 5, s = bit & 31;
    for (l = 0; l < 4; ++l) {
      const uint32_t x = in[4 * r + l];
      out[4 * k + l] |= x << s;
      if (s + bits > 32) out[4 * (k + 1) + l] |= x >> (32 - s);
    }
  }
}

/* simdunpack (e.g. __SIMD_fastunpack5_32, simdbitpacking.c:9513-).  The reference's
 * decoder for this layout is SSE code; so is this one wherever SSE2 exists (every
 * x86-64), so that the CPU baseline timed by bench.py is not handicapped by a
 * scalar unpack.  Row r of the output = four values, one per 32-bit lane. */
#if defined(__SSE2__)
#include <emmintrin.h>
void orc_unpack_simd4(const uint32_t* in, uint32_t bits, uint32_t* out) {
  uint32_t r;
  if (bits == 0) { /* simdunpack case 0: zero fill */
    memset(out, 0, 512);
    return;
  }
  if (bits == 32) {
    memcpy(out, in, 512);
    return;
  }
  {
    const __m128i mask = _mm_set1_epi32((int)((1u << bits) - 1u));
    const __m128i* src = (const __m128i*)in;
    __m128i* dst = (__m128i*)out;
    for (r = 0; r < 32; ++r) {
      const uint32_t bit = r * bits;
      const uint32_t k = bit >> 5, s = bit & 31;
      __m128i x = _mm_srl_epi32(_mm_loadu_si128(src + k), _mm_cvtsi32_si128((int)s));
      if (s + bits > 32)
        x = _mm_or_si128(x, _mm_sll_epi32(_mm_loadu_si128(src + k + 1),
                                          _mm_cvtsi32_si128((int)(32 - s))));
      _mm_storeu_si128(dst + r, _mm_and_si128(x, mask));
    }
  }
}
#else
void orc_unpack_simd4(const uint32_t* in, uint32_t bits, uint32_t* out) {
  uint32_t r, l;
  if (bits == 0) { /* simdunpack case 0: zero fill */
    memset(out, 0, 512);
    return;
  }
  if (bits == 32) {
    memcpy(out, in, 512);
    return;
  }
  for (r = 0; r < 32; ++r) {
    const uint32_t bit = r * bits;
    const uint32_t k = bit >> 5, s = bit & 31;
    for (l = 0; l < 4; ++l) {
      uint32_t x = in[4 * k + l] >> s;
      if (s + bits > 32) x |= in[4 * (k + 1) + l] << (32 - s);
      out[4 * r + l] = x & ((1u << bits) - 1u);
    }
  }
}
#endif

/* ------------------------------------------------------------ byte stream */



static uint8_t in_byte(orc_in* in) {
  if (in->p >= in->end) {
    in->bad = 1;
    return 0;
  }
  return *in->p++;
}

/* bytes_io<uint32_t>::vread — bytes_utils.hpp:176-206 */
static uint32_t in_vint(orc_in* in) {
  uint32_t out = 0, shift = 0, i;
  for (i = 0; i < 5; ++i) {
    const uint32_t b = in_byte(in);
    out |= (b & 0x7Fu) << shift;
    if (!(b & 0x80u)) break;
    shift += 7;
  }
  return out;
}
/* bytes_io<uint64_t>::vread */
static uint64_t in_vlong(orc_in* in) {
  uint64_t out = 0;
  uint32_t shift = 0, i;
  for (i = 0; i < 10; ++i) {
    const uint64_t b = in_byte(in);
    out |= (b & 0x7Fu) << shift;
    if (!(b & 0x80u)) break;
    shift += 7;
  }
  return out;
}

/* bitpack::read_block_impl32 — bitpack.hpp:149-177 */
static void in_block(orc_in* in, int layout, uint32_t* out) {
  const uint32_t bits = in_byte(in);
  uint32_t i;
  if (bits == 0) { /* ALL_EQUAL: std::fill_n(decoded, size, read_vint()) */
    const uint32_t v = in_vint(in);
    for (i = 0; i < ORC_BLOCK; ++i) out[i] = v;
    return;
  }
  if (bits > 32 || (size_t)(in->end - in->p) < 16u * bits) {
    in->bad = 1;
    return;
  }
  {
    uint32_t enc[ORC_BLOCK];
    memcpy(enc, in->p, 16u * bits); /* stream is byte-unaligned */
    in->p += 16u * bits;
    if (layout == ORC_LAYOUT_SIMD4)
      orc_unpack_simd4(enc, bits, out);
    else
      orc_unpack_scalar(enc, bits, out);
  }
}
/* bitpack::skip_block32 — bitpack.hpp:60-69 */
static void in_skip_block(orc_in* in) {
  const uint32_t bits = in_byte(in);
  if (bits == 0) {
    (void)in_vint(in);
  } else if (bits > 32 || (size_t)(in->end - in->p) < 16u * bits) {
    in->bad = 1;
  } else {
    in->p += 16u * bits;
  }
}

int64_t orc_read_block(const uint8_t* p, const uint8_t* end, int layout,
                       uint32_t* out128) {
  orc_in in = {p, end, 0};
  in_block(&in, layout, out128);
  return in.bad ? -1 : (int64_t)(in.p - p);
}
int64_t orc_skip_block(const uint8_t* p, const uint8_t* end) {
  orc_in in = {p, end, 0};
  in_skip_block(&in);
  return in.bad ? -1 : (int64_t)(in.p - p);
}

/* -------------------------------------------------------------- iterator */

/* CommonSkipWandData — formats_10.cpp:1962-1979: one size byte per scorer the field
 * was indexed with, then the payloads. */
static void in_skip_wand(orc_in* in, uint32_t wand_count) {
  uint64_t skip = 0;
  uint32_t i;
  for (i = 0; i < wand_count; ++i) skip += in_byte(in);
  if ((uint64_t)(in->end - in->p) < skip) {
    in->bad = 1;
    in->p = in->end;
  } else {
    in->p += skip;
  }
}

void orc_it_prepare_wand(orc_doc_iterator* it, const uint8_t* file, uint64_t len,
                         int layout, const orc_term_meta* m, int want_freq,
                         uint32_t wand_count) {
  orc_it_prepare(it, file, len, layout, m, want_freq);
  /* lists without a skip list carry their wand root in front of the tail
   * (written :686-688); a reader that does not use it skips it when the list is
   * shorter than one block (:2298-2301) — a 128-doc list reads its block first */
  if (m->docs_count > 1 && m->docs_count < ORC_BLOCK) in_skip_wand(&it->in, wand_count);
}

void orc_it_prepare(orc_doc_iterator* it, const uint8_t* file, uint64_t len,
                       int layout, const orc_term_meta* m, int want_freq) {
  memset(it, 0, sizeof *it);
  it->layout = layout;
  it->want_freq = want_freq;
  it->begin = ORC_BLOCK;
  it->in.end = file + len;
  if (m->docs_count == 1) { /* single_doc_iterator::prepare :1876-1890 */
    it->single = 1;
    it->next_single = 1u + (uint32_t)m->e_skip_start; /* min() + e_single_doc */
    it->freq = m->freq;
    it->in.p = it->in.end;
    return;
  }
  it->left = m->docs_count; /* :2248 */
  if (m->doc_start > len) {
    it->in.bad = 1;
    it->in.p = it->in.end;
  } else {
    it->in.p = file + m->doc_start; /* :2262 */
  }
  /* docs_count < 128 && wand disabled -> SkipWandData() (:2298-2301): see
   * orc_it_prepare_wand; this entry point is for fields indexed without scorers. */
}

/* doc_iterator_base::read_tail_block :1765-1792 */
static void it_read_tail(orc_doc_iterator* it) {
  uint32_t i = ORC_BLOCK - it->left;
  it->begin = i;
  for (; i < ORC_BLOCK; ++i) {
    const uint32_t v = in_vint(&it->in);
    if (it->field_no_freq) { /* :1786-1788 */
      it->docs[i] = v;
      it->freqs[i] = 1;
      continue;
    }
    it->docs[i] = v >> 1; /* shift_unpack_32 */
    if (v & 1u) {
      it->freqs[i] = 1;
    } else {
      it->freqs[i] = in_vint(&it->in);
    }
  }
  it->left = 0;
}

/* doc_iterator_base::refill :1741-1762 */
static void it_refill(orc_doc_iterator* it) {
  if (it->left >= ORC_BLOCK) {
    in_block(&it->in, it->layout, it->docs);
    if (it->field_no_freq)
      ; /* nothing follows the doc block (:1746-1750) */
    else if (it->want_freq)
      in_block(&it->in, it->layout, it->freqs);
    else
      in_skip_block(&it->in);
    it->begin = 0;
    it->left -= ORC_BLOCK;
  } else {
    it_read_tail(it);
  }
}

/* doc_iterator::next :2089-2119, single_doc_iterator::next :1861-1873 */
int orc_it_next(orc_doc_iterator* it) {
  if (it->single) {
    it->doc = it->next_single;
    it->next_single = UINT32_MAX;
    return it->doc != UINT32_MAX;
  }
  if (it->begin == ORC_BLOCK) {
    if (!it->left) {
      it->doc = UINT32_MAX; /* doc_limits::eof() */
      return 0;
    }
    it_refill(it);
    if (it->in.bad) {
      it->doc = UINT32_MAX;
      return 0;
    }
    it->doc += (it->doc == 0); /* :2103 */
  }
  it->freq = it->freqs[it->begin];
  it->doc += it->docs[it->begin++];
  return 1;
}

int64_t orc_decode_term(const uint8_t* doc_file, uint64_t len, int layout,
                        const orc_term_meta* meta, uint32_t* docs,
                        uint32_t* freqs, uint64_t cap) {
  return orc_decode_term_field(doc_file, len, layout, 1, meta, docs, freqs, cap);
}

int64_t orc_decode_term_field(const uint8_t* doc_file, uint64_t len, int layout,
                              int field_has_freq, const orc_term_meta* meta,
                              uint32_t* docs, uint32_t* freqs, uint64_t cap) {
  return orc_decode_term_wand(doc_file, len, layout, field_has_freq, 0, meta, docs, freqs, cap);
}

int64_t orc_decode_term_wand(const uint8_t* doc_file, uint64_t len, int layout,
                             int field_has_freq, uint32_t wand_count,
                             const orc_term_meta* meta, uint32_t* docs,
                             uint32_t* freqs, uint64_t cap) {
  orc_doc_iterator it;
  uint64_t n = 0;
  if (meta->docs_count == 0) return 0;
  if (!field_has_freq && freqs) return -3; /* cannot request FREQ from such a field */
  orc_it_prepare_wand(&it, doc_file, len, layout, meta, freqs != NULL, wand_count);
  it.field_no_freq = !field_has_freq;
  while (orc_it_next(&it)) {
    if (n >= cap) return -2;
    docs[n] = it.doc;
    if (freqs) freqs[n] = it.freq;
    ++n;
  }
  return it.in.bad ? -1 : (int64_t)n;
}

/* postings_reader::bit_union + the free ::bit_union — formats_10.cpp:3716-3806.
 * A separate code path of the reference (no iterator): read doc block, skip freq
 * block, `doc += delta; set_bit(set[doc / 64], doc % 64)`; then the vint tail.
 * Returns the sum of docs_count like the reference, or <0 on corruption. */
int64_t orc_bit_union(const uint8_t* doc_file, uint64_t len, int layout,
                      int has_freq, const orc_term_meta* metas,
                      uint32_t n_terms, uint64_t* set, uint64_t n_words) {
  return orc_bit_union_wand(doc_file, len, layout, has_freq, 0, metas, n_terms, set, n_words);
}

int64_t orc_bit_union_wand(const uint8_t* doc_file, uint64_t len, int layout,
                           int has_freq, uint32_t wand_count,
                           const orc_term_meta* metas, uint32_t n_terms,
                           uint64_t* set, uint64_t n_words) {
  uint32_t docs[ORC_BLOCK];
  uint64_t count = 0;
  uint32_t t;
  for (t = 0; t < n_terms; ++t) {
    const orc_term_meta* m = &metas[t];
    if (m->docs_count > 1) {
      orc_in in;
      uint32_t nb = m->docs_count / ORC_BLOCK, left = m->docs_count % ORC_BLOCK;
      uint32_t doc = 1; /* doc_limits::min() :3721 */
      if (m->doc_start > len) return -1;
      in.p = doc_file + m->doc_start;
      in.end = doc_file + len;
      in.bad = 0;
      if (m->docs_count < ORC_BLOCK) in_skip_wand(&in, wand_count); /* :3776-3779 */
      while (nb--) {
        uint32_t i;
        in_block(&in, layout, docs);
        if (has_freq) in_skip_block(&in);
        if (in.bad) return -1;
        for (i = 0; i < ORC_BLOCK; ++i) {
          doc += docs[i];
          if (doc / 64 < n_words) set[doc / 64] |= (uint64_t)1 << (doc % 64);
        }
      }
      while (left--) {
        uint32_t delta;
        if (has_freq) {
          const uint32_t v = in_vint(&in); /* shift_unpack_32 :3743-3745 */
          delta = v >> 1;
          if (!(v & 1u)) (void)in_vint(&in);
        } else {
          delta = in_vint(&in);
        }
        if (in.bad) return -1;
        doc += delta;
        if (doc / 64 < n_words) set[doc / 64] |= (uint64_t)1 << (doc % 64);
      }
      count += m->docs_count;
    } else if (m->docs_count == 1) {
      const uint32_t doc = 1u + (uint32_t)m->e_skip_start; /* e_single_doc :3797 */
      if (doc / 64 < n_words) set[doc / 64] |= (uint64_t)1 << (doc % 64);
      ++count;
    }
  }
  return (int64_t)count;
}

/* SkipReaderBase::Prepare (skip_list.cpp:111-156) + ReadState
 * (formats_10.cpp:1063-1080) for level 0 only. */
int64_t orc_read_skip0(const uint8_t* doc_file, uint64_t len,
                       const orc_term_meta* meta, uint32_t* last_docs,
                       uint64_t* next_block_ptrs, uint64_t cap,
                       uint32_t* num_levels) {
  return orc_read_skip0_wand(doc_file, len, 0, meta, last_docs, next_block_ptrs, cap,
                             num_levels, NULL, NULL);
}

/* With wand data: the root entry sits in front of `num_levels` (:2223) and every
 * skip entry ends with one size byte per scorer + the payloads (:990-999).  The
 * payload of scorer 0 (FreqNormSource::Read, wand_writer.hpp:318-334: vint freq
 * [+ vint(norm - freq)]) is returned per level-0 entry when max_freq/min_norm are
 * given. */
int64_t orc_read_skip0_wand(const uint8_t* doc_file, uint64_t len, uint32_t wand_count,
                            const orc_term_meta* meta, uint32_t* last_docs,
                            uint64_t* next_block_ptrs, uint64_t cap,
                            uint32_t* num_levels, uint32_t* max_freq,
                            uint32_t* norm_of_max) {
  return orc_read_skip0_pos(doc_file, len, wand_count, 0, meta, last_docs, next_block_ptrs,
                            cap, num_levels, max_freq, norm_of_max, NULL, NULL);
}

/* ... and for a field with POS every entry also holds vint(pend_pos) + vlong(Δpos_ptr)
 * (ReadState :1063-1080): positions pending at the block border and where the `.pos`
 * stream stands there. */
int64_t orc_read_skip0_pos(const uint8_t* doc_file, uint64_t len, uint32_t wand_count,
                           int field_has_pos, const orc_term_meta* meta, uint32_t* last_docs,
                           uint64_t* next_block_ptrs, uint64_t cap, uint32_t* num_levels,
                           uint32_t* max_freq, uint32_t* norm_of_max, uint32_t* pend_pos,
                           uint64_t* pos_ptrs) {
  uint64_t pos_ptr = meta->pos_start; /* CopyState(SkipState&, term_meta) :1097 */
  orc_in in;
  uint32_t levels, l;
  uint64_t n = 0, ptr;
  if (meta->docs_count <= ORC_BLOCK) return 0;
  if (meta->doc_start + meta->e_skip_start > len) return -1;
  in.p = doc_file + meta->doc_start + meta->e_skip_start;
  in.end = doc_file + len;
  in.bad = 0;
  in_skip_wand(&in, wand_count);
  levels = in_vint(&in);
  if (num_levels) *num_levels = levels;
  if (!levels) return 0;
  for (l = levels; l-- > 1;) { /* levels n..1: vlong length, bytes */
    const uint64_t length = in_vlong(&in);
    if (!length || (uint64_t)(in.end - in.p) < length) return -1;
    in.p += length;
  }
  {
    const uint64_t length = in_vlong(&in);
    const uint8_t* stop;
    if (!length || (uint64_t)(in.end - in.p) < length) return -1;
    stop = in.p + length;
    ptr = meta->doc_start;
    while (in.p < stop && !in.bad) {
      const uint32_t doc = in_vint(&in);
      ptr += in_vlong(&in);
      if (n >= cap) return -2;
      last_docs[n] = doc;
      next_block_ptrs[n] = ptr;
      if (field_has_pos) {
        const uint32_t pend = in_vint(&in);
        pos_ptr += in_vlong(&in);
        if (pend_pos) pend_pos[n] = pend;
        if (pos_ptrs) pos_ptrs[n] = pos_ptr;
      }
      if (wand_count) {
        uint32_t sizes[16], w, total = 0;
        if (wand_count > 16) return -1;
        for (w = 0; w < wand_count; ++w) total += (sizes[w] = in_byte(&in));
        if ((uint64_t)(in.end - in.p) < total) return -1;
        if (max_freq || norm_of_max) {
          orc_in pl = in;
          const uint8_t* start = pl.p;
          const uint32_t f = in_vint(&pl);
          uint32_t nrm = f;
          if ((uint32_t)(pl.p - start) != sizes[0]) nrm += in_vint(&pl);
          if (max_freq) max_freq[n] = f;
          if (norm_of_max) norm_of_max[n] = nrm;
        }
        in.p += total;
      }
      ++n;
    }
  }
  return in.bad ? -1 : (int64_t)n;
}

/* format_utils::check_header — format_utils.cpp:74-105 */
int64_t orc_check_doc_header(const uint8_t* f, uint64_t len, int32_t* version) {
  static const char name[] = "iresearch_10_postings_documents";
  const uint32_t nlen = (uint32_t)sizeof(name) - 1;
  orc_in in = {f, f + len, 0};
  uint32_t magic = 0, ver = 0, i, slen;
  for (i = 0; i < 4; ++i) magic = (magic << 8) | in_byte(&in);
  if (magic != 0x3fd76c17u) return -1;
  slen = in_vint(&in);
  if (slen != nlen || (uint64_t)(in.end - in.p) < nlen ||
      memcmp(in.p, name, nlen) != 0)
    return -1;
  in.p += nlen;
  for (i = 0; i < 4; ++i) ver = (ver << 8) | in_byte(&in);
  if (in.bad) return -1;
  if (version) *version = (int32_t)ver;
  return (int64_t)(in.p - f);
}

uint32_t orc_it_seek(orc_doc_iterator* it, uint32_t target) {
  while (it->doc < target) {
    if (!orc_it_next(it)) break;
  }
  return it->doc;
}


/* -------------------------------------------------------------- positions */

/* position_impl::prepare(const DocState&) :1472-1491 with the tail location
 * doc_iterator::prepare computes (:2270-2285; single_doc_iterator :1896-1913) */
void orc_pos_prepare(orc_pos_iterator* p, const uint8_t* pos_file, uint64_t len, int layout,
                     const orc_term_meta* m, int one_based) {
  memset(p, 0, sizeof *p);
  p->one_based = one_based;
  p->file = pos_file;
  p->layout = layout;
  p->in.end = pos_file + len;
  p->buf_pos = ORC_BLOCK;
  if (m->pos_start > len) {
    p->in.bad = 1;
    p->in.p = p->in.end;
  } else {
    p->in.p = pos_file + m->pos_start; /* pos_in_->seek(pos_start) :1488 */
  }
  if (m->freq < ORC_BLOCK)
    p->tail_start = m->pos_start;
  else if (m->freq == ORC_BLOCK)
    p->tail_start = UINT64_MAX; /* address_limits::invalid() */
  else
    p->tail_start = m->pos_start + m->pos_end;
  p->tail_length = m->freq % ORC_BLOCK;
}

/* position::refill :1653-1659, read_tail_block :1515-1537 (no payloads/offsets) */
static void pos_refill(orc_pos_iterator* p) {
  if ((uint64_t)(p->in.p - p->file) == p->tail_start) {
    uint32_t i;
    for (i = 0; i < p->tail_length; ++i) p->pos_deltas[i] = in_vint(&p->in);
  } else {
    in_block(&p->in, p->layout, p->pos_deltas);
  }
}

/* position::skip :1661-1680 */
static void pos_skip(orc_pos_iterator* p, uint32_t count) {
  uint32_t left = ORC_BLOCK - p->buf_pos;
  if (count >= left) {
    count -= left;
    while (count >= ORC_BLOCK) {
      in_skip_block(&p->in);
      count -= ORC_BLOCK;
    }
    pos_refill(p);
    p->buf_pos = 0;
    left = ORC_BLOCK;
  }
  if (count < left) p->buf_pos += count;
  p->value = 0; /* clear() */
}

/* position::notify + clear, as doc_iterator::next calls them (:2108-2112) */
void orc_pos_notify(orc_pos_iterator* p, uint32_t n) {
  p->pend_pos += n;
  p->value = 0;
}

/* position::next :1606-1633 */
int orc_pos_next(orc_pos_iterator* p, uint32_t freq) {
  if (p->pend_pos == 0) {
    p->value = UINT32_MAX;
    return 0;
  }
  if (p->pend_pos > freq) {
    pos_skip(p, (uint32_t)(p->pend_pos - freq));
    p->pend_pos = freq;
  }
  if (p->buf_pos == ORC_BLOCK) {
    pos_refill(p);
    p->buf_pos = 0;
  }
  if (p->one_based) p->value += (p->value == 0); /* :1623-1625 */
  p->value += p->pos_deltas[p->buf_pos];
  ++p->buf_pos;
  --p->pend_pos;
  return !p->in.bad;
}

/* position::seek :1578-1604 */
uint32_t orc_pos_seek(orc_pos_iterator* p, uint32_t freq, uint32_t target) {
  if (p->pend_pos > freq) {
    pos_skip(p, (uint32_t)(p->pend_pos - freq));
    p->pend_pos = freq;
  }
  while (p->value < target && p->pend_pos) {
    if (p->buf_pos == ORC_BLOCK) {
      pos_refill(p);
      p->buf_pos = 0;
    }
    if (p->one_based) p->value += (p->value == 0); /* :1589-1591 */
    p->value += p->pos_deltas[p->buf_pos];
    ++p->buf_pos;
    --p->pend_pos;
  }
  if (p->pend_pos == 0 && p->value < target) p->value = UINT32_MAX;
  return p->value;
}

/* Every position of every doc of one term: the doc iterator is advanced doc by doc and
 * the position attribute drained (`stride` = 1), or only every stride-th doc is drained
 * so that the skip() path (:1661-1680) runs too; positions of skipped docs are written as 0. */
int64_t orc_decode_positions(const uint8_t* doc_file, uint64_t len, const uint8_t* pos_file,
                             uint64_t pos_len, int layout, uint32_t wand_count,
                             const orc_term_meta* meta, uint32_t stride, uint32_t* out,
                             uint64_t cap) {
  return orc_decode_positions_v(doc_file, len, pos_file, pos_len, layout, wand_count, 0, meta,
                                stride, out, cap);
}

int64_t orc_decode_positions_v(const uint8_t* doc_file, uint64_t len, const uint8_t* pos_file,
                               uint64_t pos_len, int layout, uint32_t wand_count, int one_based,
                               const orc_term_meta* meta, uint32_t stride, uint32_t* out,
                               uint64_t cap) {
  orc_doc_iterator it;
  orc_pos_iterator pos;
  uint64_t n = 0, d = 0;
  if (meta->docs_count == 0) return 0;
  if (!stride) stride = 1;
  orc_it_prepare_wand(&it, doc_file, len, layout, meta, 1, wand_count);
  orc_pos_prepare(&pos, pos_file, pos_len, layout, meta, one_based);
  while (orc_it_next(&it)) {
    uint32_t k;
    orc_pos_notify(&pos, it.freq);
    if (n + it.freq > cap) return -2;
    if (d++ % stride == 0) {
      for (k = 0; k < it.freq; ++k) {
        if (!orc_pos_next(&pos, it.freq)) return -1;
        out[n + k] = pos.value;
      }
      if (orc_pos_next(&pos, it.freq) || pos.value != UINT32_MAX) return -4; /* eof after freq */
    } else {
      for (k = 0; k < it.freq; ++k) out[n + k] = 0;
    }
    n += it.freq;
  }
  return (it.in.bad || pos.in.bad) ? -1 : (int64_t)n;
}

/* check_header for `.pos` (prepare_input, formats_10.cpp:3369-3381) */
int64_t orc_check_pos_header(const uint8_t* f, uint64_t len, int32_t* version) {
  static const char name[] = "iresearch_10_postings_positions";
  const uint32_t nlen = (uint32_t)sizeof(name) - 1;
  orc_in in = {f, f + len, 0};
  uint32_t magic = 0, ver = 0, i, slen;
  for (i = 0; i < 4; ++i) magic = (magic << 8) | in_byte(&in);
  if (magic != 0x3fd76c17u) return -1;
  slen = in_vint(&in);
  if (slen != nlen || (uint64_t)(in.end - in.p) < nlen || memcmp(in.p, name, nlen) != 0)
    return -1;
  in.p += nlen;
  for (i = 0; i < 4; ++i) ver = (ver << 8) | in_byte(&in);
  if (in.bad) return -1;
  if (version) *version = (int32_t)ver;
  return (int64_t)(in.p - f);
}
