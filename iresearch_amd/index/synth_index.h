/* Synthetic IResearch segment builder (host only, no GPU, no oracle).
 *
 * Produces, for one segment, exactly the inputs the hot path consumes in the
 * reference:
 *   - the bytes of a `.doc` postings file as `irs::postings_writer` lays them
 *     out (reference: core/formats/formats_10.cpp:621-1025, bitpack.hpp:75-108),
 *     in either the scalar ("1_5") or the simdcomp 4-lane ("1_5simd") layout;
 *   - the `version10::term_meta` of every term
 *     (core/formats/formats_10_attributes.hpp:30-50);
 *   - the dense Norm2 column (core/index/norm.hpp:135-182): field length per
 *     doc, 1 byte wide;
 *   - the field statistics BM25/TF-IDF collectors read
 *     (core/search/bm25.cpp:52-58).
 *
 * The corpus is defined by a pure function of (seed, global doc id, token
 * position) so any thread count (and any segment split) yields identical
 * bytes.  See DESIGN.md "Synthetic index".
 */
#ifndef IRS_SYNTH_INDEX_H
#define IRS_SYNTH_INDEX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors irs::version10::term_meta (formats_10_attributes.hpp:30-50). */
typedef struct irs_synth_term_meta {
  uint32_t docs_count; /* irs::term_meta::docs_count  (formats.hpp:97)  */
  uint32_t freq;       /* irs::term_meta::freq (total tf, formats.hpp:100) */
  uint64_t doc_start;  /* offset of the term's postings in `.doc` */
  uint64_t pos_start;
  uint64_t pos_end;    /* address_limits::invalid() == UINT64_MAX when unset */
  uint64_t pay_start;
  uint64_t e_skip_start; /* union with e_single_doc (low 32 bits) */
} irs_synth_term_meta;

enum { IRS_SYNTH_LAYOUT_SCALAR = 0, IRS_SYNTH_LAYOUT_SIMD4 = 1 };
/* wand data kinds: which FreqNormProducer a scorer indexes with (wand_writer.hpp:130-148) */
enum {
  IRS_SYNTH_WAND_MAX_FREQ = 0, /* kWandTagMaxFreq: BM15, TFIDF without norms */
  IRS_SYNTH_WAND_MIN_NORM = 1, /* kWandTagMinNorm: BM25 (bm25.cpp:518)        */
  IRS_SYNTH_WAND_DIV_NORM = 2  /* kWandTagDivNorm: BM11, TFIDF with norms     */
};

typedef struct irs_synth_params {
  uint64_t seed;          /* 20260926 for every BASELINE config             */
  uint64_t first_doc;     /* global id (0-based) of this segment's doc 1    */
  uint32_t num_docs;      /* docs in this segment, local ids 1..num_docs    */
  uint32_t vocab_log2;    /* V = 2^vocab_log2 term ranks, Zipf s = 1        */
  uint32_t max_rank;      /* ranks 1..max_rank get posting lists            */
  uint32_t layout;        /* IRS_SYNTH_LAYOUT_*                             */
  uint32_t mean_len;      /* doc length ~ IrwinHall12 approx N(mean, sd)    */
  uint32_t stddev_len;    /*   clamped to [1,255]                           */
  uint32_t threads;       /* 0 = hardware_concurrency                       */
  uint32_t keep_postings; /* keep decoded (doc, tf) lists for verification  */
  uint32_t wand_count;    /* scorers the field is indexed with (0..8): wand data */
  uint32_t wand_kind;     /* IRS_SYNTH_WAND_* of every one of them           */
  uint32_t with_positions;/* field has IndexFeatures::POS: also emit the `.pos` stream
                             (position of a token = its 1-based index in the doc)   */
  uint32_t one_based_positions; /* write formats 1_0 / 1_2simd (PostingsFormat 0 / 1): the first
                             position of a doc is stored relative to pos_limits::min() = 1 */
  /* Optional CLUSTERED corpus (0 = off: the i.i.d. benchmark corpus, byte for byte): docs come
     in runs of `topic_docs` consecutive ids sharing a topic; a token is drawn from the topic's
     own `topic_terms` ranks (uniform among them, the ranks chosen by hashing the topic id)
     with probability topic_percent / 100, from the global Zipf distribution otherwise.  Posting
     lists of such a corpus are bursty, which is what block-max pruning feeds on. */
  uint32_t topic_docs;
  uint32_t topic_percent;
  uint32_t topic_terms;
  uint32_t reserved1;
} irs_synth_params;

typedef struct irs_synth_index irs_synth_index;

int irs_synth_build(const irs_synth_params* p, irs_synth_index** out);
void irs_synth_free(irs_synth_index* idx);

const uint8_t* irs_synth_doc_bytes(const irs_synth_index* idx, uint64_t* len);
/* `.pos` file image (NULL unless with_positions) */
const uint8_t* irs_synth_pos_bytes(const irs_synth_index* idx, uint64_t* len);
/* only when keep_postings && with_positions: the rank's positions, doc after doc */
int irs_synth_positions(const irs_synth_index* idx, uint32_t rank,
                        const uint32_t** positions, uint64_t* count);
/* norms[i] = field length of local doc id (i + 1) */
const uint8_t* irs_synth_norms(const irs_synth_index* idx, uint64_t* count);
/* metas[r - 1] = term_meta of rank r */
const irs_synth_term_meta* irs_synth_term_metas(const irs_synth_index* idx,
                                                uint32_t* count);
uint64_t irs_synth_docs_with_field(const irs_synth_index* idx);
uint64_t irs_synth_total_term_freq(const irs_synth_index* idx);
/* only when keep_postings: pointers into the builder's own arrays */
int irs_synth_postings(const irs_synth_index* idx, uint32_t rank,
                       const uint32_t** docs, const uint32_t** freqs,
                       uint32_t* count);

/* Encode one explicit posting list (docs ascending, local ids >= 1) the way
 * postings_writer::write does; used by tests to feed literal lists such as
 * tests/golden/postings_6098.txt.  `segment_docs` sizes the skip list
 * (skip_list.cpp:47-48).  Returns bytes written (term starts at offset 0 of
 * `out`), or a negative error. */
int64_t irs_synth_encode_term(const uint32_t* docs, const uint32_t* freqs,
                              uint32_t count, uint32_t segment_docs,
                              uint32_t layout, uint8_t* out, uint64_t out_cap,
                              irs_synth_term_meta* meta);

/* Same for a field indexed WITH scorers (formats 1_4/1_5 "wand data"): one
 * FreqNormProducer per scorer (core/formats/wand_writer.hpp:152-342) — a size byte
 * and a payload per scorer in front of the tail of lists without a skip list
 * (formats_10.cpp:686-688), in front of the skip levels (:778) and in every skip
 * entry (:990-999).  norms = 1-byte Norm2 column (norms[doc - 1]), required by the
 * MIN_NORM / DIV_NORM kinds. */
int64_t irs_synth_encode_term_wand(const uint32_t* docs, const uint32_t* freqs,
                                   uint32_t count, uint32_t segment_docs,
                                   uint32_t layout, const uint8_t* norms,
                                   const uint32_t* wand_kinds, uint32_t wand_count,
                                   uint8_t* out, uint64_t out_cap,
                                   irs_synth_term_meta* meta);

/* Same for a field with IndexFeatures::POS: `positions` holds Σ freqs entries (per doc
 * ascending, >= 1).  The term's `.pos` bytes (AddPosition formats_10.cpp:894-933, tail
 * :713-760; zero-based storage of formats 1_3+) go to `pos_out`; meta->pos_start is
 * relative to it, meta->pos_end as EndTerm sets it; skip entries carry pend_pos +
 * Δpos_ptr (:511-518). */
int64_t irs_synth_encode_term_pos(const uint32_t* docs, const uint32_t* freqs,
                                  const uint32_t* positions, uint32_t count,
                                  uint32_t segment_docs, uint32_t layout,
                                  const uint8_t* norms, const uint32_t* wand_kinds,
                                  uint32_t wand_count, uint8_t* out, uint64_t out_cap,
                                  uint8_t* pos_out, uint64_t pos_cap, uint64_t* pos_len,
                                  irs_synth_term_meta* meta);
/* ... with one_based != 0: the formats before 1_3 (one-based position storage, no wand data) */
int64_t irs_synth_encode_term_pos_v(const uint32_t* docs, const uint32_t* freqs,
                                    const uint32_t* positions, uint32_t count,
                                    uint32_t segment_docs, uint32_t layout,
                                    const uint8_t* norms, const uint32_t* wand_kinds,
                                    uint32_t wand_count, uint32_t one_based, uint8_t* out,
                                    uint64_t out_cap, uint8_t* pos_out, uint64_t pos_cap,
                                    uint64_t* pos_len, irs_synth_term_meta* meta);
/* header + body + footer of a `.doc` (is_pos == 0) or `.pos` file of either format family */
int64_t irs_synth_wrap_file(const uint8_t* body, uint64_t body_len, uint32_t layout,
                            uint32_t is_pos, uint32_t one_based, uint8_t* out, uint64_t out_cap,
                            uint64_t* body_offset);
int64_t irs_synth_wrap_pos_file(const uint8_t* body, uint64_t body_len,
                                uint32_t layout, uint8_t* out,
                                uint64_t out_cap, uint64_t* body_offset);

/* Wrap concatenated term bytes into a complete `.doc` file image:
 * header (format_utils.cpp:57-61) + body + footer (:63-67). Returns total
 * length; `body_offset` receives the header length to add to doc_start. */
int64_t irs_synth_wrap_doc_file(const uint8_t* body, uint64_t body_len,
                                uint32_t layout, uint8_t* out,
                                uint64_t out_cap, uint64_t* body_offset);

/* Corpus primitives exposed for tests. */
uint32_t irs_synth_doc_length(uint64_t seed, uint64_t global_doc,
                              uint32_t mean_len, uint32_t stddev_len);

/* Query workload: `n_queries` x `n_terms` distinct ranks, log-uniform in
 * [lo_rank, hi_rank]. ranks_out is row-major. */
int irs_synth_queries(uint64_t seed, uint32_t n_queries, uint32_t n_terms,
                      uint32_t lo_rank, uint32_t hi_rank, uint32_t* ranks_out);

/* ---- the other files of a segment the hot path is fed from (synth_dict.cpp) --------------- */
/* postings_writer::encode (formats_10.cpp:576-604) of n consecutive terms of ONE dictionary
 * block: the delta state starts from zeros.  Returns bytes written, <0 on error. */
int64_t irs_synth_term_meta_stream(const irs_synth_term_meta* metas, uint32_t n, uint32_t has_freq,
                                   uint32_t has_pos, uint32_t has_pay, uint8_t* out,
                                   uint64_t out_cap);
/* The `.tm` term dictionary of one field: `n` terms (bytes concatenated in `terms`, lengths in
 * `term_lens`, ascending) with their metas, as field_writer lays blocks out
 * (formats_burst_trie.cpp:1023-1196; min_block / max_block = 25 / 48 in the reference).
 * *root_start = file offset of the field's root block (what the term index would hold). */
int64_t irs_synth_term_dictionary(const uint8_t* terms, const uint32_t* term_lens,
                                  const irs_synth_term_meta* metas, uint32_t n,
                                  uint32_t has_freq, uint32_t has_pos, uint32_t has_pay,
                                  uint32_t min_block, uint32_t max_block, uint8_t* out,
                                  uint64_t out_cap, uint64_t* root_start);
/* columnstore2 `.csd` + `.csi` holding ONE anonymous fixed-length column (values of
 * `value_bytes` bytes for docs min_doc .. min_doc + n_docs - 1, `payload` = the feature's
 * header, e.g. a Norm2Header) — written block by block (kFixed, a fresh segment) or in one
 * piece (dense_fixed, a consolidated one) — followed by `lead_columns` named mask columns. */
int64_t irs_synth_columnstore(const uint8_t* values, uint32_t value_bytes, uint32_t n_docs,
                              uint32_t min_doc, const uint8_t* payload, uint32_t payload_len,
                              uint32_t dense_fixed, uint32_t lead_columns, uint8_t* csd_out,
                              uint64_t csd_cap, uint64_t* csd_len, uint8_t* csi_out,
                              uint64_t csi_cap, uint64_t* csi_len, uint32_t* column_id);

/* One field of a segment for irs_synth_segment_dictionary: what field_writer::write is handed. */
typedef struct irs_synth_field {
  const char* name;            /* fields must come in name order */
  uint32_t name_len;
  uint32_t index_features;     /* IndexFeatures: FREQ 1, POS 2, OFFS 4, PAY 8 */
  int64_t norm_column;         /* the field's Norm2 column id; < 0: no norms */
  uint64_t docs_with_field;    /* postings_writer::end_field's doc count */
  uint64_t wand_mask;          /* bit i: scorer i has wand data in this field */
  const uint8_t* terms;        /* as irs_synth_term_dictionary */
  const uint32_t* term_lens;
  const irs_synth_term_meta* metas;
  uint32_t n_terms;
} irs_synth_field;
/* The term dictionary `.tm` of ALL fields of a segment (one file: field_writer opens it once,
 * formats_burst_trie.cpp:1235-1290) and the term index `.ti` with the segment's feature list and
 * every field's record (EndField :1347-1432).  The FST of a field holds the root block's record
 * only.  Returns the number of fields written (fields without terms are skipped), < 0 on error. */
int64_t irs_synth_segment_dictionary(const irs_synth_field* fields, uint32_t n_fields,
                                     uint32_t min_block, uint32_t max_block, uint8_t* tm_out,
                                     uint64_t tm_cap, uint64_t* tm_len, uint8_t* ti_out,
                                     uint64_t ti_cap, uint64_t* ti_len);
/* The segment meta file `.sm` (SegmentMetaWriter::write, formats_10.cpp:3102-3140). */
int64_t irs_synth_segment_meta(const char* name, uint32_t name_len, uint64_t version,
                               uint64_t docs_count, uint64_t live_docs_count, uint64_t byte_size,
                               uint32_t has_column_store, const char* const* files,
                               const uint32_t* file_lens, uint32_t n_files, uint8_t* out,
                               uint64_t out_cap);
/* `.doc_mask` of a segment (DocumentMaskWriter::write, formats_10.cpp:3245-3268): the ids of its
 * deleted docs.  Returns bytes written, <0 on error / short buffer. */
int64_t irs_synth_document_mask(const uint32_t* docs, uint64_t n, uint8_t* out, uint64_t out_cap);

#ifdef __cplusplus
}
#endif
#endif
