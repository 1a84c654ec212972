/* oracle/ — CPU restatement of the IResearch query hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 * load this library; the product path (iresearch_amd/, include/) never does.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - block codec (orc_unpack_*, orc_pack_*): PINNED against the reference's own
 *     compiled code (oracle/_ref: core/utils/bit_packing.cpp and
 *     external/simdcomp/src/simdbitpacking.c) for every bit width, and against
 *     the literal vector of tests/utils/bit_packing_tests.cpp:102-114;
 *   - posting-list reader: restatement of formats_10.cpp reader code, checked by
 *     round trip against an independent emitter on the reference's own test
 *     lists (formats_10_tests.cpp:452-457, tests/resources/postings.txt);
 *   - scores: the ORDER of results is pinned by every ranking the reference's
 *     own tests assert on tests/resources/simple_sequential_order.json
 *     (bm25_test.cpp / tfidf_test.cpp, 16 vectors incl. two-segment ones:
 *     tests/cases.py REFERENCE_ORDERS); the DOC SETS of Or / And / min-match by the literal
 *     lists and expectations of the reference's iterator tests
 *     (tests/golden/boolean_golden.json), those of by_phrase and one phrase ranking by
 *     tests/golden/phrase_golden.json; score MAGNITUDES are "parity unpinned" —
 *     those tests hold no float literals; the formulas are
 *     restated line by line and cross-checked in double precision.
 */
#ifndef IRS_ORACLE_H
#define IRS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_LAYOUT_SCALAR = 0, ORC_LAYOUT_SIMD4 = 1 };

/* version10::term_meta — formats_10_attributes.hpp:30-50 */
typedef struct orc_term_meta {
  uint32_t docs_count;
  uint32_t freq;
  uint64_t doc_start;
  uint64_t pos_start;
  uint64_t pos_end;
  uint64_t pay_start;
  uint64_t e_skip_start; /* union with e_single_doc */
} orc_term_meta;

/* ---- block codec ---------------------------------- */
void orc_pack_scalar(const uint32_t* in128, uint32_t bits, uint32_t* out);
void orc_unpack_scalar(const uint32_t* in, uint32_t bits, uint32_t* out128);
void orc_pack_simd4(const uint32_t* in128, uint32_t bits, uint32_t* out);
void orc_unpack_simd4(const uint32_t* in, uint32_t bits, uint32_t* out128);
/* packed::at — bit_packing.hpp (random access into the scalar layout) */
uint32_t orc_at_scalar(const uint32_t* in, uint32_t i, uint32_t bits);
/* bitpack::read_block32<128> — returns bytes consumed, <0 on truncation */
int64_t orc_read_block(const uint8_t* p, const uint8_t* end, int layout,
                       uint32_t* out128);
int64_t orc_skip_block(const uint8_t* p, const uint8_t* end);

/* ---- posting list reader (a1/a4/a5/a6) ---------------------------------- */
/* Decodes the whole list of one term by driving the restated
 * doc_iterator::next() until eof. freqs may be NULL (iterator without
 * frequency: freq blocks are skipped, formats_10.cpp:1746-1750). Returns the
 * number of postings or <0 on corruption. */
int64_t orc_decode_term(const uint8_t* doc_file, uint64_t len, int layout,
                        const orc_term_meta* meta, uint32_t* docs,
                        uint32_t* freqs, uint64_t cap);
/* Same for a field indexed without IndexFeatures::FREQ when field_has_freq == 0
 * (no freq blocks, tail entries are plain vint deltas; freqs must be NULL). */
int64_t orc_decode_term_field(const uint8_t* doc_file, uint64_t len, int layout,
                              int field_has_freq, const orc_term_meta* meta,
                              uint32_t* docs, uint32_t* freqs, uint64_t cap);
/* ... and for a field indexed with `wand_count` scorers: formats 1_4/1_5 interleave
 * "wand data" (per scorer a size byte + a FreqNormProducer payload,
 * wand_writer.hpp:152-342) in front of short lists' tails, in front of the skip
 * levels and in every skip entry. */
int64_t orc_decode_term_wand(const uint8_t* doc_file, uint64_t len, int layout,
                             int field_has_freq, uint32_t wand_count,
                             const orc_term_meta* meta, uint32_t* docs,
                             uint32_t* freqs, uint64_t cap);
int64_t orc_bit_union_wand(const uint8_t* doc_file, uint64_t len, int layout,
                           int has_freq, uint32_t wand_count,
                           const orc_term_meta* metas, uint32_t n_terms,
                           uint64_t* set, uint64_t n_words);
int64_t orc_read_skip0_wand(const uint8_t* doc_file, uint64_t len, uint32_t wand_count,
                            const orc_term_meta* meta, uint32_t* last_docs,
                            uint64_t* next_block_ptrs, uint64_t cap,
                            uint32_t* num_levels, uint32_t* max_freq,
                            uint32_t* norm_of_max);
int64_t orc_read_skip0_pos(const uint8_t* doc_file, uint64_t len, uint32_t wand_count,
                           int field_has_pos, const orc_term_meta* meta, uint32_t* last_docs,
                           uint64_t* next_block_ptrs, uint64_t cap, uint32_t* num_levels,
                           uint32_t* max_freq, uint32_t* norm_of_max, uint32_t* pend_pos,
                           uint64_t* pos_ptrs);
/* ---- positions (field with IndexFeatures::POS) ------------------
 * All positions of one term, doc after doc, by driving the restated doc iterator and
 * its position attribute (formats_10.cpp:1457-1682).  stride > 1 drains only every
 * stride-th doc (others are skipped through position::skip and reported as 0).
 * Returns Σ freq, <0 on corruption. */
int64_t orc_decode_positions(const uint8_t* doc_file, uint64_t len, const uint8_t* pos_file,
                             uint64_t pos_len, int layout, uint32_t wand_count,
                             const orc_term_meta* meta, uint32_t stride, uint32_t* out,
                             uint64_t cap);
/* one_based != 0: formats 1_0..1_2 (PostingsFormat < POSITIONS_ZEROBASED): the reader adds
 * pos_limits::min() back before a doc's first delta (formats_10.cpp:1589-1591, 1623-1625) */
int64_t orc_decode_positions_v(const uint8_t* doc_file, uint64_t len, const uint8_t* pos_file,
                               uint64_t pos_len, int layout, uint32_t wand_count, int one_based,
                               const orc_term_meta* meta, uint32_t stride, uint32_t* out,
                               uint64_t cap);
int64_t orc_check_pos_header(const uint8_t* pos_file, uint64_t len, int32_t* version);
/* postings_reader::bit_union (formats_10.cpp:3716-3806): ORs bit `doc` into `set`
 * for every posting of every term; returns the sum of docs_count. */
int64_t orc_bit_union(const uint8_t* doc_file, uint64_t len, int layout,
                      int has_freq, const orc_term_meta* metas,
                      uint32_t n_terms, uint64_t* set, uint64_t n_words);
/* Level-0 skip entries of a term with docs_count > 128: absolute last doc of
 * each skipped block and absolute file offset of the following block. */
int64_t orc_read_skip0(const uint8_t* doc_file, uint64_t len,
                       const orc_term_meta* meta, uint32_t* last_docs,
                       uint64_t* next_block_ptrs, uint64_t cap,
                       uint32_t* num_levels);
/* check_header for `.doc` — format_utils.cpp:74-; returns body offset and
 * stores the PostingsFormat version, <0 on mismatch */
int64_t orc_check_doc_header(const uint8_t* doc_file, uint64_t len,
                             int32_t* version);

/* ---- scorers (a9/a10/a11) ------------------------------------------------ */
typedef struct orc_bm25_stats { /* BM25Stats — bm25.hpp:48-57 */
  float idf;
  float norm_const;
  float norm_length;
  float norm_cache[256];
} orc_bm25_stats;

void orc_bm25_collect(float k, float b, uint64_t docs_with_field,
                      uint64_t docs_with_term, uint64_t total_term_freq,
                      orc_bm25_stats* stats /* in/out: idf accumulates */);
float orc_tfidf_idf(uint64_t docs_with_field, uint64_t docs_with_term);

enum {
  ORC_SCORER_BM25 = 0, /* k, b as given; picks BM1/BM15/BM25 like bm25.cpp:447-455 */
  ORC_SCORER_TFIDF = 1 /* with_norms selects tfidf.cpp:307 */
};
enum { ORC_OP_OR = 0, ORC_OP_AND = 1, ORC_OP_MINMATCH = 2 /* + (min_match << 8) */,
       ORC_OP_PHRASE = 3 /* orc_search_phrase / orc_score_all_phrase only */ };
/* ScoreMergeType of the boolean filter (scorer.hpp:224-236), in bits 24..25 of `op` */
enum { ORC_MERGE_SUM = 0, ORC_MERGE_MAX = 1, ORC_MERGE_MIN = 2 };

#define ORC_NORM_LEGACY_F32 0x104u
typedef struct orc_segment {
  const uint8_t* doc_file;
  uint64_t doc_file_len;
  int32_t layout;
  uint32_t num_docs;
  const uint8_t* norms; /* dense Norm2 column, big-endian, doc 1 first; NULL = none */
  uint32_t norm_width;  /* 1, 2 or 4 bytes; ORC_NORM_LEGACY_F32: the legacy `Norm` feature,
                           little-endian floats 1/sqrt(len) (norm.hpp:46-70) */
  uint32_t wand_count;  /* scorers the field was indexed with (wand data to skip) */
  const uint8_t* pos_file; /* `.pos` image of a field with POS, or NULL */
  uint64_t pos_file_len;
  int32_t pos_one_based;   /* formats 1_0..1_2 */
  const uint32_t* doc_mask; /* the segment's DocumentMask (deleted doc ids, any order) or NULL:
                               every iterator is wrapped as SegmentReaderImpl::mask does
                               (core/index/segment_reader_impl.cpp:69-101, 286) */
  uint64_t doc_mask_count;
} orc_segment;

typedef struct orc_scorer {
  int32_t kind;
  float k;
  float b;
  int32_t with_norms; /* TFIDF only */
} orc_scorer;

typedef struct orc_hit {
  float score;
  uint32_t doc;
  uint32_t segment;
} orc_hit;

/* One query, restating utils/index-search.cpp:698-787 over `nsegs` segments:
 * by_term (n_terms == 1) / Or / And of by_term filters on one field.
 * metas[s * n_terms + t] = term_meta of query term t in segment s
 * (docs_count == 0 => term absent there).  Field statistics are summed over
 * segments as by_term::prepare does (term_filter.cpp:102-125).
 * out must hold k entries; returns the number of hits written (<= k). */
int64_t orc_search(const orc_segment* segs, uint32_t nsegs,
                   const orc_term_meta* metas, uint32_t n_terms, int32_t op,
                   const orc_scorer* scorer, const float* boosts /*may be NULL*/,
                   const uint64_t* docs_with_field /*per seg*/,
                   const uint64_t* total_term_freq /*per seg*/, uint32_t k,
                   orc_hit* out, uint64_t* hits_total);

/* Batch of queries sharing (op, n_terms, scorer, k), run by `threads` workers
 * popping tasks from one queue like index-search --threads (index-search.cpp
 * :673-691). metas is [q][s][t]; out is [q][k]; counts[q] = hits written. */
int64_t orc_search_batch(const orc_segment* segs, uint32_t nsegs,
                         const orc_term_meta* metas, uint32_t n_queries,
                         uint32_t n_terms, int32_t op, const orc_scorer* scorer,
                         const uint64_t* docs_with_field,
                         const uint64_t* total_term_freq, uint32_t k,
                         uint32_t threads, orc_hit* out, uint32_t* counts,
                         uint64_t* hits_total);

/* Exhaustive scoring of one query on ONE segment: scores[doc] for doc in
 * [0, num_docs], matched[doc] = 1 when the iterator returned the doc.  Global
 * statistics are supplied explicitly. Used to validate tie members. */
int64_t orc_score_all(const orc_segment* seg, const orc_term_meta* metas,
                      uint32_t n_terms, int32_t op, const orc_scorer* scorer,
                      const float* boosts, uint64_t docs_with_field,
                      const uint64_t* docs_with_term /*per term, global*/,
                      uint64_t total_term_freq, float* scores,
                      uint8_t* matched);

/* by_phrase with fixed offsets (FixedPhraseQuery, phrase_query.cpp:44-111;
 * PhraseIterator + FixedPhraseFrequency, phrase_iterator.hpp:75-166, 540-626):
 * conjunction of the terms' iterators, phrase frequency from their positions
 * (offsets[t] = position of term t relative to the first; offsets[0] == 0), score =
 * scorer(tf = phrase frequency) with the statistics of ALL phrase terms accumulated
 * into one stats blob (FixedPrepareCollect, phrase_filter.cpp:212-293: idf sums).
 * Same harness loop and outputs as orc_search. */
int64_t orc_search_phrase(const orc_segment* segs, uint32_t nsegs,
                          const orc_term_meta* metas, uint32_t n_terms,
                          const uint32_t* offsets, const orc_scorer* scorer, float boost,
                          const uint64_t* docs_with_field, const uint64_t* total_term_freq,
                          uint32_t k, orc_hit* out, uint64_t* hits_total);
/* Exhaustive form on one segment: phrase_freq[doc] (0 = no match) and scores[doc]. */
int64_t orc_score_all_phrase(const orc_segment* seg, const orc_term_meta* metas,
                             uint32_t n_terms, const uint32_t* offsets,
                             const orc_scorer* scorer, float boost, uint64_t docs_with_field,
                             const uint64_t* docs_with_term, uint64_t total_term_freq,
                             float* scores, uint32_t* phrase_freq);

/* ---- term dictionary + columnstore readers ------------------------ */
/* postings_writer_base::encode (formats_10.cpp:576-604): one term's stats, delta-coded against
 * `last` (zeroed at the start of a dictionary block), which is updated.  Returns bytes, <0. */
int64_t orc_encode_term_meta(const orc_term_meta* meta, orc_term_meta* last, int has_pos,
                             int has_pay, uint8_t* out, uint64_t cap);
/* postings_reader_base::decode (formats_10.cpp:3421-3456): `state` carries the previous term of
 * the block.  Returns bytes consumed, <0 on truncation. */
int64_t orc_decode_term_meta(const uint8_t* in, uint64_t len, int has_freq, int has_pos,
                             int has_pay, orc_term_meta* state);
/* The term iterator's walk of `.tm` from the field's root block (what the term index hands
 * out): formats_burst_trie.cpp block_iterator / term_iterator.  See dict_oracle.cpp. */
int orc_walk_term_dictionary(const uint8_t* tm, uint64_t len, uint64_t root_start, int has_freq,
                             int has_pos, int has_pay, uint32_t* n_terms, uint64_t* term_bytes,
                             uint32_t* term_lens, uint8_t* terms, orc_term_meta* metas);
/* columnstore2 reader: the fixed-length column `column_id` (columnstore2.cpp:1746-1830,
 * 650-789, 792-1011). */
int orc_read_fixed_column(const uint8_t* csi, uint64_t csi_len, const uint8_t* csd,
                          uint64_t csd_len, uint32_t column_id, uint32_t* value_bytes,
                          uint32_t* min_doc, uint32_t* docs_count, uint8_t* payload,
                          uint32_t payload_cap, uint32_t* payload_len, uint8_t* values,
                          uint64_t values_cap);

/* One field's record of the term index `.ti` (term_reader_base::prepare,
 * formats_burst_trie.cpp:1509-1545) + the root block the field's FST maps the empty prefix to
 * (ImmutableFstImpl::Read utils/fstext/immutable_fst.hpp:136-203; block_iterator's header
 * :1751-1764). */
typedef struct orc_field_record {
  char name[64];
  uint32_t name_len;
  uint32_t index_features;
  int64_t norm_column;       /* the column id stored for "iresearch::norm2", -1: none */
  uint64_t terms_count, docs_count, total_doc_freq, total_term_freq, wand_mask;
  uint64_t root_start;
  uint32_t root_meta;        /* 1 terms, 2 sub-blocks, 4 floor */
  uint32_t root_floor_blocks;/* further floor blocks of the root group */
} orc_field_record;
/* field_reader::prepare (formats_burst_trie.cpp:3323-3440), the `.ti` half.  *count = fields in
 * the file; at most `cap` records are written.  0 ok, <0 corrupt / unsupported. */
int orc_read_term_index(const uint8_t* ti, uint64_t len, orc_field_record* out, uint32_t cap,
                        uint32_t* count, uint32_t* segment_index_features);
/* SegmentMetaReader::read (formats_10.cpp:3147-3218). */
int orc_read_segment_meta(const uint8_t* sm, uint64_t len, uint64_t* docs_count,
                          uint64_t* live_docs_count, uint32_t* has_column_store, uint32_t* n_files);
/* DocumentMaskReader::read (formats_10.cpp:3275-3312): the deleted doc ids of `.doc_mask` in file
 * order; returns their number (at most `cap` are written), <0 corrupt. */
int64_t orc_read_document_mask(const uint8_t* dm, uint64_t len, uint32_t* docs, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
