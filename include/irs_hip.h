/* irs_hip.h — C ABI of the MI355X-native IResearch query-execution path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no
 * exceptions, no STL, no torch types.  A C++ adapter deriving
 * irs::postings_reader / irs::filter::prepared binds these entry points (see
 * INTEGRATION.md); each function below names the reference interface it
 * replaces (paths relative to the IResearch tree, v1.3).
 *
 * Device: AMD gfx950 (MI355X) only.  Every entry point fails with
 * IRS_HIP_EHIP when no such device is usable — there is no CPU fallback.
 */
#ifndef IRS_HIP_H
#define IRS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IRS_HIP_ABI_VERSION 12
#define IRS_HIP_BLOCK_SIZE 128u  /* postings per block, formats_10.cpp:90 */
#define IRS_HIP_MAX_TERMS 16u    /* terms per boolean query               */
#define IRS_HIP_MAX_K 4096u      /* largest top-k                         */
#define IRS_HIP_MAX_PHRASE_TERMS 8u /* terms of one by_phrase query       */
#define IRS_HIP_NO_TERM 0xFFFFFFFFu
#define IRS_HIP_POS_OFFSETS 1u
#define IRS_HIP_POS_PAYLOADS 2u
#define IRS_HIP_NORM2 0u
#define IRS_HIP_NORM_LEGACY 1u
/* irs::Scorer::WandType (scorer.hpp:196-201) of the scorer a field's wand data was written by */
#define IRS_HIP_WAND_NONE 0u      /* unknown / no wand data                               */
#define IRS_HIP_WAND_DIV_NORM 1u  /* TFIDF with norms, BM11: (freq, norm) of the doc with the
                                     largest freq / norm (wand_writer.hpp:181-186)         */
#define IRS_HIP_WAND_MAX_FREQ 2u  /* BM15, TFIDF without norms: the largest frequency      */
#define IRS_HIP_WAND_MIN_NORM 3u  /* BM25: largest frequency, smallest norm                */

/* Errors replace the reference's exceptions (io_error / index_error,
 * formats_10.cpp:158-160, 3410-3415): never thrown across this boundary. */
typedef enum irs_hip_status {
  IRS_HIP_OK = 0,
  IRS_HIP_EINVAL = -1,       /* illegal_argument                          */
  IRS_HIP_ECORRUPT = -2,     /* index_error: malformed `.doc` bytes       */
  IRS_HIP_ENOMEM = -3,
  IRS_HIP_EHIP = -4,         /* HIP runtime failure / no gfx950 device    */
  IRS_HIP_EOVERFLOW = -5,    /* candidates exceed the memory budget even after the exact re-run */
  IRS_HIP_EUNSUPPORTED = -6,
  IRS_HIP_EPEER = -7         /* a batch with a communicator (irs_hip_batch_set_comm): ANOTHER rank
                              * could not re-execute its batch; no rank did, the results are not
                              * valid (this rank's own failure is reported as what it was) */
} irs_hip_status;

typedef enum irs_hip_layout {
  IRS_HIP_LAYOUT_SCALAR = 0, /* formats "1_0".."1_5": format_traits, formats_10.cpp:86-131   */
  IRS_HIP_LAYOUT_SIMD4 = 1   /* "1_2simd".."1_5simd": format_traits_sse4, :4122-4157         */
} irs_hip_layout;

/* irs::version10::term_meta — core/formats/formats_10_attributes.hpp:30-50,
 * as filled by postings_reader_base::decode (formats_10.cpp:3421-3456). */
typedef struct irs_hip_term_meta {
  uint32_t docs_count;   /* irs::term_meta::docs_count */
  uint32_t freq;         /* irs::term_meta::freq       */
  uint64_t doc_start;
  uint64_t pos_start;
  uint64_t pos_end;
  uint64_t pay_start;
  uint64_t e_skip_start; /* union { doc_id_t e_single_doc; uint64_t e_skip_start; } */
} irs_hip_term_meta;

/* What postings_reader::prepare (formats_10.cpp:3352-3419) and the Norm2
 * column reader (norm.hpp:210-251, columnstore2.cpp:650-789) see of one
 * segment.  All pointers are HOST pointers borrowed for the call only. */
typedef struct irs_hip_segment_desc {
  int32_t device;           /* HIP device ordinal                                  */
  int32_t layout;           /* irs_hip_layout                                      */
  const uint8_t* doc_file;  /* whole `.doc` file: index_input::read_buffer(0, len,  */
  uint64_t doc_file_len;    /*   BufferHint::PERSISTENT) (data_input.hpp:136-137)  */
  uint32_t num_docs;        /* docs in the segment; ids are 1..num_docs            */
  uint32_t has_freq;        /* field indexed with IndexFeatures::FREQ              */
  const uint8_t* norms;     /* dense Norm2 column bytes (big-endian values), or NULL */
  uint32_t norm_width;      /* 1, 2 or 4 (Norm2Header::num_bytes, norm.hpp:83-125) */
  uint32_t norm_min_doc;    /* header.min: doc id of norms[0] (normally 1)         */
  uint64_t norm_count;      /* number of values in `norms`                         */
  const irs_hip_term_meta* terms; /* the field's term table (term dictionary walk) */
  uint32_t num_terms;
  uint32_t wand_count;      /* scorers the field was indexed with — the `wand_count` the
                             * reference passes to postings_reader::iterator()
                             * (formats.hpp:160-168): formats 1_4/1_5 interleave that many
                             * (size byte, payload) pairs of "wand data" in front of short
                             * lists' tails (formats_10.cpp:686-688, skipped as :2298-2301)
                             * and inside the skip data (never read here). 0..16. */
  const uint8_t* pos_file;  /* whole `.pos` file of a field with IndexFeatures::POS (and no
                             * offsets / payloads); zero-based (formats 1_3+) or one-based
                             * (1_0..1_2) position storage is told from the file's version
                             * (formats_10.cpp:283-304) — what postings_reader::prepare opens
                             * as pos_in_ (:3369-3381); NULL: no positions, no phrase queries */
  uint64_t pos_file_len;
  uint32_t pos_features;    /* what else the field stores per position: IRS_HIP_POS_OFFSETS
                             * (IndexFeatures::OFFS), IRS_HIP_POS_PAYLOADS (IndexFeatures::PAY).
                             * Both change the vint tail of `.pos` (formats_10.cpp:728-760) and
                             * are answered with IRS_HIP_EUNSUPPORTED for now instead of being
                             * mis-read; 0 = positions only */
  uint32_t norm_kind;       /* IRS_HIP_NORM2 (0): `norms` are Norm2 values, big-endian integers of
                             * norm_width bytes (norm.hpp:128-251) — the field length in tokens.
                             * IRS_HIP_NORM_LEGACY (1): the legacy `Norm` feature (norm.hpp:46-70):
                             * one little-endian float per doc, 1/sqrt(length) (norm_width 4; the
                             * adapter decodes the sparse zvfloat column into this dense array,
                             * Norm::DEFAULT() = 1 for docs without a value) */
  uint32_t wand_type;       /* IRS_HIP_WAND_*: what scorer 0 of the field's wand data wrote
                             * (Scorer::wand_type of the scorer the field was indexed with).  Only
                             * MAX_FREQ and MIN_NORM pairs bound every score function
                             * (Scorer::compatible, scorer.cpp:31-64: a DivNorm index is refused
                             * for MaxFreq / MinNorm queries) — those are read from the index;
                             * with DIV_NORM or NONE the per-block (max freq, min norm) pairs
                             * are derived from the postings, which is always sound. */
  const uint32_t* doc_mask; /* the segment's DocumentMask: ids of its deleted docs, in any order, as
                             * index_utils::ReadDocumentMask (index_utils.cpp:476) fills it from the
                             * `.doc_mask` file (DocumentMaskReader::read, formats_10.cpp:3275-3312);
                             * NULL / 0: none.  The reference filters every iterator a caller wraps
                             * with SegmentReaderImpl::mask (segment_reader_impl.cpp:69-101, 286:
                             * MaskDocIterator skips the docs the mask contains); a batch cannot be
                             * wrapped afterwards — it only yields the k best docs — so the mask
                             * belongs to the segment here: no query of a batch (Or / And /
                             * min-match / by_phrase) returns, counts or scores a masked doc, and
                             * irs_hip_bit_union does not set its bit.  The postings-level surfaces
                             * (irs_hip_decode_term / _positions / _term_directory: what
                             * postings_reader::iterator yields) are not filtered, and neither are
                             * the statistics of the scorers (the reference's are index statistics
                             * too).  Ids outside 1..num_docs are ignored. */
  uint64_t doc_mask_count;
} irs_hip_segment_desc;

typedef struct irs_hip_segment irs_hip_segment; /* opaque, immutable after open */

/* Replaces postings_reader::prepare + the per-iterator reopen()/seek()
 * (formats_10.cpp:3352-3419, 2251-2262): stages the `.doc` bytes and the norm
 * column into HBM and builds the per-term block directory ON THE GPU by
 * walking block headers (bitpack.hpp:52-69).  Validates the file header
 * (format_utils.cpp:74-105) and that `layout` matches its version. */
int irs_hip_segment_open(const irs_hip_segment_desc* desc, irs_hip_segment** out);
void irs_hip_segment_close(irs_hip_segment* seg);

/* postings_reader::CountMappedMemory analogue: bytes resident in HBM.  Grows after open by the
 * tables a segment builds on first use and keeps: the block-max pairs of irs_hip_batch_set_wand
 * (8 bytes per 128-posting block) and the posting-order copy of a 1-byte Norm2 column (one byte
 * per posting of the whole segment — built by the first batch that joins posting streams, or whose
 * conjunctions / phrases score with norms). */
uint64_t irs_hip_segment_device_bytes(const irs_hip_segment* seg);
/* SubReader::live_docs_count (index_reader.hpp): num_docs minus the docs of doc_mask. */
uint64_t irs_hip_segment_live_docs(const irs_hip_segment* seg);

/* Replaces `postings_reader::iterator(...)` + `while (it->next())`
 * (formats_10.cpp:3491-3533, 2089-2119): decodes the whole posting list of
 * term ordinal `term` bit-exactly into host arrays. freqs may be NULL
 * (iterator requested without IndexFeatures::FREQ). */
int irs_hip_decode_term(irs_hip_segment* seg, uint32_t term, uint32_t* docs,
                        uint32_t* freqs, uint32_t cap, uint32_t* count);

/* Replaces draining the `position` attribute behind `while (it->next())`
 * (position::next, formats_10.cpp:1606-1633, fed by doc_iterator::next :2106-2113):
 * every position of every doc of term ordinal `term`, doc after doc — term_meta::freq
 * values in all (freqs from irs_hip_decode_term say where each doc's run ends).
 * Needs a segment opened with pos_file. */
int irs_hip_decode_positions(irs_hip_segment* seg, uint32_t term, uint32_t* positions,
                             uint64_t cap, uint64_t* count);

/* postings_reader::bit_union (core/formats/formats_10.cpp:3716-3806; virtual at
 * core/formats/formats.hpp:182-190): ORs bit `doc` into the caller's bitset
 * (`size_t* set` in the reference: 64-bit little-endian words, bit index = doc id,
 * so the set needs num_docs + 1 bits) for every posting of every listed term (of every doc that
 * is not in the segment's doc_mask);
 * freq blocks are skipped.  IRS_HIP_NO_TERM entries are ignored.  *count receives
 * what the reference returns: the SUM of the terms' docs_count (not a popcount).
 * Docs at or beyond 64 * n_words are not representable and are dropped. */
int irs_hip_bit_union(irs_hip_segment* seg, const uint32_t* terms, uint32_t n_terms,
                      uint64_t* set, uint64_t n_words, uint64_t* count);

/* The POPULATIONS of several such unions in one call, the bitsets never leaving the device: set i
 * is the union of the postings of terms[offsets[i] .. offsets[i + 1]) over doc ids 1..num_docs
 * (deleted docs left out), counts[i] its number of docs — the `hits` of a multi-term filter
 * (MultiTermQuery::execute, multiterm_query.cpp:112-184: scored iterators + one
 * lazy_bitset_iterator over the unscored terms; index-search counts what the iterator yields,
 * utils/index-search.cpp:745-779) when its top k comes from the scored terms' disjunction and
 * only the count is wanted from the rest: 8 bytes per filter cross PCIe instead of a bit per doc. */
int irs_hip_bit_union_counts(irs_hip_segment* seg, const uint32_t* terms, const uint32_t* offsets,
                             uint32_t n_sets, uint64_t* counts);

/* The block directory of one term, for inspection/tests: absolute last doc id
 * and `.doc` byte offset of every full 128-doc block — the information the
 * reference keeps in skip level 0 (formats_10.cpp:501-533). */
int irs_hip_term_directory(irs_hip_segment* seg, uint32_t term,
                           uint32_t* last_docs, uint64_t* offsets, uint32_t cap,
                           uint32_t* count);

/* ------------------------------------------------------------ queries -- */

typedef enum irs_hip_op {
  IRS_HIP_OP_OR = 0, /* irs::Or / by_term: disjunction.hpp MakeDisjunction :1411-1467 */
  IRS_HIP_OP_AND = 1, /* irs::And: conjunction.hpp MakeConjunction :436-490 — executed block
                         by block of the rarest term (Conjunction::converge :207-223)    */
  IRS_HIP_OP_MINMATCH = 2, /* irs::Or with min_match_count: MinMatchQuery::execute
                             (boolean_query.cpp:212-247) -> min_match_iterator =
                             block_disjunction<kMinMatch> (disjunction.hpp:1378-1383)   */
  IRS_HIP_OP_PHRASE = 3   /* irs::by_phrase of plain terms: FixedPhraseQuery::execute
                             (phrase_query.cpp:44-111) -> PhraseIterator<Conjunction,
                             FixedPhraseFrequency> (phrase_iterator.hpp:75-166, 540-626).
                             terms[first_term + i] = i-th phrase term with its
                             phrase_offset; every entry carries the SAME scorer values: the
                             phrase's one stats blob, into which every term's statistics
                             were finished (phrase_filter.cpp:281-287: idf sums), times
                             the filter boost.  score = that scorer at tf = phrase
                             frequency.  An absent term empties the query in that segment.
                             A batch holds phrase queries only, or none.               */
} irs_hip_op;

/* Which ScoreFunction Scorer::prepare_scorer would have built. */
typedef enum irs_hip_scorer_kind {
  IRS_HIP_SCORE_BM25 = 0,      /* bm25.cpp:321-364; norm path by segment norm_width (:466-476),
                                  legacy Norm column => :333-337, 242-249;
                                  no norm column => norm == 1 (:487-489)               */
  IRS_HIP_SCORE_BM15 = 1,      /* bm25.cpp:288-319 (b == 0)                            */
  IRS_HIP_SCORE_BM1 = 2,       /* bm25.cpp:262-286 (k == 0): constant                  */
  IRS_HIP_SCORE_TFIDF = 3,     /* tfidf.cpp:185-187, 251                                */
  IRS_HIP_SCORE_TFIDF_NORM = 4 /* tfidf.cpp:253, normalize() == true                   */
} irs_hip_scorer_kind;

/* One query term = the (term cookie, stats blob, boost) triple TermQuery::execute
 * hands to postings()/CompileScore (term_query.cpp:35-74), flattened. */
typedef struct irs_hip_term_scorer {
  uint32_t term;      /* ordinal in the segment's term table; IRS_HIP_NO_TERM = absent here */
  int32_t kind;       /* irs_hip_scorer_kind                                            */
  float c0;           /* BM25: boost*(k+1)*idf (bm25.cpp:201); TFIDF: boost*idf (tfidf.cpp:199) */
  float norm_const;   /* BM25Stats::norm_const  (bm25.hpp:52)                            */
  float norm_length;  /* BM25Stats::norm_length (bm25.hpp:54)                            */
  uint32_t phrase_offset; /* IRS_HIP_OP_PHRASE: position of the term relative to the phrase's
                             first term (FixedPhraseQuery::positions_t, phrase_filter.cpp
                             :279-284; 0 for the first); ignored by the other ops          */
} irs_hip_term_scorer;

/* boolean_filter::merge_type() — irs::ScoreMergeType (scorer.hpp:224-236) of an Or / And:
 * how the scores of the sub-queries on one doc combine.  The mergers are the reference's
 * (scorer.hpp:390-423), including what kMin does in a disjunction: it merges with the 0 of
 * every sub-query that is not on the doc, so an Or of two scores min(a, b) where both match
 * and 0 elsewhere, and an Or of three or more (block_disjunction's zeroed score buffer,
 * disjunction.hpp:1308-1351) scores 0 everywhere.  kNoop (no scores) is not offered. */
typedef enum irs_hip_merge {
  IRS_HIP_MERGE_SUM = 0,
  IRS_HIP_MERGE_MAX = 1,
  IRS_HIP_MERGE_MIN = 2
} irs_hip_merge;

typedef struct irs_hip_query {
  int32_t op;          /* irs_hip_op                                   */
  uint32_t n_terms;    /* 1..IRS_HIP_MAX_TERMS (PHRASE: ..IRS_HIP_MAX_PHRASE_TERMS) */
  uint32_t first_term; /* index of the first entry in the `terms` array */
  uint32_t k;          /* top-k, 1..IRS_HIP_MAX_K (index-search --topN) */
  uint32_t min_match;  /* IRS_HIP_OP_MINMATCH: Or::min_match_count(); else ignored */
  uint32_t merge;      /* irs_hip_merge (OR / AND / MINMATCH); PHRASE: IRS_HIP_MERGE_SUM */
} irs_hip_query;

/* (score, segment-local doc) exactly as utils/index-search.cpp:745-787 keeps. */
typedef struct irs_hip_hit {
  float score;
  uint32_t doc;
} irs_hip_hit;

typedef struct irs_hip_batch irs_hip_batch; /* opaque: one batch of queries on one segment */

/* Replaces, for a whole batch of prepared queries on one segment,
 *   filter::prepared::execute(ExecutionContext{segment, scorers, wand}) and the
 *   harness loop `while (docs->next()) { score; heap }` + final sort
 *   (filter.hpp:52-78, utils/index-search.cpp:719-787).
 * create : validates, uploads descriptors, sizes scratch (no query work).
 * run    : enqueues every kernel of the batch on `stream` (a hipStream_t, may
 *          be NULL = default stream); asynchronous.
 * results: waits for the stream and copies the per-query top-k to the host:
 *          hits[q * k_stride + i], i < counts[q], ordered (score desc, doc asc);
 *          total_hits[q] = number of matching docs (index-search `hits=`).
 * Results are deterministic: ties are broken by ascending doc id.
 * Lifetime / threading: a segment is immutable after open and may be shared by any
 * number of threads; a batch belongs to one thread at a time and must be destroyed
 * before its segment is closed.  results / results_to_device also verify the run
 * (candidate buffer, threshold estimate) and transparently re-execute the batch when
 * that check fails — irs_hip_batch_reruns counts those; results are exact either way. */
int irs_hip_batch_create(irs_hip_segment* seg, const irs_hip_query* queries,
                         uint32_t n_queries, const irs_hip_term_scorer* terms,
                         uint32_t n_term_entries, irs_hip_batch** out);
/* The same batch of queries over SEVERAL segments of one device in one go — what the
 * harness loop `for (auto& segment : reader)` (utils/index-search.cpp:719-779) does
 * segment after segment.  Every kernel is launched once for all (segment, query) pairs, so
 * small segments do not pay per-segment launch tails.  `terms` holds n_segs consecutive
 * arrays of n_term_entries entries: the same scorers (statistics are index-global,
 * term_filter.cpp:102-125) with each segment's own term ordinals.  All result calls then
 * index by unit = segment * n_queries + query: hits[unit * k_stride + i], counts[unit],
 * total_hits[unit]; irs_hip_merge_topk turns the per-segment lists into the global top-k.
 * Segments must live on the same device and use the same block layout. */
int irs_hip_batch_create_multi(irs_hip_segment* const* segs, uint32_t n_segs,
                               const irs_hip_query* queries, uint32_t n_queries,
                               const irs_hip_term_scorer* terms, uint32_t n_term_entries,
                               irs_hip_batch** out);
/* Executes the batch on `stream` (NULL: the default stream).  Asynchronous twice over: the
 * kernels are only queued, and the host half of a run — dealing the units to the kernels, building
 * the batch's posting streams and work lists, queueing uploads and launches (about 1 ms per 1000
 * queries) — is handed to a worker thread of the library (one per device), so that a serving loop
 * prepares batch i + 1 while batch i is being queued and batch i - 1 executes (the reference runs
 * its tasks on --threads workers, index-search.cpp:673-722).  Every other call on the batch waits
 * for that hand-over first; a failure of the run is returned by the next such call (results,
 * device_results, timings, destroy ...).  IRS_HIP_ASYNC_RUN=0 in the environment, or
 * irs_hip_batch_set_async(batch, 0) for one batch, keeps the host half on the caller's thread
 * (and the status in this call's return value).
 * What asynchronous means for the caller's own work on `stream`: when this returns, nothing of
 * the run may be on the stream yet.  An event the caller records, a kernel it launches or a
 * hipStreamSynchronize it makes right after is NOT ordered behind the run — get behind it with a
 * call on the batch (irs_hip_batch_device_results waits for the run; irs_hip_batch_results_to_device
 * / _to_host queue their copies behind it), or switch the hand-over off for that batch.  The same
 * holds for a device pointer obtained from irs_hip_batch_device_results BEFORE a re-run of the
 * batch: use it only behind a later call on the batch. */
int irs_hip_batch_run(irs_hip_batch* batch, void* stream);
/* Per batch: 1 = hand the host half of irs_hip_batch_run to the library's worker thread, 0 = keep
 * it on the caller's thread, -1 = the process default (on; IRS_HIP_ASYNC_RUN=0 turns it off).  A
 * batch with a communicator (irs_hip_batch_set_comm) issues its collectives — also those of a
 * recovery re-run — in the order of the calls made on this device either way. */
int irs_hip_batch_set_async(irs_hip_batch* batch, int enable);
/* Optional: queue the PLANNING stage of the batch's next run (tile tables, work items: what
 * building the iterator tree is to filter::prepared::execute) on `stream` now; the next
 * irs_hip_batch_run then only waits for it (an event) and starts with scoring.  The stage reads
 * and writes nothing another batch's run touches, so with two streams a caller overlaps the
 * planning of batch i+1 with the scoring kernels of batch i.  Without this call run() plans
 * inline, as before. */
int irs_hip_batch_plan(irs_hip_batch* batch, void* stream);
int irs_hip_batch_results(irs_hip_batch* batch, irs_hip_hit* hits,
                          uint32_t k_stride, uint32_t* counts,
                          uint64_t* total_hits);
/* Device-resident results for a caller that merges on the GPU (RCCL path): waits for
 * and verifies the run like `results`, then hands out the batch's own buffers:
 * d_hits is [n_queries][k_max] irs_hip_hit, d_counts [n_queries] uint32. */
int irs_hip_batch_device_results(irs_hip_batch* batch, void** d_hits,
                                 void** d_counts, uint32_t* k_max);
/* Same, copied (device to device, async on `stream`) into caller-owned device
 * buffers: d_hits [n_queries][k_max] irs_hip_hit, d_counts [n_queries] uint32. */
int irs_hip_batch_results_to_device(irs_hip_batch* batch, void* d_hits,
                                    void* d_counts, void* stream);
void irs_hip_batch_destroy(irs_hip_batch* batch);

/* Convenience: create + run + results + destroy. */
int irs_hip_query_batch(irs_hip_segment* seg, const irs_hip_query* queries,
                        uint32_t n_queries, const irs_hip_term_scorer* terms,
                        uint32_t n_term_entries, irs_hip_hit* hits,
                        uint32_t k_stride, uint32_t* counts,
                        uint64_t* total_hits);

/* The checked results of the batch's last run on their way to page-locked HOST memory owned by
 * the batch — where the reference's harness ends (index-search.cpp:782-807) — without stalling the
 * caller: results_to_host verifies the run (like irs_hip_batch_device_results) and queues the copy on
 * `stream` (NULL: the device's download stream) behind the batch's own kernels only, so the hits of
 * batch i cross PCIe while batch i + 1 executes; host_results waits for that copy and hands out the
 * arrays: hits [n_queries][*k_stride] (score desc, doc asc; counts[q] valid entries), counts
 * [n_queries], total_hits [n_queries].  They stay valid until the batch is destroyed, run again or
 * copied again. */
int irs_hip_batch_results_to_host(irs_hip_batch* batch, void* stream);
int irs_hip_batch_host_results(irs_hip_batch* batch, const irs_hip_hit** hits, uint32_t* k_stride,
                               const uint32_t** counts, const uint64_t** total_hits);

/* Tuning knobs (0 keeps the default). tile_docs in {4096, 6144, 8192, 12288}: docs
 * per LDS accumulator tile (default: the largest one that lets two workgroups share
 * a compute unit's LDS, 12288 with 32-bit and 6144 with 64-bit accumulators);
 * pilot_stride P: every P-th doc tile is scored first to bound the k-th score
 * (P == 1: exact two-pass); cand_cap: candidate slots per query.  Results do not
 * depend on any of them. */
int irs_hip_batch_configure(irs_hip_batch* batch, uint32_t tile_docs,
                            uint32_t pilot_stride, uint32_t cand_cap);

/* How the doc-tile units of a batch (Or / by_term / min-match, conjunctions that qualify)
 * execute — a tuning / test knob; the hits of plain disjunctions are BIT-IDENTICAL on every path
 * (units with match counts — conjunctions, min-match — differ by the rounding of their counting
 * accumulators when they join, within the parity tolerance):
 *   IRS_HIP_PATH_ITEMS   every query decodes the blocks of its own terms (work items);
 *   IRS_HIP_PATH_JOINED  every DISTINCT (segment, term) of the batch is decoded once per run
 *                        into streams of entries which the queries then only accumulate — what
 *                        block_disjunction::refill (disjunction.hpp:1240-1351) does per query,
 *                        shared by the queries of a batch (what AUTO takes when it joins).
 *                        For sum-merged units with table-family
 *                        scorers (BM25 / BM15 / TF-IDF over 1-byte norms or none), frequencies
 *                        < 256 — plain disjunctions, and conjunctions / min-match disjunctions of
 *                        at most 15 terms whose counting accumulators stay within the parity
 *                        tolerance; units that do not qualify run as ITEMS whatever was asked;
 *   IRS_HIP_PATH_AUTO    (default) by measured cost: plain disjunctions join when the batch's
 *                        streams are shared enough or its units many enough to pay for decoding
 *                        every distinct stream once (2.9 ps per distinct posting against 0.67 ps
 *                        per referenced posting + 2.4 ns per (unit, doc tile) saved —
 *                        tools/cost_sweep.py); a conjunction joins when walking every entry of
 *                        its terms beats decoding only the blocks its rarest term's docs fall
 *                        into.
 * Call before the batch's first run (or after a configure). */
enum { IRS_HIP_PATH_AUTO = 0, IRS_HIP_PATH_ITEMS = 1, IRS_HIP_PATH_JOINED = 2 };
int irs_hip_batch_set_path(irs_hip_batch* batch, int path);
/* Which one the batch's last run used (IRS_HIP_PATH_ITEMS / IRS_HIP_PATH_JOINED). */
int irs_hip_batch_path(irs_hip_batch* batch, int* path);

/* Paired doc tiles for the joined plain disjunctions (ABI 12; on by default) — a tuning / test
 * knob like set_path: a visit of k_join_score then covers two consecutive doc tiles whose sums
 * share an accumulator word (16 bits each, contributions rounded up), which only PICKS the docs;
 * their exact 32-bit sums are formed afterwards from the streams (k_join_rescore) with the
 * arithmetic of the unpaired kernel, so hits, scores, order and totals are BIT-IDENTICAL either
 * way.  enable = 1 (default): taken where it pays — the visits saved (doc tiles per unit) against
 * the look-ups added (about min(3 k / units sharing the threshold, k) docs per term): a 10 M-doc
 * segment at k = 1000 pairs, a lone 1.25 M-doc segment does not — unless a segment of the batch's
 * plain joined units has deleted documents; enable = 2: whatever the size (tests); enable = 0:
 * never — the units run on 32-bit tiles.
 * irs_hip_batch_paired_tiles: whether the last run took them. */
int irs_hip_batch_set_paired_tiles(irs_hip_batch* batch, int enable);
int irs_hip_batch_paired_tiles(irs_hip_batch* batch, int* used);

/* A batch over several segments (irs_hip_batch_create_multi) whose per-segment lists the caller
 * MERGES into one top k per query — what the harness does with its segments
 * (index-search.cpp:719-787) and what irs_hip_merge_topk does here: with `enable` the units of a
 * query share ONE score threshold, chosen so that the segments TOGETHER yield the k best docs
 * (plus the usual margin) instead of every segment its own k.  A segment's list then holds every
 * doc of it at or above the shared threshold — possibly fewer than k although more matched — and
 * the merged top k is exactly what it is without the option; total_hits are unchanged.  The
 * candidate volume, which is what small segments spend their time on, drops by the number of
 * segments.  Applies to units on joined posting streams whose scorers bound the score alike in
 * every segment (BM25 family); other units keep their own thresholds.  Off by default; call
 * before the batch's first run (or after a configure). */
int irs_hip_batch_set_shared_threshold(irs_hip_batch* batch, int enable);

/* ExecutionContext::wand (filter.hpp:52-78; utils/index-search --search-mode wand): lets the
 * batch SKIP posting blocks that cannot reach the top k.  The bound of a block is the query
 * term's own score function on the block's (max freq, min norm) — what the wanderator
 * evaluates per skip entry (formats_10.cpp:2498-2528, wand_writer.hpp:302-342) — summed over
 * the terms of a conjunction for the blocks overlapping one doc range (BlockConjunction,
 * conjunction.hpp:380-426); the threshold is the pilot pass's lower bound of the k-th score.
 * The top k (docs, scores, order) is the one the exhaustive run returns
 * (tests/search/wand_test.cpp:231-241); total_hits only counts the docs that were evaluated,
 * as with the reference's wand mode.  The per-block (max freq, min norm) are derived from the
 * postings on first use (every block gets them, also the last one of a list, for which the
 * skip data has no entry) and kept with the segment.  Applies to IRS_HIP_OP_AND queries on the
 * block-driven kernel and to whole doc tiles of OR queries on the work-item kernel; a unit the
 * cost rules put on joined posting streams (irs_hip_batch_set_path) is executed exhaustively —
 * its top k is the exhaustive one by construction, its total_hits the full count.  Call before
 * the batch's first run. */
int irs_hip_batch_set_wand(irs_hip_batch* batch, int enable);

/* irs::score::Min (score_function.hpp:42-142): the threshold the harness pushes into the
 * iterator once its heap is full — the k-th best score so far, carried over from the segments
 * it executed before (utils/index-search.cpp:737, 756, 777).  min_scores: [n_queries] floats
 * >= 0 (0 = none), or NULL to clear.  A doc scoring below min_scores[q] is not competitive: the
 * batch may drop it (pruning then starts from that bound instead of from the pilot pass's
 * estimate alone), so a query returns exactly the docs scoring at or above its threshold,
 * at most k of them, in the usual order — possibly fewer than k (the kernels drop by score bin,
 * the final selection by the score itself); total_hits still counts every match.  May be
 * called between runs. */
int irs_hip_batch_set_min_scores(irs_hip_batch* batch, const float* min_scores);
/* The block-max data of one term, for inspection/tests (computed as for set_wand): largest
 * frequency and smallest non-zero norm of every full 128-doc block. */
int irs_hip_term_blockmax(irs_hip_segment* seg, uint32_t term, uint32_t* max_freqs,
                          uint32_t* min_norms, uint32_t cap, uint32_t* count);
/* Where the block-max pairs of a segment come from: blocks whose pair was READ from the index's
 * own wand data — the payload of scorer 0 in the level-0 skip entries (FreqNormSource::Read,
 * wand_writer.hpp:318-334), present when the field was indexed with scorers
 * (irs_hip_segment_desc.wand_count > 0) — out of all full blocks; the others (every block of an
 * index written without wand data, the last block of every list, the norm where the payload
 * only carries a frequency) are derived from the postings.  A field with positions must have
 * been opened with its `.pos` for the entries to be read (they carry position fields). */
int irs_hip_segment_wand_source(irs_hip_segment* seg, uint64_t* from_index, uint64_t* total);

/* Kernel timing with HIP events recorded on the batch's own stream.
 * When enabled, every run() brackets each kernel launch with events;
 * timings() waits for the stream and returns the durations (ms) of the last run. */
enum {
  IRS_HIP_K_PLAN = 0,   /* block-range planning per (query, term) + work items; joined path:
                           the decode + norm join of the batch's distinct terms (k_join) */
  IRS_HIP_K_PILOT = 1,  /* pilot tiles -> per-query score threshold */
  IRS_HIP_K_SCORE = 2,  /* decode + score + accumulate + candidates (phrase batches: k_phrase) */
  IRS_HIP_K_SELECT = 3, /* exact top-k of the candidates            */
  IRS_HIP_K_COUNT = 4
};
/* enable: bit 0 = time the kernels (below); bit 1 = let the block-driven kernels count what
 * they decode (irs_hip_batch_touched) — a diagnostic run: the counting costs atomics. */
int irs_hip_batch_profile(irs_hip_batch* batch, int enable);
int irs_hip_batch_timings(irs_hip_batch* batch, float ms[IRS_HIP_K_COUNT]);
/* Work accounting for the roofline: algorithmic bytes A(q)
 * summed over the batch = posting bytes of every query term + 1 norm byte per
 * posting (norm_width) + 8*k result bytes; and the number of postings. */
int irs_hip_batch_work(irs_hip_batch* batch, uint64_t* algorithmic_bytes,
                       uint64_t* postings);
/* What the last run of a conjunction / phrase batch really read, next to the algorithmic
 * bytes above ("report both A(q) and bytes actually touched"): encoded bytes of
 * the `.doc` blocks it decoded plus the norm bytes it read, and the number of positions it
 * read from `.pos`.  (Doc-tile batches read every block of every term: A(q).)  Needs
 * irs_hip_batch_profile(batch, 2 | ...) before the run. */
int irs_hip_batch_touched(irs_hip_batch* batch, uint64_t* doc_bytes, uint64_t* positions);
/* How many times fetching results had to re-execute the batch so far: the pilot's
 * estimated threshold left fewer than k candidates for some query (re-run with the
 * provable threshold), or the candidate buffer overflowed (exact re-run, then a
 * larger buffer).  Results are exact either way; this only tells what it cost. */
int irs_hip_batch_reruns(irs_hip_batch* batch, uint32_t* count);

/* Multi-segment / multi-GPU merge: merges `n_lists` per-query
 * top-k lists (device pointers, each [n_queries][k] hits + [n_queries] counts,
 * list i belonging to segment ordinal seg_ids[i]) into the global top-k ordered
 * (score desc, segment asc, doc asc) — the order tests/search/wand_test.cpp:72-86
 * defines.  Outputs are device pointers: d_out [n_queries][k] hits,
 * d_out_seg [n_queries][k] uint32, d_out_counts [n_queries]. */
int irs_hip_merge_topk(int32_t device, const void* const* d_lists,
                       const void* const* d_counts, const uint32_t* seg_ids,
                       uint32_t n_lists, uint32_t n_queries, uint32_t k,
                       void* d_out, void* d_out_seg, void* d_out_counts,
                       void* stream);

/* ---- the collective of the multi-GPU exchange --------------------------
 * One process per GPU.  The per-segment top-k lists of a step — [n_queries][k] hits and
 * [n_queries] counts per local segment, written by irs_hip_batch_results_to_device straight
 * into one send buffer — are exchanged with ONE all-gather over RCCL (xGMI); every rank then
 * merges them with irs_hip_merge_topk.  What the harness loop `for (auto& segment : reader)`
 * into one heap (utils/index-search.cpp:719-779) becomes when the segments live on different
 * GPUs.  irs_hip_comm_unique_id: on one rank; the caller distributes the 128 bytes to the
 * others by whatever it has (MPI, a file, a socket), exactly as with ncclUniqueId. */
/* Device buffers for a host that drives the exchange without a HIP toolchain of its own (the
 * C++ layer iresearch_amd/cpp/irs_hip.hpp): plain hipMalloc / hipMemcpy / hipStreamSynchronize. */
int irs_hip_device_alloc(int32_t device, uint64_t bytes, void** d_out);
void irs_hip_device_free(int32_t device, void* d_ptr);
int irs_hip_device_upload(int32_t device, void* d_dst, const void* h_src, uint64_t bytes);
int irs_hip_device_download(int32_t device, void* h_dst, const void* d_src, uint64_t bytes);
int irs_hip_device_sync(int32_t device, void* stream);
/* The library recycles the device and page-locked memory of destroyed batches (hipMalloc /
 * hipFree / hipHostMalloc cost more than a batch's kernels, and hipFree synchronises the device):
 * up to 64 GB of device memory and 4 GB of page-locked host memory per device stay with the
 * library (IRS_HIP_POOL_MB / IRS_HIP_PINNED_POOL_MB override); a closed segment's memory is freed
 * at once.  This hands all of it back to the runtime — postings_reader::CountMappedMemory's
 * counterpart for callers that watch their memory (formats.hpp:190). */
int irs_hip_device_trim(int32_t device);

typedef struct irs_hip_comm irs_hip_comm;
#define IRS_HIP_COMM_ID_BYTES 128u
int irs_hip_comm_unique_id(uint8_t id[IRS_HIP_COMM_ID_BYTES]);
int irs_hip_comm_init_rank(int32_t device, const uint8_t id[IRS_HIP_COMM_ID_BYTES], int32_t n_ranks,
                           int32_t rank, irs_hip_comm** out);
void irs_hip_comm_destroy(irs_hip_comm* comm);
/* Which RCCL the collective runs on (diagnostics): the library's path, prefixed "mapped:" when
 * the process had it loaded already — a host that also uses torch.distributed has torch's
 * bundled librccl.so mapped, and the communicator then binds THAT copy instead of loading a
 * second RCCL next to it.  EHIP when no RCCL could be bound. */
int irs_hip_comm_library(char* buf, size_t cap);
/* d_send: bytes_per_rank bytes on the device; d_recv: n_ranks * bytes_per_rank, rank r's block
 * at r * bytes_per_rank.  Asynchronous on `stream` (a hipStream_t). */
int irs_hip_topk_allgather(irs_hip_comm* comm, const void* d_send, void* d_recv,
                           uint64_t bytes_per_rank, void* stream);

/* ONE threshold per query across RANKS — the harness keeps one heap over all segments of the
 * index (utils/index-search.cpp:719-779); with the segments sharded over processes that is one
 * threshold for a query's units on every rank.  With a communicator attached, every run of the
 * batch sums the pilot histograms of each query over the ranks (one all-reduce of
 * n_queries * 514 counters between the pilot and the scoring kernels) and picks the threshold
 * from the sum, so that all segments of the index TOGETHER yield the k best docs (plus the usual
 * margin); a second, small all-reduce behind the selection sums what each group listed and
 * matched, so that every rank reaches the same verdict on the estimate and re-runs (or not) in
 * step with the others.  The merged top k over all ranks' lists is exactly what it is without
 * the option; a rank's list holds its docs at or above the shared threshold, possibly fewer
 * than k.  Requirements: the SAME queries in the same order on every rank (statistics are
 * index-global anyway), every rank runs its batches in the same order, and nothing else uses
 * `comm` concurrently — give the top-k exchange its own communicator when it overlaps the next
 * batch.  Applies to units on joined posting streams scored by the BM25 family (their score
 * bound is the same on every segment); other units keep thresholds of their own while the rank
 * still takes part in the collectives.  Implies irs_hip_batch_set_shared_threshold for the
 * rank's own segments.  NULL detaches.  Call before the batch's first run; `comm` must outlive
 * the batch. */
int irs_hip_batch_set_comm(irs_hip_batch* batch, irs_hip_comm* comm);

const char* irs_hip_strerror(int status);
uint32_t irs_hip_abi_version(void);
/* Name of the device the library would use, e.g. "gfx950"; EHIP when none. */
int irs_hip_device_arch(int32_t device, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* IRS_HIP_H */
