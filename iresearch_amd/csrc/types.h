// types.h — device-side data layout of a staged segment and of a query batch.
// See DESIGN.md "Data layout in HBM".
#pragma once
#include <cstdint>

namespace irs_hip {

constexpr uint32_t kBlock = 128;       // postings per block (formats_10.cpp:90)
constexpr uint32_t kDocMin = 1;        // doc_limits::min() (type_limits.hpp:45)
constexpr uint32_t kMaxTerms = 16;     // IRS_HIP_MAX_TERMS
constexpr uint32_t kMaxK = 4096;       // IRS_HIP_MAX_K
constexpr uint32_t kBins = 512;        // score histogram bins (pilot threshold)
constexpr uint32_t kMaxCaches = 4;     // distinct (norm_const, norm_length) per query in LDS
constexpr uint32_t kPadBytes = 64;      // zero padding after the staged `.doc` / `.pos` bytes
constexpr uint64_t kNoPlan = ~uint64_t(0);  // DevQuery::first_off of a unit without plan tables

enum Layout : int32_t { kScalar = 0, kSimd4 = 1 };

// Scorer kinds after resolving the norm source against the segment
// (bm25.cpp:447-489, tfidf.cpp:307-352).
enum Kind : int32_t {
  kBM1 = 0,
  kBM15 = 1,
  kBM25Tiny = 2,   // Norm2 1 byte: c0 - c0/(1 + tf*norm_cache[norm])   bm25.cpp:348-353
  kBM25Wide = 3,   // Norm2 2/4 bytes: c0 - c0*c1/(c1 + tf)              bm25.cpp:355-359
  kBM25One = 4,    // no norm column: norm == 1 through the tiny path   bm25.cpp:487-489
  kTfidf = 5,
  kTfidfTiny = 6,  // * kRSQRT.get<false>(norm)
  kTfidfWide = 7,  // * kRSQRT.get<true>(norm)
  // legacy `Norm` feature (norm.hpp:46-70): the column holds 1/sqrt(|doc|) as float
  kBM25Legacy = 8,   // tf = sqrt(freq), norm = 1/stored: c0 - c0*c1/(c1 + tf)   bm25.cpp:333-359, 242-249
  kTfidfLegacy = 9,  // sqrt(freq) * idf * stored                              tfidf.cpp:214-219, 253
};

// One term of the segment's term table, as staged on the device.
struct DevTerm {
  uint64_t doc_start;   // absolute offset of the term's postings in the staged file
  uint64_t dir_off;     // index of the term's first entry in the block directory
  uint64_t tail_off;    // [dir kernel] absolute offset of the vint tail
  uint32_t docs_count;
  uint32_t nblk;        // full 128-doc blocks
  uint32_t tail_n;      // docs_count % 128 (0 for a single-doc term)
  uint32_t tail_base;   // [dir kernel] last doc of the last full block (1 if none)
  uint32_t tail_bytes;  // [dir kernel] encoded length of the tail
  uint32_t blocks_bytes;// [dir kernel] encoded length of all full blocks
  uint32_t single_doc;  // docs_count == 1: the doc id (1 + e_single_doc), else 0
  uint32_t single_freq; // docs_count == 1: term_meta::freq
  uint32_t tf_bound;    // [dir kernel] upper bound of tf over the whole list
  uint32_t last_doc;    // [dir kernel] last doc id of the list
  uint32_t tail_row;    // first entry of the term's decoded tail / single doc in tail_docs / tail_freqs
  uint32_t pad;
};

// Where the positions of one term live in the staged `.pos` file (fields with POS).
struct DevPosTerm {
  uint64_t pos_start;   // term_meta::pos_start: absolute offset of the term's first pos block
  uint64_t row;         // index of the term's first entry in the pos block directory
  uint32_t nfull;       // full 128-position blocks: term_meta::freq / 128
  uint32_t tail_n;      // vint-coded positions behind them: freq % 128
  uint32_t total;       // term_meta::freq
  uint32_t bytes;       // [pos dir kernel] encoded length of the term's positions
  uint32_t tail_row;    // first entry of the term's decoded position tail in DevSegment::ptail
  uint32_t pad;
};

// One entry of the block directory as the work-item builder reads it: everything about a
// block in ONE 16-byte load (the separate arrays below stay for the binary searches and the
// position kernels).
struct alignas(16) BlkDir {
  uint32_t off;        // byte offset of the block relative to the term's doc_start
  uint32_t prev_last;  // last doc of the preceding block of the term (kDocMin for its first)
  uint32_t aoff;       // offset of the block in the packed-payload image, 16-byte units
  uint32_t bits;       // doc bits | freq bits << 8
};

struct DevSegment {
  const uint8_t* doc;        // staged `.doc` bytes (+ kPadBytes zeros)
  uint64_t doc_len;
  const uint8_t* norms;      // Norm2 column bytes, may be null
  uint32_t norm_width;
  uint32_t norm_min_doc;
  uint64_t norm_count;      // (norm_legacy below: the values are little-endian floats, width 4)
  const DevTerm* terms;
  uint32_t num_terms;
  uint32_t num_docs;
  // block directory, one entry per full block of every term
  const uint32_t* blk_off;   // byte offset relative to the term's doc_start
  const uint32_t* blk_last;  // absolute last doc id of the block
  const uint16_t* blk_bits;  // doc bits | freq bits << 8 (0 = all-equal block)
  // packed-payload image: the doc and freq payloads of every block whose two parts
  // are 1..31-bit packed, headers dropped, back to back — every payload starts on a
  // 16-byte boundary (a payload is 16*bits bytes), unlike in `.doc` where the 1-byte
  // headers leave all of them misaligned.  The hot decoder reads this copy.
  const uint8_t* pk;
  const uint32_t* blk_aoff;  // offset of the block in `pk`, in 16-byte units
  const BlkDir* blk_dir;     // the same facts per block, gathered
  const uint32_t* blk_term;  // ... and the term whose list the block belongs to: the per-block
                             // passes over the whole directory split their work by ROW
  // block-max data (null until a WAND batch asks for it): largest frequency and smallest
  // non-zero norm of every full block — what FreqNormProducer stores per skip entry
  // (wand_writer.hpp:170-209)
  const uint32_t* blk_maxf;
  const uint32_t* blk_minn;
  // decoded vint tails / single docs, term after term (DevTerm::tail_row): absolute doc ids
  // and frequencies — at most 127 entries per term, 1 for a single-doc term
  const uint32_t* tail_docs;
  const uint32_t* tail_freqs;
  int32_t has_freq;
  int32_t layout;
  uint32_t wand_count;       // scorers the field was indexed with (wand data in front of short tails)
  // positions (null unless the field has IndexFeatures::POS and `.pos` was staged)
  const uint8_t* pos;        // staged `.pos` bytes (+ kPadBytes zeros)
  uint64_t pos_len;
  const DevPosTerm* pterms;  // [num_terms]
  const uint32_t* pblk_off;  // pos block directory: byte offset relative to pos_start
  const uint8_t* pblk_bits;  //   and bit width (0 = all-equal block)
  const uint32_t* blk_pos;   // per doc-block row (+1 sentinel): positions of ALL earlier rows
                             // (exclusive scan of the blocks' frequency sums, mod 2^32)
  const uint32_t* ptail;     // decoded position-delta tails, term after term (DevPosTerm::tail_row)
  uint32_t pos_base;         // what a doc's first delta is relative to: 0 (formats 1_3+, zero-based
                             // storage) or pos_limits::min() = 1 (1_0..1_2, formats_10.cpp:1623-1625)
  uint32_t norm_legacy;      // the norm column is the legacy `Norm` feature: float 1/sqrt(|doc|)
  // the 1-byte Norm2 column once more, in POSTING order (null until a batch asks for it:
  // prepare_posting_norms): 128 bytes per directory row, and per entry of tail_docs
  const uint8_t* pnorm;
  const uint8_t* tail_norms;
  // the segment's DocumentMask as a bitmap (null: no deleted docs): bit (doc - kDocMin) set = the
  // doc is deleted — what SegmentReaderImpl::mask wraps every iterator with
  // (core/index/segment_reader_impl.cpp:69-101, 286); padded so that the bits of a whole last doc
  // tile are readable
  const uint32_t* dead;
};

// MaskDocIterator::next (segment_reader_impl.cpp:74-82): `!mask_.contains(value())`
__device__ __forceinline__ bool doc_dead(const uint32_t* dead, uint32_t doc) {
  const uint32_t j = doc - kDocMin;
  return (dead[j >> 5] >> (j & 31u)) & 1u;
}

struct DevQuery {
  int32_t op;
  uint32_t n_terms;
  uint32_t first_term;  // into the DevQTerm array
  uint32_t k;
  float bin_scale;      // kBins / (upper bound U of the query's score)
  uint32_t n_caches;
  float fx_mul;         // 2^(E-32): float score -> high word of the fixed-point value
  float fx_inv;         // 2^-E, E = 61 - ceil(log2 U): fixed-point sum -> float
  // A batch may span several segments: a (segment, query) pair is one execution unit.
  uint32_t seg;         // index into the batch's DevSegment array
  uint32_t n_tiles;     // doc tiles of that segment
  uint64_t first_off;   // start of the unit's [n_tiles + 1][jt] slice of the plan table
                        // (kNoPlan: the unit runs on joined posting streams, join.h)
  // work-queue order of k_score (NOT a property of this unit): slot i of the array names the
  // unit that runs i-th within every chunk round — units sorted by decreasing work, so the
  // last workgroups to finish hold the lightest chunks (longest-processing-time first)
  uint32_t run_unit;
  uint32_t tile_base;   // index of the unit's first doc tile in the batch-wide per-tile tables
                        // (DevQuery n_tiles summed over the units in front of it)
  uint32_t pad_q[2];
};

struct DevQTerm {
  uint32_t term;        // ordinal in the term table, 0xFFFFFFFF = absent
  int32_t kind;         // Kind
  float c0;
  float norm_const;
  float norm_length;
  uint32_t cache_id;    // < kMaxCaches: norm_cache slot in LDS; else compute on the fly
  uint32_t pad0;        // phrase queries: the term's offset in the phrase
  uint32_t pad1;        // the term's largest frequency (DevTerm::tf_bound): below the rows of a
                        // table slot, every posting of the term scores through its table row
};

// What a tile workgroup needs to know about one term of one query, gathered by the plan
// kernel into ONE record (no dependent chain of loads in the chunk prologue).  The
// postings of the term's vint tail (or its single doc) — at most 127 of them — are decoded
// once per TERM when the segment is opened (DevSegment::tail_docs / tail_freqs).
struct DevTail {
  uint32_t n;           // postings in the decoded tail (1 for a single-doc term)
  uint32_t first_doc;
  uint32_t last_doc;
  uint32_t nblk;        // copy of DevTerm::nblk
  uint64_t doc_start;   // copy of DevTerm::doc_start
  uint64_t dir_off;     // copy of DevTerm::dir_off
  uint32_t term;        // ordinal in the term table
  uint32_t tail_row;    // copy of DevTerm::tail_row
};

struct Hit {
  float score;
  uint32_t doc;
};

// candidate key: descending key order == (score desc, doc asc); scores are > 0
__host__ __device__ inline uint64_t make_key(float score, uint32_t doc) {
  uint32_t bits;
  __builtin_memcpy(&bits, &score, 4);
  return (uint64_t(bits) << 32) | uint64_t(0xFFFFFFFFu - doc);
}
__host__ __device__ inline Hit key_hit(uint64_t key) {
  Hit h;
  const uint32_t bits = uint32_t(key >> 32);
  __builtin_memcpy(&h.score, &bits, 4);
  h.doc = 0xFFFFFFFFu - uint32_t(key);
  return h;
}

}  // namespace irs_hip
