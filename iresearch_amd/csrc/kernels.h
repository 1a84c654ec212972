// kernels.h — the CDNA4 kernels of the query path.  No MFMA on purpose:
// nothing here is a dense contraction; the work is byte/integer streaming
// bounded by HBM bandwidth (DESIGN.md "Kernels").
//
//   k_build_directory  segment open: walk block headers, record per-block
//                      offset / last doc / bit widths (no decoded data kept)
//   k_decode_term      bulk decode of one posting list (bit-exact test surface)
//   k_bit_union        doc bitset of many posting lists (postings_reader::bit_union)
//   k_plan             per (query, term): first block of every doc tile + tail decode
//   k_pilot            score every P-th doc tile, derive a per-query score-bin
//                      threshold that provably keeps the top-k
//   k_score            decode + score + accumulate every doc tile in LDS,
//                      emit candidates above the threshold, count hits
//   k_select           exact top-k (score desc, doc asc) of the candidates
//   k_merge_topk       multi-segment merge (score desc, segment asc, doc asc)
#pragma once
#include "decode.h"
#include "types.h"
#include "wave.h"

namespace irs_hip {

constexpr uint32_t kNoTerm = 0xFFFFFFFFu;
constexpr uint32_t kThreads = 256;      // 4 wavefronts per workgroup (utility kernels)
constexpr uint32_t kTileThreadsMax = 1024;  // pilot/score workgroups: 256..1024 threads
constexpr uint32_t kWaves = kThreads / 64;
constexpr uint32_t kLocalCands = 256;   // per-tile candidate staging slots in LDS
#ifndef IRS_ITEM_CHUNK
#define IRS_ITEM_CHUNK 256
#endif
constexpr uint32_t kItemChunk = IRS_ITEM_CHUNK;    // (term, block) work items staged in LDS at a time

enum : uint32_t {
  kStatusCorrupt = 1u,   // malformed block header / out-of-bounds offset
  kStatusOverflow = 2u,  // candidate buffer exhausted
  kStatusUnderflow = 4u, // an estimated threshold left fewer than k candidates
};

// ------------------------------------------------------------- directory --

// 16-byte units a block occupies in the packed-payload image (DevSegment::pk):
// only blocks the straight-line decoder handles (both parts 1..31-bit packed).
__device__ __forceinline__ uint32_t pk_units(uint32_t dbits, uint32_t fbits) {
  return ((dbits - 1u) <= 30u && (fbits - 1u) <= 30u) ? dbits + fbits : 0u;
}

// One wavefront per term walks the term's full blocks front to back: header
// byte -> payload size (bitpack::skip_block32, bitpack.hpp:60-69); the block's
// last doc is base + sum(deltas).  Lane 0 then decodes the vint tail
// (formats_10.cpp:1765-1792) into the per-term tail tables.  The stream is staged through
// LDS (kDirWindow bytes per refill and wavefront): the walk is a chain of dependent reads.
constexpr uint32_t kDirWindow = 8192;
struct alignas(16) DirLine {
  uint64_t lo, hi;
};

template<int LAYOUT>
__global__ void __launch_bounds__(kThreads)
k_build_directory(DevSegment seg, DevTerm* terms, uint32_t* blk_off,
                  uint32_t* blk_last, uint16_t* blk_bits, uint32_t* blk_units,
                  uint32_t* tail_docs, uint32_t* tail_freqs, uint32_t* status) {
  __shared__ __attribute__((aligned(16))) uint8_t s_win[kWaves][kDirWindow];
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t term = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (term >= seg.num_terms) return;
  DevTerm t = terms[term];
  if (t.docs_count == 0) return;
  if (t.docs_count == 1) {  // single_doc_iterator, formats_10.cpp:1876-1890
    if (lane == 0) {
      tail_docs[t.tail_row] = t.single_doc;
      tail_freqs[t.tail_row] = t.single_freq;
      terms[term].last_doc = t.single_doc;
      terms[term].tf_bound = t.single_freq;
      terms[term].tail_off = t.doc_start;
      terms[term].tail_base = kDocMin;
    }
    return;
  }
  uint64_t cur = t.doc_start;
  uint32_t base = kDocMin;  // formats_10.cpp:636 / :2100-2105
  uint32_t tfb = 0;
  bool bad = false;
  // The walk is a chain of dependent reads (header -> payload -> next header), so the
  // stream is staged through LDS, kDirWindow bytes at a time: all 64 lanes copy, then the
  // blocks inside the window are parsed at LDS latency.
  uint8_t* win = s_win[threadIdx.x >> 6];
  const uint64_t staged = seg.doc_len + kPadBytes;  // the device copy ends with zero padding
  uint64_t win_lo = 0, win_hi = 0;
  constexpr uint32_t kMaxPair = 2u * (1u + 16u * 32u) + 16u;  // doc block + freq block
  for (uint32_t b = 0; b < t.nblk; ++b) {
    if (cur + 2 > seg.doc_len) { bad = true; break; }
    if (cur + kMaxPair > win_hi) {
      wave::sync();  // every lane is done with the old window
      win_lo = cur & ~uint64_t(15);
      uint64_t bytes = staged - win_lo;
      if (bytes > kDirWindow) bytes = kDirWindow;
      bytes &= ~uint64_t(15);
      for (uint32_t o = lane * 16u; o < bytes; o += 64u * 16u)
        *reinterpret_cast<DirLine*>(win + o) = *reinterpret_cast<const DirLine*>(seg.doc + win_lo + o);
      win_hi = win_lo + bytes;
      wave::sync();
    }
    const uint8_t* blk = win + (cur - win_lo);
    const uint32_t dbits = blk[0];
    if (dbits > 32 || cur + 1 + 16ull * dbits > seg.doc_len) { bad = true; break; }
    uint32_t x0, x1;
    uint32_t size = read_block_pair<LAYOUT>(blk, dbits, lane, x0, x1);
    const uint32_t last = base + wave::reduce_add(x0 + x1);
    uint32_t fbits = 0;
    if (seg.has_freq) {
      if (cur + size + 2 > seg.doc_len) { bad = true; break; }
      const uint8_t* fb = blk + size;
      fbits = fb[0];
      if (fbits > 32 || cur + size + 1 + 16ull * fbits > seg.doc_len) { bad = true; break; }
      if (fbits) {
        size += 1u + 16u * fbits;
        const uint32_t bound = fbits >= 32 ? 0xFFFFFFFFu : ((1u << fbits) - 1u);
        tfb = bound > tfb ? bound : tfb;
      } else {
        uint32_t len;
        const uint32_t v = vint_from(wave::load_u64(fb + 1), &len);
        size += 1u + len;
        tfb = v > tfb ? v : tfb;
      }
    }
    if (lane == 0) {
      blk_off[t.dir_off + b] = uint32_t(cur - t.doc_start);
      blk_last[t.dir_off + b] = last;
      blk_bits[t.dir_off + b] = uint16_t(dbits | (fbits << 8));
      blk_units[t.dir_off + b] = pk_units(dbits, fbits);
    }
    cur += size;
    base = last;
  }
  if (lane == 0) {
    // a list without a skip list carries its wand root in front of the tail
    // (formats_10.cpp:686-688): one size byte per scorer, then the payloads
    // (CommonSkipWandData :1962-1979).  A 128-doc list has it behind its only block,
    // where nothing is read any more.
    if (!bad && t.nblk == 0 && seg.wand_count) {
      uint64_t skip = 0;
      if (cur + seg.wand_count > seg.doc_len) {
        bad = true;
      } else {
        for (uint32_t w = 0; w < seg.wand_count; ++w) skip += seg.doc[cur + w];
        cur += seg.wand_count + skip;
        if (cur > seg.doc_len) bad = true;
      }
    }
    const uint64_t tail_off = cur;
    uint32_t doc = base;
    if (!bad) {
      for (uint32_t i = 0; i < t.tail_n; ++i) {
        if (cur + 1 > seg.doc_len) { bad = true; break; }
        uint32_t len;
        const uint32_t v = vint_from(wave::load_u64(seg.doc + cur), &len);
        cur += len;
        uint32_t f = 1;
        if (seg.has_freq) {
          doc += v >> 1;  // shift_unpack_32, store_utils.hpp:266-269
          if (v & 1u) {
            tfb = tfb ? tfb : 1u;
          } else {
            f = vint_from(wave::load_u64(seg.doc + cur), &len);
            cur += len;
            tfb = f > tfb ? f : tfb;
          }
        } else {
          doc += v;
        }
        // the decoded tail (read_tail_block, formats_10.cpp:1765-1792) is kept per term
        tail_docs[t.tail_row + i] = doc;
        tail_freqs[t.tail_row + i] = f;
      }
      if (cur > seg.doc_len) bad = true;
    }
    terms[term].tail_off = tail_off;
    terms[term].tail_base = base;
    terms[term].tail_bytes = uint32_t(cur - tail_off);
    terms[term].blocks_bytes = t.nblk ? uint32_t(tail_off - t.doc_start) : 0u;  // not the wand root
    terms[term].tf_bound = seg.has_freq ? tfb : 1u;
    terms[term].last_doc = doc;
    if (bad) atomicOr(status, kStatusCorrupt);
  }
}

// Exclusive prefix sum of n u32 values in place (block sizes -> block offsets),
// three launches: per-chunk totals, scan of the totals (one workgroup), apply.
constexpr uint32_t kScanChunk = kThreads * 8;

__global__ void __launch_bounds__(kThreads)
k_scan_totals(const uint32_t* v, uint64_t n, uint64_t* totals) {
  __shared__ uint32_t wsum[kWaves];
  const uint64_t base = uint64_t(blockIdx.x) * kScanChunk;
  uint32_t s = 0;
  for (uint32_t i = threadIdx.x; i < kScanChunk; i += blockDim.x)
    if (base + i < n) s += v[base + i];
  s = wave::reduce_add(s);
  if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t t = 0;
    for (uint32_t w = 0; w < kWaves; ++w) t += wsum[w];
    totals[blockIdx.x] = t;
  }
}

// totals[i] -> sum of totals[0..i), totals[n_parts] = grand total (one wavefront)
__global__ void __launch_bounds__(64)
k_scan_parts(uint64_t* totals, uint32_t n_parts) {
  const unsigned lane = threadIdx.x;
  uint64_t carry = 0;
  for (uint32_t i0 = 0; i0 < n_parts; i0 += 64) {
    const uint32_t i = i0 + lane;
    const uint64_t x = i < n_parts ? totals[i] : 0;
    // chunk totals are < 2^32 * kScanChunk: scan the halves separately
    const uint32_t lo = wave::inclusive_scan(uint32_t(x & 0xFFFFu));
    const uint32_t mid = wave::inclusive_scan(uint32_t((x >> 16) & 0xFFFFu));
    const uint32_t hi = wave::inclusive_scan(uint32_t(x >> 32));
    const uint64_t incl = uint64_t(lo) + (uint64_t(mid) << 16) + (uint64_t(hi) << 32);
    if (i < n_parts) totals[i] = carry + incl - x;
    const uint32_t l2 = wave::bcast(lo, 63), m2 = wave::bcast(mid, 63), h2 = wave::bcast(hi, 63);
    carry += uint64_t(l2) + (uint64_t(m2) << 16) + (uint64_t(h2) << 32);
  }
  if (lane == 0) totals[n_parts] = carry;
}

__global__ void __launch_bounds__(kThreads)
k_scan_apply(uint32_t* v, uint64_t n, const uint64_t* totals) {
  __shared__ uint32_t wsum[kWaves];
  const uint64_t base = uint64_t(blockIdx.x) * kScanChunk;
  const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  // thread t owns 8 consecutive values
  const uint64_t at = base + uint64_t(threadIdx.x) * 8u;
  uint32_t x[8], s = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    x[e] = at + e < n ? v[at + e] : 0u;
    s += x[e];
  }
  const uint32_t incl = wave::inclusive_scan(s);
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  uint64_t off = totals[blockIdx.x] + (incl - s);
  for (uint32_t w = 0; w < wv; ++w) off += wsum[w];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (at + e < n) v[at + e] = uint32_t(off);   // the host checked the grand total fits
    off += x[e];
  }
}

// Copies the payloads of the decodable blocks into the packed-payload image.
// grid = num_terms * slices, as k_bit_union.
__global__ void __launch_bounds__(kThreads)
k_pack_payloads(DevSegment seg, uint32_t slices, uint8_t* pk) {
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t slice = blockIdx.x % slices;
  const DevTerm t = seg.terms[blockIdx.x / slices];
  if (t.docs_count < 2) return;
  for (uint32_t b = slice * kWaves + (threadIdx.x >> 6); b < t.nblk; b += slices * kWaves) {
    const uint64_t e = t.dir_off + b;
    const uint32_t bits = seg.blk_bits[e];
    const uint32_t dbits = bits & 0xFFu, fbits = bits >> 8;
    if (!pk_units(dbits, fbits)) continue;
    const uint8_t* blk = seg.doc + t.doc_start + seg.blk_off[e];
    uint64_t* dst = reinterpret_cast<uint64_t*>(pk + (uint64_t(seg.blk_aoff[e]) << 4));
    // 8-byte pieces: 2*dbits of the doc payload (after its header byte), then
    // 2*fbits of the freq payload (after the second header byte)
    for (uint32_t i = lane; i < 2u * (dbits + fbits); i += 64) {
      const uint8_t* src = i < 2u * dbits ? blk + 1u + 8u * i : blk + 2u + 8u * i;
      dst[i] = wave::load_u64(src);
    }
  }
}

// LEB128 read one byte at a time (safe for LDS and for unaligned global bytes).
__device__ __forceinline__ uint32_t vint_bytes(const uint8_t* p, uint32_t* len) {
  uint32_t v = 0, n = 0, shift = 0;
  for (;;) {
    const uint32_t b = p[n++];
    v |= (b & 0x7Fu) << shift;
    if (!(b & 0x80u) || n == 5) break;
    shift += 7;
  }
  *len = n;
  return v;
}

// One-lane sequential decode of a vint tail
// (doc_iterator_base::read_tail_block, formats_10.cpp:1765-1792).
__device__ __forceinline__ void decode_tail_serial(const uint8_t* p, uint32_t n,
                                                   uint32_t base, bool has_freq,
                                                   uint32_t* docs, uint32_t* freqs,
                                                   uint32_t* last_out) {
  uint32_t doc = base;
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t len;
    const uint32_t v = vint_bytes(p, &len);
    p += len;
    uint32_t f = 1;
    if (has_freq) {
      doc += v >> 1;  // shift_unpack_32, store_utils.hpp:266-269
      if (!(v & 1u)) {
        f = vint_bytes(p, &len);
        p += len;
      }
    } else {
      doc += v;
    }
    docs[i] = doc;
    if (freqs) freqs[i] = f;
  }
  *last_out = doc;
}

// ----------------------------------------------------------- bulk decode --

// grid.x = nblk + 1 wave-sized work items of ONE term, kWaves per workgroup.
template<int LAYOUT>
__global__ void __launch_bounds__(kThreads)
k_decode_term(DevSegment seg, uint32_t term, uint32_t* out_docs,
              uint32_t* out_freqs) {
  const unsigned lane = threadIdx.x & 63u;
  const DevTerm t = seg.terms[term];
  const uint32_t item = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (t.docs_count == 1) {
    if (item == 0 && lane == 0) {
      out_docs[0] = t.single_doc;
      if (out_freqs) out_freqs[0] = t.single_freq;
    }
    return;
  }
  if (item < t.nblk) {
    const uint64_t e = t.dir_off + item;
    const uint32_t bits = seg.blk_bits[e];
    const uint32_t base = item ? seg.blk_last[e - 1] : kDocMin;
    uint32_t d0, d1, f0, f1;
    const uint8_t* blk = seg.doc + t.doc_start + seg.blk_off[e];
    if (seg.has_freq) {
      decode_block<LAYOUT, true>(blk, bits & 0xFFu, bits >> 8, base, lane, d0, d1, f0, f1);
    } else {
      decode_block<LAYOUT, false>(blk, bits & 0xFFu, 0, base, lane, d0, d1, f0, f1);
    }
    const uint32_t o = item * kBlock + 2u * lane;
    out_docs[o] = d0;
    out_docs[o + 1] = d1;
    if (out_freqs) {
      out_freqs[o] = f0;
      out_freqs[o + 1] = f1;
    }
  } else if (item == t.nblk && lane == 0 && t.tail_n) {
    uint32_t last;
    decode_tail_serial(seg.doc + t.tail_off, t.tail_n, t.tail_base, seg.has_freq != 0,
                       out_docs + t.nblk * kBlock,
                       out_freqs ? out_freqs + t.nblk * kBlock : nullptr, &last);
  }
}

// ------------------------------------------------------------- bit union --

// postings_reader::bit_union (formats_10.cpp:3716-3806): set bit `doc` of a doc
// bitset for every posting of every given term; freq blocks are never touched
// (the directory knows where each doc block starts).  Work is cut by BLOCKS, not by
// terms (posting lists are Zipf-distributed: the longest list of a prefix expansion can
// hold most of the postings): one workgroup = up to kUnionBlocks consecutive blocks of
// one term, a wavefront per block; the workgroup that owns a term's last blocks also
// takes its decoded vint tail / single doc.
constexpr uint32_t kUnionBlocks = 64;
struct UnionWg {
  uint32_t term;
  uint32_t first_block;
};

template<int LAYOUT>
__global__ void __launch_bounds__(kThreads)
k_bit_union(DevSegment seg, const UnionWg* wgs, uint32_t* set32, uint64_t n_bits) {
  const unsigned lane = threadIdx.x & 63u;
  const UnionWg wg = wgs[blockIdx.x];
  const DevTerm t = seg.terms[wg.term];
  auto mark = [&](uint32_t doc) {
    if (doc < n_bits) atomicOr(&set32[doc >> 5], 1u << (doc & 31u));
  };
  uint32_t end = wg.first_block + kUnionBlocks;
  if (end > t.nblk) end = t.nblk;
  for (uint32_t b = wg.first_block + (threadIdx.x >> 6); b < end; b += kWaves) {
    const uint64_t e = t.dir_off + b;
    const uint32_t base = b ? seg.blk_last[e - 1] : kDocMin;
    uint32_t d0, d1, f0, f1;
    decode_block<LAYOUT, false>(seg.doc + t.doc_start + seg.blk_off[e], seg.blk_bits[e] & 0xFFu,
                                0, base, lane, d0, d1, f0, f1);
    mark(d0);
    mark(d1);
  }
  // the decoded tail (or the single doc, formats_10.cpp:3797-3801): at most 127 docs
  if (end == t.nblk && threadIdx.x < kBlock) {
    const uint32_t n = t.docs_count == 1 ? 1u : t.tail_n;
    if (threadIdx.x < n) mark(seg.tail_docs[t.tail_row + threadIdx.x]);
  }
}

// ------------------------------------------------------------------ plan --

// One workgroup per (query, term slot).  Threads binary-search the block
// directory for the first block whose last doc reaches each tile's first doc
// (what SkipReader::Seek does per iterator, skip_list.hpp:208-249); thread 0
// decodes the term's vint tail into the batch scratch.
__global__ void __launch_bounds__(kThreads)
k_plan(const DevSegment* segs, const DevQuery* queries, const DevQTerm* qterms,
       uint32_t jt /*term slots per query*/, uint32_t tile_docs,
       uint32_t* first /*per unit: [n_tiles+1][jt]*/, DevTail* tails /*[unit][jt]*/) {
  const uint32_t q = blockIdx.x / jt, j = blockIdx.x % jt;   // q: (segment, query) unit
  const DevQuery qd = queries[q];
  const DevSegment seg = segs[qd.seg];
  const uint32_t n_tiles = qd.n_tiles;
  DevTail* tl = tails + (uint64_t(q) * jt + j);
  // table layout [unit][tile][term slot]: one tile's entries for all terms are adjacent
  uint32_t* col = first + qd.first_off + j;
  if (j >= qd.n_terms || qterms[qd.first_term + j].term == kNoTerm) {
    for (uint32_t tile = threadIdx.x; tile <= n_tiles; tile += blockDim.x)
      col[uint64_t(tile) * jt] = 0;
    if (threadIdx.x == 0) {
      tl->n = 0; tl->first_doc = 0; tl->last_doc = 0;
      tl->nblk = 0; tl->doc_start = 0; tl->dir_off = 0;
      tl->term = 0; tl->tail_row = 0;
    }
    return;
  }
  const DevTerm t = seg.terms[qterms[qd.first_term + j].term];
  const uint32_t* last = seg.blk_last + t.dir_off;
  for (uint32_t tile = threadIdx.x; tile <= n_tiles; tile += blockDim.x) {
    const uint64_t lo64 = uint64_t(kDocMin) + uint64_t(tile) * tile_docs;
    const uint32_t lo = lo64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : uint32_t(lo64);
    uint32_t a = 0, b = t.nblk;  // lower_bound(last, lo)
    while (a < b) {
      const uint32_t m = (a + b) >> 1;
      if (last[m] < lo) a = m + 1; else b = m;
    }
    col[uint64_t(tile) * jt] = a;
  }
  if (threadIdx.x == 0) {
    const uint32_t term = qterms[qd.first_term + j].term;
    tl->nblk = t.nblk;
    tl->doc_start = t.doc_start;
    tl->dir_off = t.dir_off;
    tl->term = term;
    tl->tail_row = t.tail_row;
    // the tail's postings were decoded when the segment was opened
    tl->n = t.docs_count == 1 ? 1u : t.tail_n;
    tl->first_doc = tl->n ? seg.tail_docs[t.tail_row] : 0u;
    tl->last_doc = tl->n ? t.last_doc : 0u;
  }
}

// ------------------------------------------------------------ tile score --
//
// A workgroup owns one doc tile [lo, lo+TILE) of one query.  Per-doc score
// accumulators live in LDS as FIXED-POINT integers (ACC = 64-bit, or 32-bit when
// the query's score range allows it): integer addition is associative, so the
// (term, block) work items of a tile are processed by the wavefronts in any
// order, with no barrier between terms, and the sum is still bit-reproducible.
// Each posting's score follows the reference's float expression; the fixed-point
// sum differs from the reference's sequential float sum by rounding only.
//
// The kernel is bound by instruction issue (VALU + SALU), not by HBM, LDS or
// latency (rocprofv3 PMC, profiles/): everything below is written to minimise
// wave-instructions per 128-posting block.

struct TermL {          // one query term of this tile, staged in LDS
  uint64_t doc_start;   // absolute offset of the term's postings
  uint64_t dir_off;     // first directory entry of the term
  uint32_t b0;          // first block overlapping the tile
  uint32_t nb;          // number of blocks overlapping the tile
  uint32_t item_off;    // prefix sum of nb over the query's terms
  uint32_t tail_n;      // postings in the term's decoded tail
};

struct alignas(16) ItemL {   // one (term, block) work item: one 16-byte LDS read per lane
  uint32_t off;         // straight-line item: offset in the packed-payload image (16-byte
                        // units); generic item: byte offset of the block in `.doc`
  uint32_t base;        // last doc of the preceding block (kDocMin for block 0)
  uint32_t pack;        // doc bits | freq bits << 8 | cache slot << 16 | term slot << 20 | fast << 31
  float cs;             // the term's c0 pre-multiplied by the fixed-point scale
};

// Scorers of the straight-line path are all evaluated from one 256-entry table row
// `tab` in LDS (indexed by the doc's norm byte), in one of two forms:
//   reciprocal form  score = c0 - c0 / (1 + tf * tab[norm])
//     BM25, 1-byte norms  tab[n] = norm_cache[n] = 1/(norm_const + norm_length*n), [0] = 0
//                         (bm25.cpp:348-353, 404-409)
//     BM25, no norms      tab[n] = 1/(norm_const + norm_length)   (norm == 1, bm25.cpp:487-489)
//     BM15                tab[n] = 1/norm_const                    (bm25.cpp:313)
//   square-root form score = sqrt(tf) * c0 * tab[norm]
//     TF-IDF              tab[n] = 1                               (tfidf.cpp:185-187)
//     TF-IDF with norms   tab[n] = 1/sqrt(n), [0] = 0              (tfidf.cpp:251-253)
// Rows that ignore the norm are constant, so whatever byte the norm stage reads is fine.
__device__ __forceinline__ bool table_kind(int32_t kind) {
  return kind == kBM25Tiny || kind == kBM25One || kind == kBM15 || kind == kTfidf ||
         kind == kTfidfTiny;
}
__device__ __forceinline__ bool sqrt_kind(int32_t kind) {
  return kind == kTfidf || kind == kTfidfTiny;
}
__device__ __forceinline__ float table_value(int32_t kind, float nc, float nl, uint32_t n) {
  switch (kind) {
    case kBM25Tiny: return n ? 1.f / (nc + nl * static_cast<float>(n)) : 0.f;
    case kBM25One: return 1.f / (nc + nl * 1.f);
    case kBM15: return 1.f / nc;
    case kTfidf: return 1.f;
    default: return n ? 1.f / sqrtf(static_cast<float>(n)) : 0.f;  // kTfidfTiny
  }
}
// One thread per table entry: the row of slot c comes from the first term using it.
template<typename SM>
__device__ __forceinline__ void build_tables(const SM& sm, uint32_t n_caches, uint32_t n_terms) {
  for (uint32_t e = threadIdx.x; e < n_caches * 256u; e += blockDim.x) {
    const uint32_t c = e >> 8, n = e & 255u;
    float v = 0.f;
    for (uint32_t j = 0; j < n_terms; ++j) {
      if (sm.qts[j].cache_id == c) {
        v = table_value(sm.qts[j].kind, sm.qts[j].norm_const, sm.qts[j].norm_length, n);
        break;
      }
    }
    sm.caches[e] = v;
  }
}

// Bits 16.. of ItemL::pack for term slot j: table slot (16-19), term slot (20-24),
// square-root form (25); bit 30 = "this scorer has a straight-line path", resolved
// against the block's bit widths by item_pack().
constexpr uint32_t kPackSqrt = 0x02000000u;
__device__ __forceinline__ uint32_t term_pack(uint32_t j, int32_t kind, uint32_t cache_id) {
  const bool fk = table_kind(kind) && cache_id < kMaxCaches;
  return ((cache_id < 15u ? cache_id : 15u) << 16) | (j << 20) |
         (sqrt_kind(kind) ? kPackSqrt : 0u) | (fk ? 0x40000000u : 0u);
}
__device__ __forceinline__ uint32_t item_pack(uint32_t bits16, uint32_t tpack) {
  const uint32_t dbits = bits16 & 0xFFu, fbits = (bits16 >> 8) & 0xFFu;
  // the straight-line decoder handles 1..31-bit packed blocks
  const bool fast = (tpack & 0x40000000u) && (dbits - 1u) <= 30u && (fbits - 1u) <= 30u;
  return (bits16 & 0xFFFFu) | (tpack & 0x03FF0000u) | (fast ? 0x80000000u : 0u);
}

template<typename ACC>
struct TileSmemT {
  ACC* acc;          // [TILE + 64] fixed-point score accumulators (score_buf of
                     // block_disjunction, disjunction.hpp:1087-1092, widened to TILE docs)
                     // + one private dummy slot per lane
  uint32_t* cnt;     // [(TILE + 64)/4] per-doc match counters, 1 byte each (AND only; + dummies)
  uint8_t* lnorm;    // [TILE] Norm2 bytes of the tile
  float* caches;     // [kMaxCaches][256] BM25Stats::norm_cache
  DevQTerm* qts;     // [kMaxTerms] the query's term scorers
  TermL* tl;         // [kMaxTerms]
  ItemL* items;      // [kItemChunk]
  uint32_t* vars;    // [8] 0: item count
};

template<typename ACC, int TILE, bool AND>
constexpr uint32_t tile_smem_bytes() {
  return uint32_t(sizeof(ACC)) * (TILE + 64) + (AND ? TILE + 64 : 0) + TILE +
         sizeof(float) * 256 * kMaxCaches + sizeof(DevQTerm) * kMaxTerms +
         sizeof(TermL) * kMaxTerms + sizeof(ItemL) * kItemChunk + 32;
}

// Byte offsets of the tile arrays inside the workgroup's LDS block (== carve() below);
// the hot path addresses them absolutely (wave::lds_*).
template<typename ACC, int TILE, bool AND>
struct TileOff {
  static constexpr uint32_t acc = 0;
  static constexpr uint32_t cnt = uint32_t(sizeof(ACC)) * (TILE + 64);
  static constexpr uint32_t lnorm = cnt + (AND ? TILE + 64 : 0);
  static constexpr uint32_t caches = lnorm + TILE;
};

template<typename ACC, int TILE, bool AND>
__device__ __forceinline__ TileSmemT<ACC> carve(unsigned char* smem, unsigned char** rest) {
  TileSmemT<ACC> sm;
  sm.acc = reinterpret_cast<ACC*>(smem);
  smem += sizeof(ACC) * (TILE + 64);
  sm.cnt = reinterpret_cast<uint32_t*>(smem);
  if (AND) smem += TILE + 64;
  sm.lnorm = smem;
  smem += TILE;
  sm.caches = reinterpret_cast<float*>(smem);
  smem += sizeof(float) * 256 * kMaxCaches;
  sm.qts = reinterpret_cast<DevQTerm*>(smem);
  smem += sizeof(DevQTerm) * kMaxTerms;
  sm.tl = reinterpret_cast<TermL*>(smem);
  smem += sizeof(TermL) * kMaxTerms;
  sm.items = reinterpret_cast<ItemL*>(smem);
  smem += sizeof(ItemL) * kItemChunk;
  sm.vars = reinterpret_cast<uint32_t*>(smem);
  smem += 32;
  *rest = smem;
  return sm;
}

// A score already multiplied by DevQuery::fx_mul (a power of two) -> fixed point.
// 64-bit: x < 2^29 is the HIGH word, the fraction becomes the low word (2^E units,
// E = 61 - ceil(log2 U)).  32-bit: x < 2^30 truncated (2^(30-e) units).  `| 1`
// keeps every posting's contribution non-zero: "accumulator != 0" == "matched".
template<typename ACC>
__device__ __forceinline__ ACC fixed_from_scaled(float x);
template<>
__device__ __forceinline__ unsigned long long fixed_from_scaled<unsigned long long>(float x) {
  const uint32_t hi = static_cast<uint32_t>(x);   // truncates
  const float rem = x - static_cast<float>(hi);   // exact
  const uint32_t lo = static_cast<uint32_t>(rem * 4294967296.f);
  return ((static_cast<unsigned long long>(hi) << 32) | lo) | 1ull;
}
template<>
__device__ __forceinline__ uint32_t fixed_from_scaled<uint32_t>(float x) {
  return static_cast<uint32_t>(x) | 1u;
}
template<typename ACC>
__device__ __forceinline__ float from_fixed(ACC a, float fx_inv) {
  return static_cast<float>(a) * fx_inv;
}

__device__ __forceinline__ uint32_t norm_global(const DevSegment& seg, uint32_t doc) {
  // dense fixed-length column, big-endian values (columnstore2.cpp:736-740, norm.hpp:170-182)
  const uint8_t* p = seg.norms + uint64_t(seg.norm_width) * (doc - seg.norm_min_doc);
  if (seg.norm_width == 2) return (uint32_t(p[0]) << 8) | p[1];
  return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3];
}

// Score of one posting — the reference's float expressions, evaluated in the
// same order with no FMA contraction (bm25.cpp:313, 353, 359; tfidf.cpp:185-187, 251-253).
template<typename SM>
__device__ __forceinline__ float score_posting(const DevSegment& seg, const DevQTerm& qt,
                                               float inv_one, const SM& sm, uint32_t freq,
                                               uint32_t doc, uint32_t idx) {
  const float tf = static_cast<float>(freq);
  switch (qt.kind) {
    case kBM1:
      return qt.c0;
    case kBM15:
      return qt.c0 - qt.c0 / (1.f + tf / qt.norm_const);
    case kBM25Tiny: {
      const uint32_t n = sm.lnorm[idx];
      float inv;
      if (qt.cache_id < kMaxCaches) {
        inv = sm.caches[qt.cache_id * 256u + n];
      } else {
        inv = n ? 1.f / (qt.norm_const + qt.norm_length * static_cast<float>(n)) : 0.f;
      }
      return qt.c0 - qt.c0 / (1.f + tf * inv);
    }
    case kBM25One:
      return qt.c0 - qt.c0 / (1.f + tf * inv_one);
    case kBM25Wide: {
      const float c1 = qt.norm_const +
                       qt.norm_length * static_cast<float>(norm_global(seg, doc));
      return qt.c0 - qt.c0 * c1 / (c1 + tf);
    }
    case kTfidf:
      return sqrtf(tf) * qt.c0;
    case kTfidfTiny: {
      const uint32_t n = sm.lnorm[idx];
      const float r = n ? 1.f / sqrtf(static_cast<float>(n)) : 0.f;
      return sqrtf(tf) * qt.c0 * r;
    }
    default: {  // kTfidfWide
      const uint32_t n = norm_global(seg, doc);
      const float r = n ? 1.f / sqrtf(static_cast<float>(n)) : 0.f;
      return sqrtf(tf) * qt.c0 * r;
    }
  }
}

// generic scorer (every kind), used off the hot path
template<typename ACC, int TILE, bool AND>
__device__ __forceinline__ void tile_apply(const DevSegment& seg, const TileSmemT<ACC>& sm,
                                           const DevQTerm& qt, float inv_one, uint32_t doc,
                                           uint32_t freq, uint32_t lo, uint32_t span,
                                           float fx_mul) {
  const uint32_t idx = doc - lo;  // doc < lo wraps to a huge value
  if (idx < span) {
    const float s = score_posting(seg, qt, inv_one, sm, freq, doc, idx);
    atomicAdd(&sm.acc[idx], fixed_from_scaled<ACC>(s * fx_mul));
    if (AND) atomicAdd(&sm.cnt[idx >> 2], 1u << (8u * (idx & 3u)));
  }
}

// Scoring of N postings at once on the hot path: any scorer of the table family
// (see table_kind above; `tab[k]` is the LDS byte offset of posting k's table row).
// Reciprocal form, e.g. BM25 over 1-byte norms — bm25.cpp:348-353:
// c0 - c0/(1 + tf*norm_cache[norm]): the division is one v_rcp_f32 and the two
// multiply-adds are fused; square-root form (TF-IDF): one v_sqrt_f32 and two
// multiplies.  Either is within 2 ulp of the reference expression, far inside the
// 1e-5 parity tolerance; `cs` is c0 pre-multiplied by fx_mul so the result is
// already in fixed-point units.
// Staged so that the N norm-byte reads, then the N table reads, then the N LDS
// atomics are issued back to back: one LDS latency per stage instead of one per
// posting.  wave::keep*() pins each stage (the compiler would otherwise sink the
// whole computation behind a per-posting branch).
// Postings outside the tile are not branched around: `doc - lo` wraps to a huge
// value for doc < lo, and one v_min clamps every out-of-tile index to the lane's
// private dummy accumulator acc[TILE + lane]; whatever byte sits at
// lnorm[TILE + lane] (the next LDS array) yields some finite garbage that is
// added to that dummy slot, which nothing ever reads.  (The last tile of a
// segment needs no extra test: docs >= lo + span do not exist.)
template<typename ACC, int TILE, bool AND, int N>
__device__ __forceinline__ void tile_post_bm25(const TileSmemT<ACC>& sm, const float (&cs)[N],
                                               const uint32_t (&tab)[N],
                                               const uint32_t (&doc)[N],
                                               const uint32_t (&freq)[N], uint32_t lo,
                                               unsigned lane, bool sqrt_form) {
  using Off = TileOff<ACC, TILE, AND>;
  const unsigned char* base = reinterpret_cast<const unsigned char*>(sm.acc);  // LDS offset 0
  uint32_t idx[N], nb[N];
  float inv[N];
  ACC fx[N];
  const uint32_t dummy = uint32_t(TILE) + lane;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const uint32_t raw = doc[k] - lo;
    idx[k] = raw < dummy ? raw : dummy;
    nb[k] = wave::lds_u8(base, Off::lnorm + idx[k]);
  }
  wave::keep_all(nb);   // one asm statement over all N values: one s_waitcnt
#pragma unroll
  for (int k = 0; k < N; ++k) inv[k] = wave::lds_f32(base, tab[k] + nb[k] * 4u);
  wave::keep_all_f(inv);
#ifdef IRS_NO_SQRT_FORM   // A/B experiment only
  sqrt_form = false;
#endif
  if (sqrt_form) {   // wave-uniform: one scorer per query
#pragma unroll
    for (int k = 0; k < N; ++k) {
      float scaled = wave::fast_sqrt(static_cast<float>(freq[k])) * cs[k] * inv[k];
      wave::keep_f(scaled);
      fx[k] = fixed_from_scaled<ACC>(scaled);
    }
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float r = wave::fast_rcp(wave::fma(static_cast<float>(freq[k]), inv[k], 1.f));
      float scaled = wave::fma(-cs[k], r, cs[k]);
      wave::keep_f(scaled);
      fx[k] = fixed_from_scaled<ACC>(scaled);
    }
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    wave::lds_add(base, Off::acc + idx[k] * uint32_t(sizeof(ACC)), fx[k]);
    // (out-of-tile postings bump a dummy counter byte, like their dummy accumulator)
    if (AND) wave::lds_add(base, Off::cnt + (idx[k] & ~3u), 1u << (8u * (idx[k] & 3u)));
  }
}

// Payload words of values 2*lane, 2*lane+1 of a packed block, 1 <= bits <= 32, with no
// branch on `bits` (any other framing is re-read by the generic path): the
// same address arithmetic as raw_load.  Reads at most 16 bytes past the block.
template<int LAYOUT>
__device__ __forceinline__ void raw_load_packed(const uint8_t* payload, uint32_t bits,
                                                unsigned lane, uint64_t& a, uint64_t& b) {
#ifdef IRS_ABL_ALIGNED   // timing experiment only (wrong results): 8-byte aligned addresses
  payload = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(payload) & ~uintptr_t(7));
#endif
  if (LAYOUT == kSimd4) {
#ifdef IRS_ABL_PINNED    // timing experiment only (wrong results): every lane reads the same 8 bytes
    const uint32_t k = 0;
    const uint32_t voff = 0;
#else
    const uint32_t k = wave::mul24(lane >> 1, bits) >> 5;
    const uint32_t voff = 16u * k + ((lane & 1u) << 3);
#endif
    a = wave::load_u64(payload + voff);
    b = wave::load_u64(payload + voff + 16);
  } else {
    const uint32_t voff = (wave::mul24(lane << 1, bits) >> 5) << 2;
    a = wave::load_u64(payload + voff);
    b = wave::load_u64(payload + voff + 4);
  }
}

// Values 2*lane, 2*lane+1 out of the prefetched payload words, for 1 <= bits <= 31:
// one funnel shift (v_alignbit_b32) + one bit-field extract (v_bfe_u32) each.
template<int LAYOUT>
__device__ __forceinline__ void extract_fast(uint64_t a, uint64_t b, uint32_t bits,
                                             unsigned lane, uint32_t& v0, uint32_t& v1) {
  if (LAYOUT == kSimd4) {
    const uint32_t s = wave::mul24(lane >> 1, bits) & 31u;
    v0 = wave::bfe(wave::funnel(uint32_t(b), uint32_t(a), s), bits);
    v1 = wave::bfe(wave::funnel(uint32_t(b >> 32), uint32_t(a >> 32), s), bits);
  } else {
    const uint32_t s = wave::mul24(lane << 1, bits) & 31u;
    const uint32_t w0 = uint32_t(a), w1 = uint32_t(a >> 32), w2 = uint32_t(b >> 32);
    v0 = wave::bfe(wave::funnel(w1, w0, s), bits);
    const uint32_t s1 = s + bits;  // <= 62
    const bool hi = s1 >= 32u;
    v1 = wave::bfe(wave::funnel(hi ? w2 : w1, hi ? w1 : w0, s1 & 31u), bits);
  }
}

// The work items of one tile are processed in two steps so that k_score can put
// a whole tile epilogue between them:
//   items_prepare: wavefront w takes items w, w+nw, ...; lane k loads the metadata
//     of the k-th one (the loop broadcasts it with v_readlane: scalar results, no
//     LDS round trip on an item's critical path) and the payload words of the
//     first two items are requested;
//   items_run: decode + score + accumulate, always two items ahead with the loads.
constexpr uint32_t kPackPair = 0x40000000u;   // ItemRegs::pack, set by items_prepare

struct ItemRegs {
  uint32_t n;                        // items of this wavefront (<= 64), wave-uniform
  uint32_t pack, base, off;          // lane k: metadata of the k-th item
  float cs;
  uint64_t ada, adb, afa, afb;       // payload words of the next item pair
  uint64_t bda, bdb, bfa, bfb;
};

// raw payload words of item k (two per block part), from the packed-payload image
// where every part starts 16-byte aligned; no branch on the bit widths.  Items of
// the generic path read `.doc` for themselves.
template<int LAYOUT>
__device__ __forceinline__ void item_load(const DevSegment& seg, const ItemRegs& r, uint32_t k,
                                          unsigned lane, uint64_t& da, uint64_t& db,
                                          uint64_t& fa, uint64_t& fb) {
  const uint32_t pack = wave::read_lane(r.pack, k & 63u);
  if (pack >> 31) {   // scalar branch: only straight-line items live in the packed image
    const uint32_t dbits = pack & 0xFFu, fbits = (pack >> 8) & 0xFFu;
    const uint8_t* blk = seg.pk + (uint64_t(wave::read_lane(r.off, k & 63u)) << 4);
    raw_load_packed<LAYOUT>(blk, dbits, lane, da, db);
    raw_load_packed<LAYOUT>(blk + 16u * dbits, fbits, lane, fa, fb);   // right behind
  }
}

template<int LAYOUT>
__device__ __forceinline__ void items_prepare(const DevSegment& seg, const ItemL* items,
                                              uint32_t n, uint32_t inv_nw, ItemRegs& r) {
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t wv = wave::uniform(threadIdx.x >> 6);
  const uint32_t nw = blockDim.x >> 6;
  // ceil((n - wv) / nw) for n <= 256, nw <= 16, with inv_nw = ceil(2^16 / nw)
  r.n = wave::uniform(n > wv ? ((n - wv + nw - 1) * inv_nw) >> 16 : 0u);
  // lanes past the last item never carry the straight-line flag: nothing is loaded
  r.pack = 0x0101u;
  r.base = 0;
  r.off = 0;
  r.cs = 0.f;
  if (r.n) {
    const uint32_t mine = lane < r.n ? lane : r.n - 1u;
    const ItemL I = items[wv + mine * nw];
    r.off = I.off;
    r.base = I.base;
    r.pack = lane < r.n ? I.pack : (I.pack & 0xFFFFu);
    r.cs = I.cs;
  }
  // bit 30 of lane k: items k and k+1 are both straight-line, same score form (read for even k only)
  const uint32_t next = __shfl_down(r.pack, 1, 64);
  if (((r.pack & next) >> 31) && !((r.pack ^ next) & kPackSqrt)) r.pack |= kPackPair;
  r.ada = r.adb = r.afa = r.afb = r.bda = r.bdb = r.bfa = r.bfb = 0;
  item_load<LAYOUT>(seg, r, 0, lane, r.ada, r.adb, r.afa, r.afb);
  item_load<LAYOUT>(seg, r, 1, lane, r.bda, r.bdb, r.bfa, r.bfb);
}

template<typename ACC, int LAYOUT, int TILE, bool AND>
__device__ __forceinline__ void items_run(const DevSegment& seg, const TileSmemT<ACC>& sm,
                                          ItemRegs& r, uint32_t lo, uint32_t span,
                                          float fx_mul) {
  const unsigned lane = threadIdx.x & 63u;
  // generic item: any block framing, any scorer; does its own loads
  auto slow_item = [&](uint32_t k) {
    const uint32_t pack = wave::read_lane(r.pack, k);
    const uint32_t base = wave::read_lane(r.base, k);
    const uint32_t j = (pack >> 20) & 0x1Fu;
    uint32_t d0, d1, f0, f1;
    decode_block<LAYOUT, true>(seg.doc + wave::read_lane(r.off, k), pack & 0xFFu,
                               (pack >> 8) & 0xFFu, base, lane, d0, d1, f0, f1);
    const DevQTerm qt = sm.qts[j];
    const float inv_one = 1.f / (qt.norm_const + qt.norm_length * 1.f);
    tile_apply<ACC, TILE, AND>(seg, sm, qt, inv_one, d0, f0, lo, span, fx_mul);
    tile_apply<ACC, TILE, AND>(seg, sm, qt, inv_one, d1, f1, lo, span, fx_mul);
  };
  // hot path, one item: straight-line code
  auto fast_item = [&](uint32_t k, uint64_t da, uint64_t db, uint64_t fa, uint64_t fb) {
    const uint32_t pack = wave::read_lane(r.pack, k);
    const float cs = wave::read_lane_f(r.cs, k);
    const uint32_t tab = TileOff<ACC, TILE, AND>::caches + ((pack >> 16) & 0xFu) * 1024u;
    uint32_t x0, x1, f0, f1;
    extract_fast<LAYOUT>(da, db, pack & 0xFFu, lane, x0, x1);
    extract_fast<LAYOUT>(fa, fb, (pack >> 8) & 0xFFu, lane, f0, f1);
    const uint32_t d1 = wave::read_lane(r.base, k) + wave::inclusive_scan(x0 + x1);
    const float css[2] = {cs, cs};
    const uint32_t tabs2[2] = {tab, tab};
    const uint32_t docs2[2] = {d1 - x1, d1};
    const uint32_t freqs2[2] = {f0, f1};
    tile_post_bm25<ACC, TILE, AND, 2>(sm, css, tabs2, docs2, freqs2, lo, lane,
                                      (pack & kPackSqrt) != 0u);
  };
  // hot path, two items fused: 4 postings per lane in flight, two independent
  // DPP scan chains, all LDS lookups issued back to back
  auto fast_pair = [&](uint32_t k, uint64_t ada, uint64_t adb, uint64_t afa, uint64_t afb,
                       uint64_t bda, uint64_t bdb, uint64_t bfa, uint64_t bfb) {
    const uint32_t pA = wave::read_lane(r.pack, k), pB = wave::read_lane(r.pack, k + 1);
    const float csA = wave::read_lane_f(r.cs, k), csB = wave::read_lane_f(r.cs, k + 1);
    const uint32_t tabA = TileOff<ACC, TILE, AND>::caches + ((pA >> 16) & 0xFu) * 1024u;
    const uint32_t tabB = TileOff<ACC, TILE, AND>::caches + ((pB >> 16) & 0xFu) * 1024u;
    uint32_t ax0, ax1, af0, af1, bx0, bx1, bf0, bf1;
    extract_fast<LAYOUT>(ada, adb, pA & 0xFFu, lane, ax0, ax1);
    extract_fast<LAYOUT>(bda, bdb, pB & 0xFFu, lane, bx0, bx1);
    extract_fast<LAYOUT>(afa, afb, (pA >> 8) & 0xFFu, lane, af0, af1);
    extract_fast<LAYOUT>(bfa, bfb, (pB >> 8) & 0xFFu, lane, bf0, bf1);
    uint32_t sa = ax0 + ax1, sb = bx0 + bx1;
    wave::inclusive_scan2(sa, sb);
    const uint32_t ad1 = wave::read_lane(r.base, k) + sa;
    const uint32_t bd1 = wave::read_lane(r.base, k + 1) + sb;
    const float css[4] = {csA, csA, csB, csB};
    const uint32_t tabs4[4] = {tabA, tabA, tabB, tabB};
    const uint32_t docs4[4] = {ad1 - ax1, ad1, bd1 - bx1, bd1};
    const uint32_t freqs4[4] = {af0, af1, bf0, bf1};
    tile_post_bm25<ACC, TILE, AND, 4>(sm, css, tabs4, docs4, freqs4, lo, lane,
                                      (pA & kPackSqrt) != 0u);
  };

  uint64_t ada = r.ada, adb = r.adb, afa = r.afa, afb = r.afb;
  uint64_t bda = r.bda, bdb = r.bdb, bfa = r.bfa, bfb = r.bfb;
  const uint32_t my_n = r.n;
  for (uint32_t k = 0; k < my_n; k += 2) {
    // look-ahead loads of items k+2, k+3 (scalar branches on the items' straight-line
    // flag — lanes past the last item never carry it; measured faster than
    // unconditional loads: the vector memory pipeline is a scarce resource here).
    // Registers of an item that is not loaded are never read: left undefined.
    uint64_t nada = wave::undef64(), nadb = wave::undef64(), nafa = wave::undef64(),
             nafb = wave::undef64(), nbda = wave::undef64(), nbdb = wave::undef64(),
             nbfa = wave::undef64(), nbfb = wave::undef64();
    item_load<LAYOUT>(seg, r, k + 2, lane, nada, nadb, nafa, nafb);
    item_load<LAYOUT>(seg, r, k + 3, lane, nbda, nbdb, nbfa, nbfb);
    const uint32_t pA = wave::read_lane(r.pack, k);
    if (pA & kPackPair) {   // items k and k+1 both exist and are straight-line
      fast_pair(k, ada, adb, afa, afb, bda, bdb, bfa, bfb);
    } else {
      if (pA >> 31) fast_item(k, ada, adb, afa, afb); else slow_item(k);
      if (k + 1 < my_n) {
        if (wave::read_lane(r.pack, k + 1) >> 31) fast_item(k + 1, bda, bdb, bfa, bfb);
        else slow_item(k + 1);
      }
    }
    ada = nada; adb = nadb; afa = nafa; afb = nafb;
    bda = nbda; bdb = nbdb; bfa = nbfa; bfb = nbfb;
  }
}

template<typename ACC, int LAYOUT, int TILE, bool AND>
__device__ __forceinline__ void process_items(const DevSegment& seg, const TileSmemT<ACC>& sm,
                                              const ItemL* items, uint32_t n, uint32_t lo,
                                              uint32_t span, float fx_mul) {
  ItemRegs r;
  const uint32_t nw = blockDim.x >> 6;
  items_prepare<LAYOUT>(seg, items, n, (65536u + nw - 1) / nw, r);
  items_run<ACC, LAYOUT, TILE, AND>(seg, sm, r, lo, span, fx_mul);
}

__device__ __forceinline__ uint32_t score_bin(float v, float scale) {
  const float x = fminf(v * scale, float(kBins - 1));
  return uint32_t(x);
}

// A conservative fixed-point image of the lower edge of score bin `bs` (the
// exact float bin test follows for the few accumulators that pass it).
template<typename ACC>
__device__ __forceinline__ ACC bin_threshold(uint32_t bs, const DevQuery& qd) {
  if (!bs) return ACC(1);
  const double edge = double(bs) / double(qd.bin_scale);
  return static_cast<ACC>(edge / double(qd.fx_inv) * (1.0 - 1e-6));
}

// ----------------------------------------------------------------- pilot --

// Zero the accumulators, stage the tile's norms, and — all terms in parallel,
// one thread each, every load independent — fetch each term's block range.
template<typename ACC, int TILE, bool AND>
__device__ __forceinline__ void tile_begin(const DevSegment& seg, const DevQuery& qd,
                                           const DevQTerm* qts_g, const uint32_t* first_q,
                                           uint32_t jt, const DevTail* tails_q, uint32_t tile,
                                           const TileSmemT<ACC>& sm, bool build_caches) {
  if (threadIdx.x < qd.n_terms) {
    const uint32_t j = threadIdx.x;
    if (build_caches) sm.qts[j] = qts_g[j];
    const DevTail* tl = tails_q + j;
    const uint32_t nblk = tl->nblk, tn = tl->n;
    const uint64_t doc_start = tl->doc_start, dir_off = tl->dir_off;
    const uint32_t b0 = first_q[uint64_t(tile) * jt + j];
    uint32_t b1 = first_q[uint64_t(tile + 1) * jt + j] + 1u;
    b1 = b1 < nblk ? b1 : nblk;
    sm.tl[j].doc_start = doc_start;
    sm.tl[j].dir_off = dir_off;
    sm.tl[j].b0 = b0;
    sm.tl[j].nb = b1 > b0 ? b1 - b0 : 0u;
    sm.tl[j].item_off = 0;
    sm.tl[j].tail_n = tn;
  }
  for (uint32_t i = threadIdx.x; i < uint32_t(TILE) + 64u; i += blockDim.x) sm.acc[i] = ACC(0);
  if (AND) {
    for (uint32_t i = threadIdx.x; i < TILE / 4; i += blockDim.x) sm.cnt[i] = 0u;
  }
  if (seg.norms && seg.norm_width == 1) {
    const uint64_t first = uint64_t(tile) * TILE + (kDocMin - seg.norm_min_doc);
    for (uint32_t i = threadIdx.x * 4; i < TILE; i += blockDim.x * 4) {
      // norms are staged with kPadBytes of slack, 4-byte granules stay in bounds
      uint32_t w = 0;
      if (first + i < seg.norm_count) w = wave::load_u32(seg.norms + first + i);
      *reinterpret_cast<uint32_t*>(sm.lnorm + i) = w;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t off = 0;
    for (uint32_t j = 0; j < qd.n_terms; ++j) {
      sm.tl[j].item_off = off;
      off += sm.tl[j].nb;
    }
    sm.vars[0] = off;
  }
  if (build_caches) {
    build_tables(sm, qd.n_caches, qd.n_terms);
  }
  __syncthreads();
}

// All postings of the query's terms that fall into doc tile `tile`: the GPU
// form of block_disjunction::refill (disjunction.hpp:1240-1351), with the
// 512-doc window widened to TILE docs in LDS, and of Conjunction via per-doc
// match counters.  (Used by k_pilot; k_score pipelines the same pieces.)
template<typename ACC, int LAYOUT, int TILE, bool AND>
__device__ __forceinline__ void tile_accumulate(const DevSegment& seg, const DevQuery& qd,
                                                const DevTail* tails_q, uint32_t tile,
                                                const TileSmemT<ACC>& sm) {
  const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t nw = blockDim.x >> 6;
  const uint32_t lo = kDocMin + tile * TILE;
  const uint32_t span = (seg.num_docs + kDocMin - lo) < uint32_t(TILE)
                          ? (seg.num_docs + kDocMin - lo) : uint32_t(TILE);
  const float fx_mul = qd.fx_mul;
  const uint32_t n_items = sm.vars[0];
  for (uint32_t c0 = 0; c0 < n_items; c0 += kItemChunk) {
    const uint32_t n = (n_items - c0) < kItemChunk ? (n_items - c0) : kItemChunk;
    if (c0) __syncthreads();  // previous chunk fully consumed
    if (threadIdx.x < n) {
      const uint32_t id = c0 + threadIdx.x;
      uint32_t j = 0;
      while (id >= sm.tl[j].item_off + sm.tl[j].nb) ++j;
      const uint32_t b = sm.tl[j].b0 + (id - sm.tl[j].item_off);
      const uint64_t e = sm.tl[j].dir_off + b;
      ItemL I;
      I.base = b ? seg.blk_last[e - 1] : kDocMin;
      I.pack = item_pack(seg.blk_bits[e], term_pack(j, sm.qts[j].kind, sm.qts[j].cache_id));
      I.off = (I.pack >> 31) ? seg.blk_aoff[e] : uint32_t(sm.tl[j].doc_start) + seg.blk_off[e];
      I.cs = sm.qts[j].c0 * fx_mul;
      sm.items[threadIdx.x] = I;
    }
    __syncthreads();
    process_items<ACC, LAYOUT, TILE, AND>(seg, sm, sm.items, n, lo, span, fx_mul);
  }
  // decoded vint tails / single-doc terms (k_plan), one term per wavefront
  for (uint32_t j = wv; j < qd.n_terms; j += nw) {
    const uint32_t tn = sm.tl[j].tail_n;
    const DevTail* tl = tails_q + j;
    if (tn && tl->first_doc < lo + span && tl->last_doc >= lo) {
      const DevQTerm qt = sm.qts[j];
      const float inv_one = 1.f / (qt.norm_const + qt.norm_length * 1.f);
      for (uint32_t i = lane; i < tn; i += 64)
        tile_apply<ACC, TILE, AND>(seg, sm, qt, inv_one,
                                   seg.tail_docs[tl->tail_row + i],
                                   seg.tail_freqs[tl->tail_row + i], lo, span,
                                   fx_mul);
    }
  }
  __syncthreads();
}

// One workgroup per query scores the tiles {phase, phase+P, ...}, histograms
// their scores into kBins linear bins over [0, U] and picks a bin b*; k_score
// drops everything below b*.
//   sound (margin == 0): the largest bin with at least k sampled docs at or above
//     it.  Those docs exist, so the final k-th score is >= the lower edge of b*.
//     The full set then holds about k*P candidates.
//   estimated (margin > 0): the largest bin with at least margin*k*(sampled
//     tiles)/(all tiles) sampled docs at or above it (never more than k, never less
//     than kPilotMinSample), i.e. an expected margin*k candidates.  Not a proof:
//     k_select checks "fewer than k candidates although more docs matched"
//     (kStatusUnderflow) and the host then re-runs the batch in sound mode.
constexpr uint32_t kPilotMinSample = 48;

template<typename ACC, int LAYOUT, int TILE, bool AND>
__global__ void __launch_bounds__(kTileThreadsMax)
k_pilot(const DevSegment* segs, const DevQuery* queries, const DevQTerm* qterms, uint32_t jt,
        uint32_t stride, const uint32_t* first, const DevTail* tails,
        uint32_t* bstar, uint32_t margin) {
  RT_DYN_SMEM(smem);
  if (!wave::lds_is_at_zero(smem)) __builtin_trap();  // the tile arrays are addressed absolutely
  unsigned char* rest;
  const TileSmemT<ACC> sm = carve<ACC, TILE, AND>(smem, &rest);
  uint32_t* hist = reinterpret_cast<uint32_t*>(rest);  // [kBins]
  const uint32_t q = blockIdx.x;
  const DevQuery qd = queries[q];
  const DevQTerm* qts = qterms + qd.first_term;
  const DevSegment seg = segs[qd.seg];
  const uint32_t n_tiles = qd.n_tiles;
  const uint32_t* first_q = first + qd.first_off;
  const DevTail* tails_q = tails + uint64_t(q) * jt;
  for (uint32_t i = threadIdx.x; i < kBins; i += blockDim.x) hist[i] = 0u;
  bool first_tile = true;
  for (uint32_t tile = (q * 7u) % stride; tile < n_tiles; tile += stride) {
    tile_begin<ACC, TILE, AND>(seg, qd, qts, first_q, jt, tails_q, tile, sm, first_tile);
    first_tile = false;
    tile_accumulate<ACC, LAYOUT, TILE, AND>(seg, qd, tails_q, tile, sm);
    for (uint32_t i = threadIdx.x; i < TILE; i += blockDim.x) {
      const ACC a = sm.acc[i];
      bool m = a != ACC(0);
      if (AND && (qd.op & 0xFF) == 1)  // AND / min-match: op = 1 | required matches << 8
        m = ((sm.cnt[i >> 2] >> (8u * (i & 3u))) & 0xFFu) >= uint32_t(qd.op >> 8);
      if (m) atomicAdd(&hist[score_bin(from_fixed<ACC>(a, qd.fx_inv), qd.bin_scale)], 1u);
    }
    __syncthreads();
  }
  // docs the sample must show at or above b*
  uint32_t need = qd.k;
  if (margin) {
    const uint32_t phase = (q * 7u) % stride;
    const uint32_t sampled = phase < n_tiles ? (n_tiles - phase + stride - 1) / stride : 0u;
    const uint64_t est = (uint64_t(margin) * qd.k * sampled + n_tiles - 1) / n_tiles;
    const uint32_t lo = est < kPilotMinSample ? kPilotMinSample : uint32_t(est < 0xFFFFFFFFull ? est : 0xFFFFFFFFull);
    need = lo < qd.k ? lo : qd.k;
  }
  // suffix search: lane L of wave 0 owns the 8 bins of chunk 63-L
  if (threadIdx.x < 64) {
    const unsigned lane = threadIdx.x;
    const uint32_t chunk = 63u - lane;
    uint32_t s = 0;
    for (uint32_t i = 0; i < kBins / 64; ++i) s += hist[chunk * (kBins / 64) + i];
    const uint32_t incl = wave::inclusive_scan(s);  // docs in chunks >= chunk
    const uint64_t reach = wave::ballot(incl >= need);
    uint32_t result = 0;
    if (reach) {
      const int src = __builtin_ctzll(reach);  // highest chunk reaching `need`
      const uint32_t above = wave::bcast(incl - s, src);
      const uint32_t c = 63u - uint32_t(src);
      uint32_t cum = above;
      for (int i = int(kBins / 64) - 1; i >= 0; --i) {
        cum += hist[c * (kBins / 64) + uint32_t(i)];
        if (cum >= need) { result = c * (kBins / 64) + uint32_t(i); break; }
      }
    }
    if (lane == 0) bstar[q] = result;
  }
}

// ----------------------------------------------------------------- score --
//
// Persistent, software-pipelined workgroups.  The grid is sized to fill the
// chip once; every workgroup pulls CHUNKS of kChunkTiles consecutive doc tiles
// of one query from a global counter and walks them with a 3-deep pipeline.
// While the wavefronts decode/score tile u out of registers and LDS,
//   * the directory entries and norm bytes of tile u+1 are in flight into
//     registers (requested during tile u-1), landing in LDS after the compute,
//   * right after the compute barrier the requests for tile u+2 go out, every
//     wavefront picks up its work items of tile u+1 and requests the payload of
//     the first two — all of it covered by tile u's epilogue,
//   * the returning atomic that reserves candidate slots for tile u-1 and the
//     dequeue of the next chunk are in flight the same way.
// Loaded values are kept RAW in registers until they are stored: any arithmetic
// on them would make the compiler wait for the load where it was issued.

#ifndef IRS_CHUNK_TILES
#define IRS_CHUNK_TILES 16
#endif
constexpr uint32_t kChunkTiles = IRS_CHUNK_TILES;
#ifndef IRS_SCORE_CANDS
#define IRS_SCORE_CANDS 128
#endif
constexpr uint32_t kScoreCands = IRS_SCORE_CANDS;   // per-tile candidate staging slots (x2 buffers)
constexpr uint32_t kToffStride = 20;    // words per row of the per-tile prefix table:
                                        // [0..16] exclusive prefix sums of the terms' block counts
                                        // (0xFFFFFFFF past n_terms), [17] the tile's item count

enum : uint32_t {  // indices into the workgroup's LDS scratch words
  kVChunk = 0,     // current chunk id
  kVHits = 1,      // hits of this chunk
  kVBase = 2,      // global candidate base of the previous tile
  kVBaseLast = 3,  // ... of the chunk's last tile (own word: slow threads may still read kVBase)
  kVNc0 = 4,       // kVNc0 + (u % 3): candidate count of tile u
  kVWords = 8,
};

struct alignas(16) TermC {   // per-chunk view of one query term (aliases TileSmemT::tl)
  uint64_t dir_off;     // first entry of the term in the block directory
  uint32_t dstart;      // byte offset of the term's posting list in `.doc`
  uint32_t tpack;       // term_pack()
  float cs;             // c0 * fx_mul
  uint32_t tail_n;
  uint32_t pad[2];
};
static_assert(sizeof(TermC) == sizeof(TermL), "TermC aliases the TermL table");

template<typename ACC, int TILE, bool AND>
constexpr uint32_t score_smem_bytes() {
  return tile_smem_bytes<ACC, TILE, AND>()                 // acc, cnt, lnorm, caches, qts, tl, items[0]
         + sizeof(ItemL) * kItemChunk                      // items[1]
         + 4u * (kChunkTiles + 1) * kMaxTerms              // rows
         + 4u * kChunkTiles * kToffStride                  // per-tile item prefix sums
         + 4u * 3u * kMaxTerms                             // nblk, tail first, tail last
         + 4u * kChunkTiles                                // per-tile tail masks
         + 8u * 2u * kScoreCands                           // candidate staging x2
         + 4u * kVWords;
}

#ifndef IRS_SCORE_WAVES_PER_EU
#define IRS_SCORE_ATTR
#else
#define IRS_SCORE_ATTR __attribute__((amdgpu_waves_per_eu(IRS_SCORE_WAVES_PER_EU, IRS_SCORE_WAVES_PER_EU)))
#endif
template<typename ACC, int LAYOUT, int TILE, bool AND>
__global__ void __launch_bounds__(kTileThreadsMax) IRS_SCORE_ATTR
k_score(const DevSegment* segs, const DevQuery* queries, const DevQTerm* qterms, uint32_t jt,
        uint32_t cpq /*chunks per unit*/, uint32_t n_queries, const uint32_t* first,
        const DevTail* tails,
        const uint32_t* bstar, uint64_t* cands, uint32_t cand_cap, uint32_t* cand_count,
        unsigned long long* hits, uint32_t* work_counter) {
  RT_DYN_SMEM(smem);
  if (!wave::lds_is_at_zero(smem)) __builtin_trap();  // the tile arrays are addressed absolutely
  unsigned char* rest;
  const TileSmemT<ACC> sm = carve<ACC, TILE, AND>(smem, &rest);
  // the two item tables are addressed as base + (u & 1) * delta: plain LDS pointer
  // arithmetic (selecting between two pointers would degrade to flat addressing)
  unsigned char* const items_b = reinterpret_cast<unsigned char*>(sm.items);
  const uint32_t items_delta = uint32_t(rest - items_b);
  rest += sizeof(ItemL) * kItemChunk;
  uint32_t* rows = reinterpret_cast<uint32_t*>(rest);      // [kChunkTiles + 1][kMaxTerms]
  rest += 4u * (kChunkTiles + 1) * kMaxTerms;
  uint32_t* toff = reinterpret_cast<uint32_t*>(rest);      // [kChunkTiles][kToffStride]
  rest += 4u * kChunkTiles * kToffStride;
  uint32_t* tnblk = reinterpret_cast<uint32_t*>(rest);
  uint32_t* tfirst = tnblk + kMaxTerms;
  uint32_t* tlast = tfirst + kMaxTerms;
  rest += 4u * 3u * kMaxTerms;
  uint32_t* tmask = reinterpret_cast<uint32_t*>(rest);     // [kChunkTiles]
  rest += 4u * kChunkTiles;
  uint64_t* lcand = reinterpret_cast<uint64_t*>(rest);      // [2][kScoreCands]
  rest += 8u * 2u * kScoreCands;
  uint32_t* vars = reinterpret_cast<uint32_t*>(rest);
  TermC* tc = reinterpret_cast<TermC*>(sm.tl);
  auto items_of = [&](uint32_t u) {
    return reinterpret_cast<ItemL*>(items_b + (u & 1u) * items_delta);
  };

  const uint32_t tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const uint32_t wv = wave::uniform(tid >> 6);
  const uint32_t nw = blockDim.x >> 6;
  const uint32_t inv_nw = (65536u + nw - 1) / nw;   // wavefronts per workgroup: 4..16
  // every (segment, query) unit owns `cpq` chunk ids (sized for the segment with the most
  // tiles; ids past a shorter segment's last tile are empty chunks)
  const uint32_t total_chunks = n_queries * cpq;

  for (uint32_t i = tid; i < uint32_t(TILE) + 64u; i += blockDim.x) sm.acc[i] = ACC(0);
  if (AND) {
    for (uint32_t i = tid; i < uint32_t(TILE) / 4; i += blockDim.x) sm.cnt[i] = 0u;
  }
  if (tid < kVWords) vars[tid] = 0u;
  if (tid == 0) vars[kVChunk] = atomicAdd(work_counter, 1u);
  __syncthreads();
  uint32_t chunk = wave::uniform(vars[kVChunk]);
  __syncthreads();  // (an empty chunk has no barrier before thread 0 publishes the next id)

  while (chunk < total_chunks) {
    // dequeue of the NEXT chunk: issued now, consumed after this chunk
    uint32_t next_chunk = 0;
    if (tid == 0) next_chunk = atomicAdd(work_counter, 1u);

    // chunk-major ids: every unit's first chunk, then every unit's second, ... so the short
    // last chunks of the units are handed out at the very end (smaller tail)
    const uint32_t q = wave::uniform(queries[chunk % n_queries].run_unit);
    const uint32_t tile0 = (chunk / n_queries) * kChunkTiles;
    const DevQuery qd = queries[q];
    const DevSegment seg = segs[qd.seg];
    const uint32_t n_tiles = qd.n_tiles;
    const uint32_t ntile = tile0 >= n_tiles ? 0u
                           : ((n_tiles - tile0) < kChunkTiles ? (n_tiles - tile0) : kChunkTiles);
    const DevTail* tails_q = tails + uint64_t(q) * jt;
    const uint32_t* first_q = first + qd.first_off;
    const uint32_t bs = bstar[q];
    const float fx_mul = qd.fx_mul;

    uint32_t pend_base = 0;   // thread 0: reserved candidate base of the previous tile (in flight)
    if (ntile) {   // (an empty chunk id of a shorter segment only runs the hand-over below)
    // ---- chunk prologue: everything that is per query / per chunk ----------
    if (tid < qd.n_terms) {
      const DevQTerm qt = qterms[qd.first_term + tid];
      sm.qts[tid] = qt;
      const DevTail* tl = tails_q + tid;
      TermC c;
      c.dir_off = tl->dir_off;
      c.dstart = uint32_t(tl->doc_start);
      c.tpack = term_pack(tid, qt.kind, qt.cache_id);
      c.cs = qt.c0 * fx_mul;
      c.tail_n = tl->n;
      c.pad[0] = c.pad[1] = 0;
      tc[tid] = c;
      tnblk[tid] = tl->nblk;
      tfirst[tid] = tl->first_doc;
      tlast[tid] = tl->last_doc;
    }
    for (uint32_t i = tid; i < (ntile + 1) * jt; i += blockDim.x) {
      const uint32_t c = i / jt, j = i % jt;
      rows[c * kMaxTerms + j] = first_q[uint64_t(tile0 + c) * jt + j];
    }
    __syncthreads();
    build_tables(sm, qd.n_caches, qd.n_terms);
    // per tile: exclusive prefix sums of each term's block count, and the set of
    // terms whose decoded tail reaches into the tile
    if (tid < ntile) {
      uint32_t* to = toff + tid * kToffStride;
      const uint32_t tlo = kDocMin + (tile0 + tid) * uint32_t(TILE);
      uint32_t off = 0, mask = 0;
      for (uint32_t j = 0; j < qd.n_terms; ++j) {
        to[j] = off;
        const uint32_t b0 = rows[tid * kMaxTerms + j];
        uint32_t b1 = rows[(tid + 1) * kMaxTerms + j] + 1u;
        b1 = b1 < tnblk[j] ? b1 : tnblk[j];
        off += b1 > b0 ? b1 - b0 : 0u;
        // (tiles past the end of the segment hold no docs: no need to clip to span)
        if (tc[j].tail_n && tfirst[j] < tlo + uint32_t(TILE) && tlast[j] >= tlo) mask |= 1u << j;
      }
      to[qd.n_terms] = off;
      for (uint32_t j = qd.n_terms + 1; j <= kMaxTerms; ++j) to[j] = 0xFFFFFFFFu;
      to[kMaxTerms + 1] = off;
      tmask[tid] = mask;
    }
    const ACC thr = bin_threshold<ACC>(bs, qd);
    __syncthreads();

    // Requests the directory entry of item `skip + ftid` of local tile c. The loaded
    // words stay raw (x_off, x_last, x_bits); store_item() combines them later.
    // Items are handed out from the LAST thread down: the tables are filled by the
    // last wavefront(s), which have the fewest work items of the tile (wavefront w
    // decodes items w, w+nw, ...), so this duty evens the wavefronts out.
    const uint32_t ftid = blockDim.x - 1u - tid;
    auto fetch_item = [&](uint32_t c, uint32_t skip, uint32_t& x_off, uint32_t& x_aoff,
                          uint32_t& x_last, uint32_t& x_bits, uint32_t& x_meta,
                          uint32_t& x_dstart, float& x_cs) {
      const uint32_t* to = toff + c * kToffStride;
      const uint32_t id = skip + ftid;
      if (ftid < kItemChunk && id < to[kMaxTerms + 1]) {
        // term slot j: to[j] <= id < to[j+1]  <=>  j = #{t >= 1 : to[t] <= id}; all the
        // reads are issued together (one LDS round trip, no search loop)
        uint32_t j = 0;
#pragma unroll
        for (uint32_t t = 1; t <= 8; ++t) j += to[t] <= id ? 1u : 0u;
        if (qd.n_terms > 8) {
#pragma unroll
          for (uint32_t t = 9; t <= kMaxTerms; ++t) j += to[t] <= id ? 1u : 0u;
        }
        const TermC T = tc[j];
        const uint32_t b = rows[c * kMaxTerms + j] + (id - to[j]);
        const uint64_t e = T.dir_off + b;
        x_dstart = T.dstart;
        x_cs = T.cs;
        x_meta = T.tpack | (b ? 0u : 0x20000000u);   // bit 29: first block of the term
        x_off = seg.blk_off[e];
        x_aoff = seg.blk_aoff[e];
        x_last = seg.blk_last[e - (b ? 1u : 0u)];
        x_bits = seg.blk_bits[e];
      }
    };
    auto store_item = [&](ItemL* dst, uint32_t n, uint32_t x_off, uint32_t x_aoff,
                          uint32_t x_last, uint32_t x_bits, uint32_t x_meta, uint32_t x_dstart,
                          float x_cs) {
      if (ftid < kItemChunk && ftid < n) {
        ItemL I;
        I.base = (x_meta & 0x20000000u) ? kDocMin : x_last;
        I.pack = item_pack(x_bits, x_meta);
        I.off = (I.pack >> 31) ? x_aoff : x_dstart + x_off;
        I.cs = x_cs;
        dst[ftid] = I;
      }
    };
    // kNormPieces 8-byte pieces of the tile's norm bytes per thread (the host sizes
    // workgroups to >= TILE / (8 * kNormPieces) threads)
    constexpr int kNormPieces = TILE > 8192 ? 3 : 2;
    struct NormRegs { uint64_t w[kNormPieces]; };
    auto norm_load = [&](uint32_t tile, NormRegs& r) {
#pragma unroll
      for (int e = 0; e < kNormPieces; ++e) r.w[e] = 0;
      if (seg.norms && seg.norm_width == 1) {
        const uint64_t base = uint64_t(tile) * TILE + (kDocMin - seg.norm_min_doc);
        const uint32_t i0 = tid * (8u * kNormPieces);
        if (i0 < uint32_t(TILE) && base + i0 < seg.norm_count) {
#pragma unroll
          for (int e = 0; e < kNormPieces; ++e) r.w[e] = wave::load_u64(seg.norms + base + i0 + 8 * e);
        }
      }
    };
    auto norm_store = [&](const NormRegs& r) {
      const uint32_t i0 = tid * (8u * kNormPieces);
      if (i0 < uint32_t(TILE)) {
        uint64_t* d = reinterpret_cast<uint64_t*>(sm.lnorm + i0);
#pragma unroll
        for (int e = 0; e < kNormPieces; ++e) d[e] = r.w[e];
      }
    };

    // ---- prime the pipeline: tile 0 synchronously, requests for tile 1 --------
    uint32_t x_off = 0, x_aoff = 0, x_last = 0, x_bits = 0, x_meta = 0, x_dstart = 0;
    float x_cs = 0.f;
    NormRegs nrm;
    uint32_t n_cur = toff[kMaxTerms + 1], n_next = 0;
    fetch_item(0, 0, x_off, x_aoff, x_last, x_bits, x_meta, x_dstart, x_cs);
    norm_load(tile0, nrm);
    store_item(items_of(0), n_cur, x_off, x_aoff, x_last, x_bits, x_meta, x_dstart, x_cs);
    norm_store(nrm);
    __syncthreads();
    if (1 < ntile) {
      n_next = toff[kToffStride + kMaxTerms + 1];
      fetch_item(1, 0, x_off, x_aoff, x_last, x_bits, x_meta, x_dstart, x_cs);
      norm_load(tile0 + 1, nrm);
    }
    ItemRegs R;
    items_prepare<LAYOUT>(seg, items_of(0), n_cur < kItemChunk ? n_cur : kItemChunk, inv_nw, R);

    for (uint32_t u = 0; u < ntile; ++u) {
      const uint32_t tile = tile0 + u;
      const uint32_t lo = kDocMin + tile * TILE;
      const uint32_t span = (seg.num_docs + kDocMin - lo) < uint32_t(TILE)
                              ? (seg.num_docs + kDocMin - lo) : uint32_t(TILE);
      const bool has_next = u + 1 < ntile;
      // compute of tile u: decode + score + accumulate
      items_run<ACC, LAYOUT, TILE, AND>(seg, sm, R, lo, span, fx_mul);
      for (uint32_t done = kItemChunk; done < n_cur; done += kItemChunk) {  // rare: > 256 items
        __syncthreads();
        uint32_t yo = 0, ya = 0, yl = 0, yb = 0, ym = 0, yd = 0;
        float yc = 0.f;
        fetch_item(u, done, yo, ya, yl, yb, ym, yd, yc);
        store_item(items_of(u), n_cur - done, yo, ya, yl, yb, ym, yd, yc);
        __syncthreads();
        const uint32_t n = (n_cur - done) < kItemChunk ? (n_cur - done) : kItemChunk;
        process_items<ACC, LAYOUT, TILE, AND>(seg, sm, items_of(u), n, lo, span, fx_mul);
      }
      const uint32_t tm = wave::uniform(tmask[u]);
      if (tm) {  // decoded vint tails / single docs reaching into this tile
        // rare path: the table pointers are re-read from the segment record here instead of
        // living in scalar registers across the whole tile loop (wave::opaque stops the
        // compiler from hoisting the loads)
        const DevSegment* sp = segs + wave::opaque(wave::uniform(qd.seg));
        const uint32_t* tdocs = sp->tail_docs;
        const uint32_t* tfreqs = sp->tail_freqs;
        for (uint32_t j = wv; j < qd.n_terms; j += nw) {
          if ((tm >> j) & 1u) {
            const DevQTerm qt = sm.qts[j];
            const float inv_one = 1.f / (qt.norm_const + qt.norm_length * 1.f);
            const uint32_t row = tails_q[j].tail_row;
            const uint32_t tn = tc[j].tail_n;
            for (uint32_t i = lane; i < tn; i += 64)
              tile_apply<ACC, TILE, AND>(seg, sm, qt, inv_one, tdocs[row + i], tfreqs[row + i], lo,
                                         span, fx_mul);
          }
        }
      }
      // the directory entries of tile u+1 (requested a tile ago) land in the other table
      if (has_next)
        store_item(items_of(u + 1u), n_next, x_off, x_aoff, x_last, x_bits, x_meta, x_dstart, x_cs);
#ifndef IRS_ABL_NOBAR1   // timing experiment only
      __syncthreads();  // B1: every accumulation of tile u has landed; items of u+1 visible
#endif

      uint32_t n_next2 = 0;
      if (has_next) {
        norm_store(nrm);  // norms of tile u+1 (tile u no longer reads them)
        if (u + 2 < ntile) {   // requests for tile u+2
          n_next2 = toff[(u + 2u) * kToffStride + kMaxTerms + 1];
          fetch_item(u + 2u, 0, x_off, x_aoff, x_last, x_bits, x_meta, x_dstart, x_cs);
          norm_load(tile + 2u, nrm);
        }
        // this wavefront's items of tile u+1 and the payload of the first two
        items_prepare<LAYOUT>(seg, items_of(u + 1u), n_next < kItemChunk ? n_next : kItemChunk,
                              inv_nw, R);
      }

      // epilogue of tile u: read + clear the accumulators, count hits, stage candidates
      uint64_t* lc = lcand + (u & 1u) * kScoreCands;
      uint32_t* ncand = vars + kVNc0 + (u % 3u);
      auto candidate = [&](uint32_t i, ACC a) {
        const float v = from_fixed<ACC>(a, qd.fx_inv);
        if (score_bin(v, qd.bin_scale) >= bs) {
          const uint64_t key = make_key(v, lo + i);
          const uint32_t slot = atomicAdd(ncand, 1u);
          if (slot < kScoreCands) {
            lc[slot] = key;
          } else {  // rare: more candidates in one tile than staging slots
            const uint32_t g = atomicAdd(&cand_count[q], 1u);
            if (g < cand_cap) cands[uint64_t(q) * cand_cap + g] = key;
          }
        }
      };
      {
        const bool is_and = AND && (qd.op & 0xFF) == 1;  // op = 1 | required matches << 8
        const uint32_t need = uint32_t(qd.op >> 8);
        // eight accumulators per lane per step: two 4-wide LDS reads in flight, two
        // wide clears; hits are counted per wavefront with ballots (SALU adds)
        uint32_t wave_hits = 0;
        const uint32_t step = blockDim.x * 4u;
        for (uint32_t i = tid * 4u; i < uint32_t(TILE); i += 2u * step) {
          const bool two = i + step < uint32_t(TILE);  // same for the whole workgroup
          const uint32_t i2 = two ? i + step : i;
          ACC a[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = sm.acc[i + e];
#pragma unroll
          for (int e = 0; e < 4; ++e) a[4 + e] = sm.acc[i2 + e];
#pragma unroll
          for (int e = 0; e < 8; ++e) wave::keep_acc(a[e]);
          uint32_t cw0 = 0, cw1 = 0;   // match counters of docs i..i+3 / i2..i2+3, a byte each
          if (AND) {
            cw0 = sm.cnt[i >> 2];
            cw1 = sm.cnt[i2 >> 2];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) sm.acc[i + e] = ACC(0);
#pragma unroll
          for (int e = 0; e < 4; ++e) sm.acc[i2 + e] = ACC(0);
          if (AND) {   // this thread is the only reader of those counter words: clear them here
            sm.cnt[i >> 2] = 0u;
            sm.cnt[i2 >> 2] = 0u;
          }
          if (!two) {
#pragma unroll
            for (int e = 4; e < 8; ++e) a[e] = ACC(0);
          }
          if (AND && is_and) {
            // AND / min-match: a doc counts only with >= `need` matching terms; the others
            // are made to look untouched
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const uint32_t c = ((e < 4 ? cw0 : cw1) >> (8u * (uint32_t(e) & 3u))) & 0xFFu;
              a[e] = c >= need ? a[e] : ACC(0);
            }
          }
          ACC top = a[0];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            wave_hits += uint32_t(__builtin_popcountll(wave::ballot(a[e] != ACC(0))));
            top = a[e] > top ? a[e] : top;
          }
          if (top >= thr) {  // rare: one copy of the candidate code, per-lane loop
            uint32_t cm = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) cm |= a[e] >= thr ? (1u << e) : 0u;
            while (cm) {
              const uint32_t e = uint32_t(__builtin_ctz(cm));
              cm &= cm - 1u;
              ACC x = a[0];
#pragma unroll
              for (int f = 1; f < 8; ++f) x = e == uint32_t(f) ? a[f] : x;
              candidate((e < 4u ? i : i2 - 4u) + e, x);
            }
          }
        }
        if (lane == 0 && wave_hits) atomicAdd(&vars[kVHits], wave_hits);
      }
      if (tid == 0) {
        vars[kVBase] = pend_base;                 // tile u-1's reservation has arrived by now
        vars[kVNc0 + ((u + 1u) % 3u)] = 0u;       // counter of tile u+1 (last used by tile u-2)
      }
#ifndef IRS_ABL_NOBAR2   // timing experiment only
      __syncthreads();  // B2: accumulators are clear again
#endif
      // flush tile u-1's staged candidates to its reserved global range
      if (u > 0) {
        const uint32_t pn_raw = vars[kVNc0 + ((u - 1u) % 3u)];
        const uint32_t pn = pn_raw < kScoreCands ? pn_raw : kScoreCands;
        const uint32_t gbase = vars[kVBase];
        const uint64_t* pl = lcand + ((u - 1u) & 1u) * kScoreCands;
        for (uint32_t i = tid; i < pn; i += blockDim.x) {
          const uint32_t g = gbase + i;
          if (g < cand_cap) cands[uint64_t(q) * cand_cap + g] = pl[i];
        }
      }
      // reserve global slots for tile u (returning atomic; consumed one tile later)
      if (tid == 0) {
        const uint32_t cn_raw = *ncand;
        const uint32_t cn = cn_raw < kScoreCands ? cn_raw : kScoreCands;
        pend_base = cn ? atomicAdd(&cand_count[q], cn) : 0u;
      }
      n_cur = n_next;
      n_next = n_next2;
    }
    }
    // ---- chunk epilogue: flush the last tile, publish hits, pick up the next chunk
    if (tid == 0) {
      vars[kVBaseLast] = pend_base;
      vars[kVChunk] = next_chunk;
    }
    __syncthreads();
    {
      const uint32_t lu = ntile ? ntile - 1u : 0u;
      const uint32_t pn_raw = vars[kVNc0 + (lu % 3u)];
      const uint32_t pn = pn_raw < kScoreCands ? pn_raw : kScoreCands;
      const uint32_t gbase = vars[kVBaseLast];
      const uint64_t* pl = lcand + (lu & 1u) * kScoreCands;
      for (uint32_t i = tid; i < pn; i += blockDim.x) {
        const uint32_t g = gbase + i;
        if (g < cand_cap) cands[uint64_t(q) * cand_cap + g] = pl[i];
      }
    }
    chunk = wave::uniform(vars[kVChunk]);
    __syncthreads();  // everyone has read the chunk id and the staging buffers
    if (tid == 0) {
      if (vars[kVHits]) atomicAdd(&hits[q], (unsigned long long)vars[kVHits]);
      vars[kVHits] = 0u;
      vars[kVNc0] = vars[kVNc0 + 1] = vars[kVNc0 + 2] = 0u;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- select --

// in-LDS bitonic sort, descending, n a power of two
__device__ __forceinline__ void bitonic_desc(uint64_t* a, uint32_t n) {
  for (uint32_t size = 2; size <= n; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < n / 2; i += blockDim.x) {
        const uint32_t pos = 2u * i - (i & (stride - 1u));
        const uint32_t other = pos + stride;
        const bool desc = (pos & size) == 0;
        const uint64_t x = a[pos], y = a[other];
        if ((x < y) == desc) { a[pos] = y; a[other] = x; }
      }
    }
  }
  __syncthreads();
}

// One workgroup (kThreads) per query: exact top-k of the candidate keys.  Keys are
// unique ((score, doc) pairs), descending key order == (score desc, doc asc) — the
// deterministic refinement of the harness heap (index-search.cpp:745-787).
//   1. the candidates are staged in LDS once (up to `stage_cap`; beyond that the
//      passes re-read them from global memory);
//   2. if there are more than `sort_cap` of them, an MSB-first radix select (8 bits
//      per pass, histogram in LDS, bucket search by a 256-thread suffix scan) finds
//      the smallest key that still belongs to the top k; it stops at the first pass
//      whose bucket is needed entirely;
//   3. the survivors (exactly min(k, n): keys are unique) are sorted by a bitonic
//      network in LDS.
// Dynamic LDS: stage_cap + sort_cap keys; sort_cap = pow2ceil(k_max).
constexpr uint32_t kSelectStage = 8192;

__global__ void __launch_bounds__(kThreads)
k_select(const DevQuery* queries, const uint64_t* cands, uint32_t cand_cap,
         const uint32_t* cand_count, const unsigned long long* hits, Hit* out, uint32_t k_max,
         uint32_t* out_count, uint32_t* status, uint32_t stage_cap, uint32_t sort_cap) {
  RT_DYN_SMEM(smem);
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);         // [sort_cap]
  uint64_t* stage = keys + sort_cap;                          // [stage_cap]
  __shared__ uint32_t hist[256];
  __shared__ uint32_t wsum[kWaves];
  __shared__ uint64_t sh_prefix;
  __shared__ uint32_t sh_want, sh_n, sh_done;
  const uint32_t tid = threadIdx.x;
  const unsigned lane = tid & 63u, wv = tid >> 6;
  const uint32_t q = blockIdx.x;
  const DevQuery qd = queries[q];
  uint32_t n = cand_count[q];
  if (n > cand_cap) {
    if (tid == 0) atomicOr(status, kStatusOverflow);
    n = cand_cap;
  }
  // an estimated threshold (k_pilot) cut off docs that belong to the top k
  if (n < qd.k && hits[q] > n && tid == 0) atomicOr(status, kStatusUnderflow);
  const uint64_t* src = cands + uint64_t(q) * cand_cap;
  const uint32_t kk = qd.k < n ? qd.k : n;
  const bool staged = n <= stage_cap;
  uint32_t m = n;  // keys that end up in the sort region
  if (n <= sort_cap) {
    for (uint32_t i = tid; i < n; i += blockDim.x) keys[i] = src[i];
  } else {
    if (staged) {
      for (uint32_t i = tid; i < n; i += blockDim.x) stage[i] = src[i];
    }
    if (tid == 0) { sh_prefix = 0; sh_want = kk; sh_done = 0; }
    uint64_t kth = 0;
    for (int pass = 0; pass < 8; ++pass) {
      const int shift = 56 - 8 * pass;
      if (tid < 256) hist[tid] = 0u;
      __syncthreads();
      const uint64_t prefix = sh_prefix;
      const uint32_t want = sh_want;
      for (uint32_t i = tid; i < n; i += blockDim.x) {
        const uint64_t key = staged ? stage[i] : src[i];
        if (pass == 0 || (key >> (shift + 8)) == prefix)
          atomicAdd(&hist[uint32_t(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      // thread t looks at digit 255 - t: `before` = keys in larger digits
      const uint32_t d = 255u - tid;
      const uint32_t h = tid < 256 ? hist[d] : 0u;
      const uint32_t incl = wave::inclusive_scan(h);
      if (lane == 63 && wv < kWaves) wsum[wv] = incl;
      __syncthreads();
      if (tid < 256) {
        uint32_t before = incl - h;
        for (uint32_t w = 0; w < wv; ++w) before += wsum[w];
        if (before < want && want <= before + h) {  // exactly one thread
          sh_want = want - before;
          sh_prefix = (prefix << 8) | uint64_t(d);
          sh_done = (h == want - before) ? 1u : 0u;  // the whole bucket is needed
        }
      }
      __syncthreads();
      kth = sh_prefix << shift;  // smallest possible key of the bucket
      if (sh_done) break;
    }
    if (tid == 0) sh_n = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += blockDim.x) {
      const uint64_t key = staged ? stage[i] : src[i];
      if (key >= kth) {
        const uint32_t slot = atomicAdd(&sh_n, 1u);
        if (slot < sort_cap) keys[slot] = key;
      }
    }
    __syncthreads();
    m = sh_n < sort_cap ? sh_n : sort_cap;
  }
  uint32_t p2 = 1;
  while (p2 < m) p2 <<= 1;
  __syncthreads();
  for (uint32_t i = m + tid; i < p2; i += blockDim.x) keys[i] = 0;
  bitonic_desc(keys, p2);
  for (uint32_t i = tid; i < kk; i += blockDim.x)
    out[uint64_t(q) * k_max + i] = key_hit(keys[i]);
  if (tid == 0) out_count[q] = kk;
}

// ----------------------------------------------------------------- merge --

// Merge of per-segment top-k lists in the order (score desc, segment asc, doc asc)
// (tests/search/wand_test.cpp:72-86).  Every input list is already sorted, so the
// final position of element p of list l is
//   p + sum over the other lists m of #{x in m : x before the element},
// and "before" only needs the score: x.score > s, or x.score >= s when m's segment
// id is the smaller one.  One workgroup per query stages the score bits of all
// lists in LDS (positive floats order like their bit patterns) and every thread
// ranks its elements with binary searches, giving up as soon as the rank
// reaches k.  No sort, no barriers after the staging.
constexpr uint32_t kMergeMax = 32768;   // n_lists * k score words staged in LDS (128 KB)
constexpr uint32_t kMergeLists = 16;

struct MergeLists {
  const Hit* hits[kMergeLists];
  const uint32_t* counts[kMergeLists];
  uint32_t seg_ids[kMergeLists];
};

constexpr uint32_t merge_smem_bytes(uint32_t n_lists, uint32_t k) {
  return 4u * n_lists * k + 4u * kMergeLists * 2u + 8u * kMergeLists + 8u;
}

__global__ void __launch_bounds__(kThreads)
k_merge_topk(MergeLists lists, uint32_t n_lists, uint32_t k, Hit* out, uint32_t* out_seg,
             uint32_t* out_counts) {
  RT_DYN_SMEM(smem);
  uint32_t* sc = reinterpret_cast<uint32_t*>(smem);   // [n_lists][k], descending
  uint32_t* cnt = sc + n_lists * k;                     // [kMergeLists]
  uint32_t* sid = cnt + kMergeLists;                    // [kMergeLists]
  // the lists' base pointers, so that a thread can fetch its element without a
  // 16-way select over kernel arguments (8-byte aligned: 4*n_lists*k + 128 bytes in)
  const Hit** hp = reinterpret_cast<const Hit**>(sid + kMergeLists + ((n_lists * k) & 1u));
  const uint32_t q = blockIdx.x;
  if (threadIdx.x < kMergeLists) {
    uint32_t c = 0, id = 0;
    // (kernel-argument arrays are only indexed by unrolled constants)
#pragma unroll
    for (uint32_t l = 0; l < kMergeLists; ++l) {
      if (l == threadIdx.x && l < n_lists) {
        c = lists.counts[l][q];
        id = lists.seg_ids[l];
      }
    }
    cnt[threadIdx.x] = c < k ? c : k;
    sid[threadIdx.x] = id;
    const Hit* base = nullptr;
#pragma unroll
    for (uint32_t l = 0; l < kMergeLists; ++l) {
      if (l == threadIdx.x && l < n_lists) base = lists.hits[l] + uint64_t(q) * k;
    }
    hp[threadIdx.x] = base;
  }
  __syncthreads();
#pragma unroll
  for (uint32_t l = 0; l < kMergeLists; ++l) {
    if (l < n_lists) {
      const Hit* src = lists.hits[l] + uint64_t(q) * k;
      const uint32_t c = cnt[l];
      for (uint32_t r = threadIdx.x; r < c; r += blockDim.x) {
        const float f = src[r].score;
        uint32_t bits;
        __builtin_memcpy(&bits, &f, 4);
        sc[l * k + r] = bits;
      }
    }
  }
  __syncthreads();
  uint32_t total = 0;
  for (uint32_t l = 0; l < n_lists; ++l) total += cnt[l];
  // Cheap lower bound of the final k-th score: with g = ceil(k / n_lists), if every list
  // holds at least g elements then the union holds >= k elements scoring at least
  // min_l score_l[g - 1], so nothing below that can be in the top k.  Most elements are
  // dismissed by this one comparison; only about k of them are ranked.
  const uint32_t g = (k + n_lists - 1) / n_lists;
  uint32_t floor_bits = 0xFFFFFFFFu;
  for (uint32_t l = 0; l < n_lists; ++l)
    floor_bits = cnt[l] >= g ? (sc[l * k + g - 1] < floor_bits ? sc[l * k + g - 1] : floor_bits) : 0u;
  // (a list shorter than g voids the bound: floor 0 keeps everything; the loop above lets
  // a later list raise it again only through `<`, so once 0 it stays 0)
  // element i -> (list i % n_lists, position i / n_lists): a wavefront works at one depth
  for (uint32_t i = threadIdx.x; i < n_lists * k; i += blockDim.x) {
    const uint32_t l = i % n_lists, p = i / n_lists;
    if (p >= cnt[l]) continue;
    const uint32_t s = sc[l * k + p], my_id = sid[l];
    if (s < floor_bits) continue;
    uint32_t rank = p;
    for (uint32_t m = 0; m < n_lists && rank < k; ++m) {
      if (m == l) continue;
      const bool ties_first = sid[m] < my_id;
      const uint32_t* v = sc + m * k;
      uint32_t lo = 0, hi = cnt[m];
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t x = v[mid];
        if (ties_first ? x >= s : x > s) lo = mid + 1u; else hi = mid;
      }
      rank += lo;
    }
    if (rank < k) {
      out[uint64_t(q) * k + rank] = hp[l][p];
      out_seg[uint64_t(q) * k + rank] = my_id;
    }
  }
  if (threadIdx.x == 0) out_counts[q] = total < k ? total : k;
}

}  // namespace irs_hip
