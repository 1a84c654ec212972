// conj.h — irs::And as a block-driven conjunction, with block-max (WAND) pruning.
//
// Reference: Conjunction::converge (conjunction.hpp:207-223: leap-frog on cost-sorted
// iterators, the cheapest leads, the others seek() through their skip lists);
// BlockConjunction (:230-433: aligns skip "leaf" blocks and prunes by sum of block-max
// scores <= threshold, :380-426); wanderator (formats_10.cpp:2424-2824: per skip entry the
// SAME score function on the block's (max freq, min norm), blocks whose bound is <= the
// threshold are skipped :2521-2528); the per-block (max freq, min norm) the index carries
// for it: FreqNormProducer (wand_writer.hpp:152-342).
//
// Here ONE WAVEFRONT owns one 128-posting block (or the decoded vint tail) of the query's
// rarest term (the lead; MakeConjunction sorts by cost, conjunction.hpp:450-453):
//   0. WAND mode: bound = block-max score of the lead block + for every other term the
//      largest block-max score among its blocks overlapping the lead block's doc range; a
//      bound below the threshold bin (from the pilot pass) skips the lead block — nothing is
//      decoded;
//   1. the lead block is decoded: 128 ascending docs and their frequencies into LDS;
//   2. for every other term, cheapest first: binary search of the block directory for the
//      first block reaching the lead block's first doc, then 64 directory entries per step, a
//      lane each, tested against the docs still alive ("is any of them in (previous last,
//      last]"); only blocks that pass are decoded, and each decoded posting looks its doc up
//      among the lead docs (binary search in LDS): a hit leaves the term's frequency in the
//      term's row.  A term that leaves no doc alive ends the block;
//   3. docs every term reached are the conjunction.  Only THEY are scored — as the reference
//      scores a doc after converge() has accepted it (conjunction.hpp:176-187): their norm is
//      read, the per-term scores are summed in cost order (the order the reference's
//      Conjunction sums them); docs at or above the threshold bin become candidates for
//      k_select.  (Frequencies are kept for kConjRows terms at a time: a longer conjunction
//      scores the docs still alive every kConjRows terms.)
// The same kernel run over every P-th lead block with `pilot` set histograms the scores of
// the matches instead (k_conj_threshold turns the histogram into the threshold bin).
#pragma once
#include "gpu_rt.h"
#include "phrase.h"
#include "score.h"

namespace irs_hip {

// ---------------------------------------------------------------- block max --

// Per full block: largest frequency and smallest non-zero norm value of its docs — what
// FreqNormProducer keeps per skip entry (wand_writer.hpp:170-209), recomputed from the
// postings so that it exists for every block of every index (the skip data has no entry for
// a list's last block).  Work split by directory row, as k_pack_payloads.
template<int LAYOUT>
__global__ void __launch_bounds__(kThreads)
k_block_max(DevSegment seg, uint64_t rows, uint32_t* blk_maxf, uint32_t* blk_minn) {
  const unsigned lane = threadIdx.x & 63u;
  IRS_FOR_ROWS(e, rows) {
    const uint32_t bits = seg.blk_bits[e];
    const uint32_t base = seg.blk_dir[e].prev_last;
    uint32_t d0, d1, f0, f1;
    decode_block<LAYOUT, true>(row_block(seg, e), bits & 0xFFu, bits >> 8,
                               base, lane, d0, d1, f0, f1);
    uint32_t mf = f0 > f1 ? f0 : f1;
    uint32_t mn = 0xFFFFFFFFu;
    if (seg.norms && !seg.norm_legacy) {
      const uint32_t n0 = seg.norm_width == 1 ? seg.norms[d0 - seg.norm_min_doc] : norm_global(seg, d0);
      const uint32_t n1 = seg.norm_width == 1 ? seg.norms[d1 - seg.norm_min_doc] : norm_global(seg, d1);
      if (n0) mn = n0;
      if (n1 && n1 < mn) mn = n1;
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      const uint32_t of = __shfl_xor(mf, s, 64), on = __shfl_xor(mn, s, 64);
      mf = of > mf ? of : mf;
      mn = on < mn ? on : mn;
    }
    if (lane == 0) {
      blk_maxf[e] = mf;
      blk_minn[e] = mn == 0xFFFFFFFFu ? 1u : mn;   // no norm column: norm == 1 (bm25.cpp:487-489)
    }
  }
}

// The index's OWN block-max data, where the field was indexed with scorers: one wavefront per term
// walks level 0 of the term's skip list (SkipWriter layout: per level a vlong length + bytes,
// level 0 last; formats_10.cpp:501-533, 1063-1080) and takes, for every entry, the payload of
// scorer 0 — FreqNormSource::Read (wand_writer.hpp:318-334): vint(max freq) [+ vint(norm -
// freq)] — as the block's (max freq, min norm), overwriting what k_block_max derived from the
// postings.  The index stores max(min norm, max freq) as the norm (FreqNormProducer,
// wand_writer.hpp:198-209; a valid bound because a doc's norm is never below its frequency);
// the last block of a list has no entry and keeps the derived pair.  `taken` counts the
// entries read.
// The header (root entry, level count, the upper levels' lengths) is read where it lies.  Level 0
// — one entry per full block but the last, a chain of LEB128 fields — goes through LDS windows:
// the entries of a window are found by pointer doubling (chain_orbit, kernels.h: "if an entry
// starts at byte p, the next one starts at link(p)" for every p at once), then every entry is
// parsed, checked against the block directory and written by a thread of its own.  (A thread per
// term reading global bytes took 87 ms for the 48 k entries of a 6.25 M-doc segment's longest
// list; a lone wavefront walking an LDS window 34 ms.)
struct skip_link {
  const uint8_t* win;
  uint32_t vlongs;       // per entry: last doc, pointer delta [, pend_pos, pos pointer delta]
  uint32_t wand_count;   // then one size byte per scorer and the payloads
  uint32_t stop;         // level 0 ends here (window offset; beyond the window: anything larger)
  __device__ __forceinline__ uint32_t operator()(uint32_t p) const {
    if (p >= stop) return kLinkNone;   // (not an entry: the chain ends in front of it)
    uint32_t o = p;
    for (uint32_t f = 0; f < vlongs; ++f) {
      uint32_t n = 0;
      while (win[o++] & 0x80u) {
        // (the 64 readable bytes behind the window are as far as a field may run: the caller
        // re-stages from any entry that starts in the window's last kSkipHead bytes)
        if (++n == 10u) return kLinkBad;
      }
    }
    uint32_t total = 0;
    for (uint32_t w = 0; w < wand_count; ++w) total += win[o++];
    return o + total;
  }
};

constexpr uint32_t kSkipHead = 4u * 10u + 16u;   // vlongs + size bytes of an entry

__global__ void __launch_bounds__(kChainThreads)
k_wand_skip0(DevSegment seg, const uint64_t* skip_at /*[term] absolute offset of the skip data,
             0 = the list has none*/, uint32_t has_pos, uint32_t* blk_maxf, uint32_t* blk_minn,
             unsigned long long* taken, uint32_t* status) {
  __shared__ __attribute__((aligned(16))) uint8_t win[kChainWindow + 64];
  __shared__ ChainTables s_chain;
  __shared__ uint32_t s_at[kChainCap];
  __shared__ uint64_t s_cur, s_stop;
  __shared__ uint32_t s_flag;
  const uint32_t tid = threadIdx.x;
  const uint32_t term = blockIdx.x;
  if (term >= seg.num_terms) return;
  const DevTerm t = seg.terms[term];
  const uint64_t at = skip_at[term];
  if (!at || !t.nblk) return;
  if (tid == 0) {
    uint32_t bad = 0;
    uint64_t cur = 0, stop = 0;
    const uint8_t* p = seg.doc + at;
    const uint8_t* end = seg.doc + seg.doc_len;   // (the staged copy is zero padded behind it)
    auto vlong = [&]() {
      uint64_t v = 0;
      for (uint32_t sh = 0; sh < 64 && p < end; sh += 7) {
        const uint8_t b = *p++;
        v |= uint64_t(b & 0x7Fu) << sh;
        if (!(b & 0x80u)) return v;
      }
      bad = 1;
      return v;
    };
    // the root entry in front of the level count: one size byte per scorer, then the payloads
    // (CommonSkipWandData :1962-1979)
    uint64_t total = 0;
    for (uint32_t w = 0; w < seg.wand_count && p < end; ++w) total += *p++;
    p += total;
    const uint32_t levels = p < end ? uint32_t(vlong()) : 0u;
    if (!levels || bad) {
      bad = 2;   // nothing to take (not an error)
    } else {
      for (uint32_t l = levels; l-- > 1 && !bad;) {   // levels n..1
        const uint64_t len = vlong();
        if (!len || uint64_t(end - p) < len) bad = 1; else p += len;
      }
      const uint64_t len0 = bad ? 0 : vlong();
      if (bad || !len0 || uint64_t(end - p) < len0) {
        bad = 1;
      } else {
        cur = uint64_t(p - seg.doc);
        stop = cur + len0;
      }
    }
    s_cur = cur;
    s_stop = stop;
    s_flag = bad;
  }
  __syncthreads();
  if (s_flag) {
    if (tid == 0 && s_flag == 1) atomicOr(status, kStatusCorrupt);
    return;
  }
  uint64_t cur = s_cur;
  const uint64_t stop = s_stop;
  const uint64_t staged = seg.doc_len + kPadBytes;
  const uint32_t vlongs = has_pos ? 4u : 2u;
  uint32_t n = 0, bad = 0, framing = 0;
  __syncthreads();
  while (cur < stop && n < t.nblk) {
    const uint64_t win_lo = cur & ~uint64_t(15);
    uint64_t bytes = staged - win_lo;
    if (bytes > kChainWindow) bytes = kChainWindow;
    bytes &= ~uint64_t(15);
    for (uint32_t o = tid * 16u; o < bytes; o += kChainThreads * 16u)
      *reinterpret_cast<ChainLine*>(win + o) = *reinterpret_cast<const ChainLine*>(seg.doc + win_lo + o);
    if (tid == 0) s_flag = 0;
    __syncthreads();
    // entries that start in the window's last kSkipHead bytes are the next window's: their
    // fields may run past the staged bytes (an entry is at most 4136 bytes: the first always fits)
    const uint32_t lim = uint32_t(bytes);
    const uint64_t stop_o = stop - win_lo;
    const uint32_t head_lim = lim == kChainWindow ? lim - kSkipHead : lim;
    const uint32_t stop_w = stop_o < head_lim ? uint32_t(stop_o) : head_lim;
    const uint32_t left = t.nblk - n;
    uint32_t next;
    const uint32_t m = chain_orbit(win, s_chain, uint32_t(cur - win_lo), lim, seg.doc_len - win_lo,
                                   left < kChainCap ? left : kChainCap, s_at, &next, &bad,
                                   skip_link{win, vlongs, seg.wand_count, stop_w});
    // every entry by a thread of its own: its last doc must be the directory's — anything else
    // means the entries are framed differently from what this walk assumes (the caller then
    // keeps the derived pairs) — and the payload of scorer 0 is the block's pair
    for (uint32_t i = tid; i < m; i += kChainThreads) {
      uint32_t o = s_at[i] >> 8;
      auto vlong = [&]() {
        uint64_t v = 0;
        for (uint32_t sh = 0; sh < 64; sh += 7) {
          const uint32_t b = win[o++];
          v |= uint64_t(b & 0x7Fu) << sh;
          if (!(b & 0x80u)) break;
        }
        return v;
      };
      const uint32_t last = uint32_t(vlong());
      for (uint32_t f = 1; f < vlongs; ++f) (void)vlong();   // pointer delta [, pend_pos, pos pointer delta]
      const uint32_t s0 = seg.wand_count ? win[o] : 0u;
      o += seg.wand_count;
      const uint64_t e = t.dir_off + n + i;
      if (last != seg.blk_last[e]) {
        s_flag = 1;
      } else if (s0) {
        const uint32_t payload = o;
        const uint32_t f = uint32_t(vlong());
        // (no more bytes: norm == freq — what a frequency-only payload means to a scorer that
        // wants a norm, "compatibility between BM25 in the index and TFIDF in the query")
        const uint32_t nrm = (o - payload) != s0 ? f + uint32_t(vlong()) : f;
        blk_maxf[e] = f;
        blk_minn[e] = nrm;
      }
    }
    __syncthreads();
    if (s_flag) { framing = 1; break; }   // (before `bad`: a walk that is off the framing may
    if (bad) break;                       //  run into anything)
    n += m;
    cur = win_lo + next;
    if (m == 0) break;   // (an entry that does not fit a window of its own: not from a writer)
    __syncthreads();     // the window and the lists are rewritten next
  }
  if (tid == 0) {
    if (framing) atomicOr(status, kStatusWandFraming);
    else if (bad) atomicOr(status, kStatusCorrupt);
    else if (n) atomicAdd(taken, static_cast<unsigned long long>(n));
  }
}

// Doc block `e` of a term, from the packed image when both parts live there (one funnel
// shift + one bit-field extract per value), else from `.doc` (any framing).
template<int LAYOUT>
__device__ __forceinline__ void decode_dir_block(const DevSegment& seg, uint64_t doc_start,
                                                 uint32_t bits, uint32_t off, uint32_t aoff,
                                                 uint32_t base, unsigned lane, uint32_t& d0,
                                                 uint32_t& d1, uint32_t& f0, uint32_t& f1) {
  const uint32_t dbits = bits & 0xFFu, fbits = bits >> 8;
  if (pk_both(dbits, fbits)) {
    const uint64_t pl = reinterpret_cast<uint64_t>(seg.pk) + (uint64_t(aoff) << 4);
    uint64_t da, db, fa, fb;
    raw_load_packed_g<LAYOUT>(pl, dbits, lane, da, db);
    raw_load_packed_g<LAYOUT>(pl + 16u * dbits, fbits, lane, fa, fb);
    uint32_t x0, x1;
    extract_fast<LAYOUT>(da, db, dbits, lane, x0, x1);
    extract_fast<LAYOUT>(fa, fb, fbits, lane, f0, f1);
    d1 = base + wave::inclusive_scan(x0 + x1);
    d0 = d1 - x1;
  } else {
    decode_block<LAYOUT, true>(seg.doc + doc_start + off, dbits, fbits, base, lane, d0, d1, f0, f1);
  }
}

// Conjunction::Score2 / ScoreN (conjunction.hpp:105-126): the first sub-score, then the
// filter's merger over the others (SumMerger / MaxMerger / MinMerger, scorer.hpp:390-423).
__device__ __forceinline__ float merge_scores(uint32_t merge, bool first, float acc, float s) {
  if (first) return acc + s;   // (acc is 0: the same float the reference starts from)
  if (merge == kScoreMax) return acc < s ? s : acc;
  if (merge == kScoreMin) return s < acc ? s : acc;
  return acc + s;
}

constexpr uint32_t kConjWaves = 4;  // wavefronts (= lead blocks) per workgroup
constexpr uint32_t kConjRows = 4;   // frequency rows per wavefront (terms scored per flush)

// LDS of one wavefront.  The lead block's doc range [dlo, dhi] is cut into kConjBuckets buckets of
// 2^s docs (s = 0 when the range is that small).  Per bucket: the entry index of its FIRST lead
// doc (`first`, 0 = none) — "is this decoded posting's doc a lead doc, and which" is one byte
// read, a walk over the bucket's lead docs (1.1 on average) and a compare, where a search of the
// sorted docs walks seven dependent reads (round 6: the search ran for ~every decoded block of a
// sparse lead, a third of the kernel's instructions); "has every earlier term reached it" is
// cnt[].  And one BIT per bucket in doc-range bitmaps: "does this block of the other term hold a
// doc still alive" is two prefix-count reads for a directory entry.
struct ConjWave {
  uint32_t docs[kBlock];
  float score[kBlock];
  uint32_t fr[kConjRows][kBlock];       // frequencies of the current terms
  alignas(16) uint8_t first[kConjBuckets];   // bucket -> 1 + entry index of its first lead doc
  uint32_t bm[3][kConjWords + 4];       // bitmaps: [0] the lead docs, [1] / [2] alternately the
                                        // docs the current term reached (= alive for the next)
  uint8_t lpre[kConjWords + 4];         // lead-bitmap bits in the words before word w
  uint8_t apre[kConjWords + 4];         // the same for the alive bitmap of the current term
  uint8_t cnt[kBlock];                  // terms that reached the doc so far
};

// 8 wavefronts per SIMD (a 64-VGPR budget): the kernel is a chain of short dependent steps,
// resident wavefronts count for more than registers
#define IRS_CONJ_ATTR RT_WAVES_PER_SIMD(8)
template<int LAYOUT>
__global__ void __launch_bounds__(kConjWaves * 64) IRS_CONJ_ATTR
k_conj(ConjArgs A, uint32_t pilot) {
  __shared__ ConjWave s_wave[kConjWaves];
  const uint32_t tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const uint32_t wv = wave::uniform(tid >> 6);
  // one lead item per wavefront: record e (pilot pass: the sampled items' records)
  uint32_t e = blockIdx.x * kConjWaves + wv;
  if (pilot) {
    if (e >= A.n_pilot) return;
    const PhraseWg w = A.wgs[e];
    e = wave::uniform(A.unit_items[w.unit] + w.first_item);
  } else if (e >= A.n_items) {
    return;
  }
  // (wave-uniform records are read with scalar loads, at the point of use)
  const ConjItem R = wave::sload<ConjItem>(reinterpret_cast<uint64_t>(A.recs) + uint64_t(e) * sizeof(ConjItem));
  const uint32_t unit = R.unit, item = R.item & kConjItemBlock;
  // (a piece of a lead block: postings [piece * (n >> lg) ...) of it, ConjItem)
  const uint32_t split_lg = R.item >> 28, piece = (R.item >> 24) & 15u;
  const DevQuery qd = wave::sload<DevQuery>(reinterpret_cast<uint64_t>(A.queries) + uint64_t(unit) * sizeof(DevQuery));
  const uint32_t m = qd.n_terms;
  if (m == 0) return;
  const uint32_t mrg = query_merge(qd.op);
  const DevSegment& seg = A.segs[qd.seg];   // (read field by field)
  const uint64_t tl_at = reinterpret_cast<uint64_t>(A.tails) + uint64_t(unit) * A.jt * sizeof(DevTail);
  const uint64_t qt_at = reinterpret_cast<uint64_t>(A.qterms) + uint64_t(qd.first_term) * sizeof(DevQTerm);
  auto term_tail = [&](uint32_t i) { return wave::sload<DevTail>(tl_at + i * sizeof(DevTail)); };
  auto term_q = [&](uint32_t i) { return wave::sload<DevQTerm>(qt_at + i * sizeof(DevQTerm)); };
  const DevTail ld = term_tail(0);   // the host sorted the terms by cost: the cheapest leads
  const uint32_t n_items = ld.nblk + (ld.n ? 1u : 0u);
  const uint32_t bs = pilot ? 0u : A.bstar[unit];
  ConjWave& W = s_wave[wv];
  uint32_t* docs = W.docs;
  float* score = W.score;
  uint8_t* cnt = W.cnt;
  const uint32_t* seek = A.seek + uint64_t(e) * (A.jt - 1u);
  const uint32_t r_lo = R.r_lo, r_hi = R.r_hi;   // the lead block's docs lie in [r_lo, r_hi]

  // ---- 0. block-max bound of every doc of this lead block (WandContext: ExecutionContext::wand)
  if (A.wand && bs) {
    float bound = 0.f;
    for (uint32_t i = 0; i < m; ++i) {
      const DevTail tl = term_tail(i);
      const DevQTerm qt = term_q(i);
      float ub = 0.f;
      if (i == 0 && item < ld.nblk) {
        ub = block_bound(qt, seg.blk_maxf[ld.dir_off + item], seg.blk_minn[ld.dir_off + item]);
      } else if (i == 0) {
        ub = term_bound(qt, seg.terms[tl.term].tf_bound);
      } else {
        // blocks of term i overlapping [r_lo, r_hi]: rows [a, z): from the seek table (the next
        // lead item starts behind r_hi, so its first row bounds this item's last one)
        const uint32_t a = seek[i - 1u];
        // (the pieces of one lead block share its seek row: the next BLOCK's is the bound)
        uint32_t z = item + 1u < n_items ? seek[((1u << split_lg) - piece) * (A.jt - 1u) + i - 1u] + 1u : tl.nblk;
        z = z < tl.nblk ? z : tl.nblk;
        for (uint32_t k = a + lane; k < z; k += 64) {
          const float sc = block_bound(qt, seg.blk_maxf[tl.dir_off + k], seg.blk_minn[tl.dir_off + k]);
          ub = sc > ub ? sc : ub;
        }
        // the decoded tail (no block-max entry): the term's global bound
        if (tl.n && tl.first_doc <= r_hi && tl.last_doc >= r_lo) {
          const float sc = term_bound(qt, seg.terms[tl.term].tf_bound);
          ub = sc > ub ? sc : ub;
        }
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) {
          const float o = __shfl_xor(ub, sh, 64);
          ub = o > ub ? o : ub;
        }
      }
      bound += ub;   // the same order the scores are summed in
    }
    // every doc of the block scores <= bound: below the threshold bin none can be a candidate
    // (wanderator: skip_scores_[level] <= threshold_, formats_10.cpp:2521-2528)
    if (score_bin(bound, qd.bin_scale) < bs) {
      // (k_select: with blocks skipped, "fewer than k candidates" no longer shows in the hit
      // count — an estimated threshold that was too high must still be noticed)
      if (lane == 0 && A.pruned) A.pruned[unit] = 1u;
      return;
    }
  }

  // ---- 1. the lead block: entry index 2*lane + h (block) or lane + 64*h (tail)
  uint32_t n = kBlock;
  uint32_t bytes = 0;   // (wave-uniform) encoded bytes of the blocks this wavefront decodes
  const bool counting = !pilot && A.touched != nullptr;   // (irs_hip_batch_profile bit 1)
  auto block_bytes = [](uint32_t bits) {   // header bytes + payloads (all-equal: ~1 byte of vint)
    const uint32_t db = bits & 0xFFu, fb = bits >> 8;
    return 2u + (db ? 16u * db : 1u) + (fb ? 16u * fb : 1u);
  };
  uint32_t ld_d[2], ld_e[2];   // this lane's two lead docs and their entry indices
  bool live[2];                // ... and whether they are docs of the conjunction at all
  uint32_t sub_lo = 0;         // first posting of the block that is this item's (a piece: > 0)
  {
    uint32_t f[2], estep;
    if (item < ld.nblk) {
      decode_dir_block<LAYOUT>(seg, ld.doc_start, R.bits, R.off, R.aoff, R.base, lane, ld_d[0],
                               ld_d[1], f[0], f[1]);
      if (counting) bytes += block_bytes(R.bits);
      ld_e[0] = 2u * lane;
      estep = 1u;
    } else {
      n = ld.n;
      ld_d[0] = lane < n ? seg.tail_docs[ld.tail_row + lane] : 0u;
      ld_d[1] = lane + 64u < n ? seg.tail_docs[ld.tail_row + lane + 64u] : 0u;
      f[0] = lane < n ? seg.tail_freqs[ld.tail_row + lane] : 0u;
      f[1] = lane + 64u < n ? seg.tail_freqs[ld.tail_row + lane + 64u] : 0u;
      ld_e[0] = lane;
      estep = 64u;
    }
    ld_e[1] = ld_e[0] + estep;
    if (split_lg) {   // (wave-uniform) this item's piece of the block: its postings move to the front
      const uint32_t per = (n + (1u << split_lg) - 1u) >> split_lg;
      sub_lo = piece * per;
      n = sub_lo < n ? (n - sub_lo < per ? n - sub_lo : per) : 0u;
      ld_e[0] -= sub_lo;   // (in front of the piece: wraps to "not one of mine")
      ld_e[1] -= sub_lo;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool on = ld_e[h] < n;
      if (!on) {
        live[h] = false;
        continue;
      }
      docs[ld_e[h]] = ld_d[h];
      W.fr[0][ld_e[h]] = f[h];
      score[ld_e[h]] = 0.f;
      // a deleted doc (SegmentReaderImpl::mask) keeps its place among the lead docs — the doc range
      // and the ranks of the others do not change — but is never alive: no term can reach it
      live[h] = on && !(seg.dead && doc_dead(seg.dead, ld_d[h]));
      cnt[ld_e[h]] = live[h] ? 1u : 0u;
    }
    // (words 2*lane, 2*lane+1 of the three bitmaps; the 4 slack words stay zero from here)
    if (lane < (kConjWords + 4u) / 2u) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        W.bm[k][2u * lane] = 0u;
        W.bm[k][2u * lane + 1u] = 0u;
      }
    }
    static_assert(kConjBuckets == 64u * 16u, "one 16-byte store per lane clears `first`");
    reinterpret_cast<ConjQuad*>(W.first)[lane] = ConjQuad{0u, 0u, 0u, 0u};
  }
  // the docs every one of the terms [g0, g1) reached (and every earlier one) get those terms'
  // scores: norm read here, once per flush, for these docs only — they are postings of the lead
  // block: from the posting-order copy of the column where the segment has one (consecutive
  // bytes instead of a gather by doc id)
  const bool with_norm = needs_norm(term_q(0).kind);
  const uint8_t* lead_norms = nullptr;
  if (seg.pnorm)
    lead_norms = (item < ld.nblk ? seg.pnorm + (ld.dir_off + item) * kBlock : seg.tail_norms + ld.tail_row) + sub_lo;
  auto lead_norm = [&](uint32_t sl, uint32_t doc) {
    return lead_norms ? uint32_t(lead_norms[sl]) : norm_value(seg, doc);
  };
  auto flush = [&](uint32_t g0, uint32_t g1) {
    uint32_t scored = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t sl = lane + 64u * uint32_t(h);
      const bool on = sl < n && cnt[sl] == g1;
      if (counting) scored += uint32_t(__builtin_popcountll(wave::ballot(on)));
      if (on) {
        const uint32_t nv = lead_norm(sl, docs[sl]);
        float v = score[sl];
        for (uint32_t j = g0; j < g1; ++j)
          v = merge_scores(mrg, j == 0u, v, score_value(term_q(j), W.fr[j - g0][sl], nv));
        score[sl] = v;
      }
    }
    if (with_norm) bytes += scored * seg.norm_width;
  };
  wave::sync();
  if (n == 0) return;   // (a piece behind the end of a short tail)
  const uint32_t dlo = wave::uniform(docs[0]), dhi = wave::uniform(docs[n - 1]);
  // bucket of a doc: (doc - dlo) >> s, below 32 * kConjWords
  const uint32_t span = dhi - dlo;
  const uint32_t s = span < kConjBuckets ? 0u
                     : 32u - uint32_t(__builtin_clz(span)) - (5u + uint32_t(__builtin_ctz(kConjWords)));
  const bool masked = seg.dead != nullptr;   // (wave-uniform)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (ld_e[h] < n) {
      const uint32_t bk = (ld_d[h] - dlo) >> s;
      atomicOr(&W.bm[0][bk >> 5], 1u << (bk & 31u));
      // (what the first other term may reach is the LIVE lead docs: bm[2], free until the third
      // term marks into it)
      if (masked && live[h]) atomicOr(&W.bm[2][bk >> 5], 1u << (bk & 31u));
      // the bucket's first lead doc (entries are in doc order: its predecessor lies in another)
      if (ld_e[h] == 0u || ((docs[ld_e[h] - 1u] - dlo) >> s) != bk) W.first[bk] = uint8_t(ld_e[h] + 1u);
    }
  }
  wave::sync();
  // bits in the words before word w (lane: words 2*lane, 2*lane+1; [kConjWords] = all)
  auto prefix = [&](const uint32_t* bmap, uint8_t* pre) {
    uint32_t p0 = 0, p1 = 0;
    if (lane < kConjWords / 2u) {
      p0 = uint32_t(__builtin_popcount(bmap[2u * lane]));
      p1 = uint32_t(__builtin_popcount(bmap[2u * lane + 1u]));
    }
    const uint32_t incl = wave::inclusive_scan(p0 + p1);
    if (lane < kConjWords / 2u) {
      pre[2u * lane] = uint8_t(incl - p0 - p1);
      pre[2u * lane + 1u] = uint8_t(incl - p1);
    }
    if (lane == kConjWords / 2u - 1u) pre[kConjWords] = uint8_t(incl);
    wave::sync();
    return wave::read_lane(incl, 63);
  };
  prefix(W.bm[0], W.lpre);

  // ---- 2. the other terms, cheapest first
  for (uint32_t i = 1; i < m; ++i) {
    const DevTail tl = term_tail(i);
    uint32_t* row = W.fr[i % kConjRows];
    if (i % kConjRows == 0) {   // the rows are full: score what is still alive, reuse them
      flush(i - kConjRows, i);
      wave::sync();
    }
    // alive = the docs every earlier term reached: the lead bitmap, then what the previous
    // term marked; `mark` collects what this term reaches
    const uint32_t mk = 1u + ((i - 1u) & 1u);
    const uint32_t* alive = i == 1u ? (masked ? W.bm[2] : W.bm[0]) : W.bm[3u - mk];
    uint32_t* mark = W.bm[mk];
    const uint8_t* apre = W.lpre;
    if (i > 1u || masked) {
      if (i > 1u && lane < kConjWords / 2u) {
        mark[2u * lane] = 0u;
        mark[2u * lane + 1u] = 0u;
      }
      apre = W.apre;
      if (prefix(alive, W.apre) == 0u) {   // no doc reached by every term so far: done
        if (!pilot && A.touched && lane == 0)
          atomicAdd(&A.touched[2u * unit], static_cast<unsigned long long>(bytes));
        return;
      }
    }
    // alive bits in the buckets below x, 0 <= x <= 32 * kConjWords
    auto alive_below = [&](uint32_t x) {
      return uint32_t(apre[x >> 5]) + uint32_t(__builtin_popcount(alive[x >> 5] & ((1u << (x & 31u)) - 1u)));
    };
    // a decoded posting of term i: is its doc one of the lead docs still alive?
    auto put = [&](uint32_t doc, uint32_t f) {
      const uint32_t x = doc - dlo;
      if (f == 0 || x > span) return;
      const uint32_t bk = x >> s;
      const uint32_t t = lead_index(W.first, docs, n, bk, s, doc);
      // (a lead doc every earlier term reached: deleted lead docs count 0 and stay out)
      if (t == n || cnt[t] != i) return;
      row[t] = f;
      cnt[t] = uint8_t(i + 1u);
      if (i + 1u < m) atomicOr(&mark[bk >> 5], 1u << (bk & 31u));   // (alive for the next term)
    };
    if (tl.nblk) {
      // (integer addresses: loads through the global address space, see raw_load_packed_g)
      const uint64_t last_at = reinterpret_cast<uint64_t>(seg.blk_last + tl.dir_off);
      const uint64_t dir_at = reinterpret_cast<uint64_t>(seg.blk_dir + tl.dir_off);
      const uint64_t pk_at = reinterpret_cast<uint64_t>(seg.pk);
      uint32_t b_first = seek[i - 1u];
      if (piece) {   // (wave-uniform) the seek row is the whole lead block's: this piece starts
        // further on — the first block of term i whose last doc reaches the piece's first doc, by a
        // 64-ary search of the directory's last docs (one load per lane and round)
        uint32_t hi = tl.nblk;
        while (hi - b_first > 64u) {
          const uint32_t step = (hi - b_first + 63u) / 64u;
          const uint32_t p = b_first + lane * step;
          const uint32_t v = p < hi ? wave::gload_u32(last_at, p * 4u) : 0xFFFFFFFFu;
          const uint64_t ge = wave::ballot(v >= dlo);
          const uint32_t j = ge ? uint32_t(__builtin_ctzll(ge)) : 64u;
          const uint32_t reach = b_first + j * step;   // (j < 64: block `reach` reaches dlo)
          if (j < 64u && reach < hi) hi = reach;
          if (j) b_first += (j - 1u) * step + 1u;
        }
      }
      for (uint32_t b0 = b_first; b0 < tl.nblk; b0 += 64) {
        const uint32_t bl = b0 + lane;
        const bool valid = bl < tl.nblk;
        // the block's last doc AND its directory record in one round trip (the record also
        // holds the preceding block's last doc): the kernel is a chain of dependent loads —
        // fetching the record only for the blocks that turn out to be wanted made it one longer
        const uint32_t lst = valid ? wave::gload_u32(last_at, bl * 4u) : 0xFFFFFFFFu;
        BlkDir d{};
        if (valid) {
          uint32_t w[4];
          wave::gload_u32x4(dir_at, bl * uint32_t(sizeof(BlkDir)), w);
          d = BlkDir{w[0], w[1], w[2], w[3]};
        }
        // the block holds docs in (prv, lst], prv = the preceding block's lst
        const uint32_t prv = bl ? d.prev_last : 0u;
        const bool reach = valid && prv < dhi && lst >= dlo;
        // some doc every earlier term reached lies in a bucket touching (prv, lst]
        bool want = false;
        if (reach) {
          const uint32_t x0 = prv + 1u > dlo ? prv + 1u - dlo : 0u;
          const uint32_t x1 = (lst < dhi ? lst : dhi) - dlo;
          want = alive_below((x1 >> s) + 1u) > alive_below(x0 >> s);
        }
        uint64_t mask = wave::ballot(want);
        const bool more = wave::ballot(valid && prv >= dhi) == 0;  // no block started behind dhi yet
        while (mask) {
          const uint32_t k = uint32_t(__builtin_ctzll(mask));
          mask &= mask - 1;
          const uint32_t kbits = wave::read_lane(d.bits, k);
          if (counting) bytes += block_bytes(kbits);
          // two wanted blocks at a time where both live in the packed image: their payload
          // loads are in flight together (one round trip for the pair)
          if (mask && pk_both(kbits & 0xFFu, kbits >> 8)) {
            const uint32_t k2 = uint32_t(__builtin_ctzll(mask));
            const uint32_t kbits2 = wave::read_lane(d.bits, k2);
            if (pk_both(kbits2 & 0xFFu, kbits2 >> 8)) {
              mask &= mask - 1;
              if (counting) bytes += block_bytes(kbits2);
              const uint64_t pl1 = pk_at + (uint64_t(wave::read_lane(d.aoff, k)) << 4);
              const uint64_t pl2 = pk_at + (uint64_t(wave::read_lane(d.aoff, k2)) << 4);
              uint64_t da1, db1, fa1, fb1, da2, db2, fa2, fb2;
              raw_load_packed_g<LAYOUT>(pl1, kbits & 0xFFu, lane, da1, db1);
              raw_load_packed_g<LAYOUT>(pl1 + 16u * (kbits & 0xFFu), kbits >> 8, lane, fa1, fb1);
              raw_load_packed_g<LAYOUT>(pl2, kbits2 & 0xFFu, lane, da2, db2);
              raw_load_packed_g<LAYOUT>(pl2 + 16u * (kbits2 & 0xFFu), kbits2 >> 8, lane, fa2, fb2);
              uint32_t x0, x1, f0, f1;
              extract_fast<LAYOUT>(da1, db1, kbits & 0xFFu, lane, x0, x1);
              extract_fast<LAYOUT>(fa1, fb1, kbits >> 8, lane, f0, f1);
              uint32_t d1 = wave::read_lane(d.prev_last, k) + wave::inclusive_scan(x0 + x1);
              put(d1 - x1, f0);
              put(d1, f1);
              extract_fast<LAYOUT>(da2, db2, kbits2 & 0xFFu, lane, x0, x1);
              extract_fast<LAYOUT>(fa2, fb2, kbits2 >> 8, lane, f0, f1);
              d1 = wave::read_lane(d.prev_last, k2) + wave::inclusive_scan(x0 + x1);
              put(d1 - x1, f0);
              put(d1, f1);
              continue;
            }
          }
          uint32_t d0, d1, f0, f1;
          decode_dir_block<LAYOUT>(seg, tl.doc_start, kbits, wave::read_lane(d.off, k),
                                   wave::read_lane(d.aoff, k), wave::read_lane(d.prev_last, k),
                                   lane, d0, d1, f0, f1);
          put(d0, f0);
          put(d1, f1);
        }
        if (!more) break;
      }
    }
    if (tl.n && tl.first_doc <= dhi && tl.last_doc >= dlo) {  // vint tail / single doc
      const uint32_t t0 = lane < tl.n ? seg.tail_docs[tl.tail_row + lane] : 0u;
      const uint32_t t1 = lane + 64u < tl.n ? seg.tail_docs[tl.tail_row + lane + 64u] : 0u;
      const uint32_t g0 = lane < tl.n ? seg.tail_freqs[tl.tail_row + lane] : 0u;
      const uint32_t g1 = lane + 64u < tl.n ? seg.tail_freqs[tl.tail_row + lane + 64u] : 0u;
      put(t0, g0);
      put(t1, g1);
    }
    wave::sync();
  }

  // ---- 3. docs every term reached: compacted (usually one pass of the wavefront), the terms
  // not yet scored added, straight into the histogram (pilot) / the candidate list
  const uint32_t g0 = ((m - 1u) / kConjRows) * kConjRows;
  const bool h0 = lane < n && cnt[lane] == m, h1 = lane + 64u < n && cnt[lane + 64u] == m;
  const uint64_t m0 = wave::ballot(h0), m1 = wave::ballot(h1);
  const uint64_t below = (1ull << lane) - 1ull;
  const uint32_t c0 = uint32_t(__builtin_popcountll(m0));
  const uint32_t total = c0 + uint32_t(__builtin_popcountll(m1));
  uint8_t* list = W.first;   // (the bucket table has served: room for the 128 entry indices)
  if (h0) list[__builtin_popcountll(m0 & below)] = uint8_t(lane);
  if (h1) list[c0 + uint32_t(__builtin_popcountll(m1 & below))] = uint8_t(lane + 64u);
  wave::sync();
  if (counting && with_norm) bytes += total * seg.norm_width;
  if (counting && lane == 0)
    atomicAdd(&A.touched[2u * unit], static_cast<unsigned long long>(bytes));
  for (uint32_t p0 = 0; p0 < total; p0 += 64) {
    const bool on = p0 + lane < total;
    bool cand = false;
    float v = 0.f;
    uint32_t doc = 0;
    if (on) {
      const uint32_t sl = list[p0 + lane];
      doc = docs[sl];
      const uint32_t nv = lead_norm(sl, doc);
      v = score[sl];
      for (uint32_t j = g0; j < m; ++j)
        v = merge_scores(mrg, j == 0u, v, score_value(term_q(j), W.fr[j - g0][sl], nv));
      const uint32_t bin = score_bin(v, qd.bin_scale);
      if (pilot) atomicAdd(&A.hist[uint64_t(unit) * kBins + bin], 1u);
      else cand = bin >= bs;
    }
    // one reservation per wavefront for all its candidates of this pass
    const uint64_t cm = wave::ballot(cand);
    if (cm) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&A.cand_count[unit], uint32_t(__builtin_popcountll(cm)));
      base = wave::read_lane(base, 0);
      const uint32_t slot = base + uint32_t(__builtin_popcountll(cm & below));
      if (cand && slot < A.cand_cap) A.cands[uint64_t(unit) * A.cand_cap + slot] = make_key(v, doc);
    }
  }
  if (!pilot && lane == 0 && total) A.item_hits[e] = total;
}

// hits of a block-driven unit = sum over its lead items (ConjArgs::item_hits): one wavefront per unit.
__global__ void __launch_bounds__(64)
k_conj_hits(const uint32_t* conj_units, const uint32_t* item_base, const uint32_t* item_hits,
            unsigned long long* hits) {
  const unsigned lane = threadIdx.x;
  const uint32_t c = blockIdx.x;
  unsigned long long sum = 0;
  for (uint32_t i = item_base[c] + lane; i < item_base[c + 1u]; i += 64u) sum += item_hits[i];
  // (24-bit pieces: 64 of them cannot overflow the 32-bit wavefront sum)
  const uint32_t lo = wave::reduce_add(uint32_t(sum) & 0xFFFFFFu), hi = wave::reduce_add(uint32_t(sum >> 24));
  if (lane == 0) hits[conj_units[c]] = (static_cast<unsigned long long>(hi) << 24) + lo;
}

// Threshold bin of a unit from the pilot histogram (the rule of k_pilot): one wavefront per unit.
// `items[unit]` = lead items of the unit.
__global__ void __launch_bounds__(64)
k_conj_threshold(const DevQuery* queries, const uint32_t* conj_units, const uint32_t* n_items,
                 const uint32_t* hist, uint32_t stride, uint32_t margin, uint32_t* bstar,
                 const uint32_t* min_bin /*[unit] bin of the caller's score::Min; null: none*/) {
  const unsigned lane = threadIdx.x;
  const uint32_t unit = conj_units[blockIdx.x];
  const DevQuery qd = queries[unit];
  const uint32_t* h = hist + uint64_t(unit) * kBins;
  const uint32_t total = n_items[blockIdx.x];
  uint32_t need = qd.k;
  if (margin) {
    const uint32_t phase = (unit * 7u) % stride;
    const uint32_t sampled = phase < total ? (total - phase + stride - 1) / stride : 0u;
    const uint64_t est = total ? (uint64_t(margin) * qd.k * sampled + total - 1) / total : 0u;
    const uint32_t lo = est < kPilotMinSample ? kPilotMinSample : uint32_t(est < 0xFFFFFFFFull ? est : 0xFFFFFFFFull);
    need = lo < qd.k ? lo : qd.k;
  }
  const uint32_t chunk = 63u - lane;
  uint32_t s = 0;
  for (uint32_t i = 0; i < kBins / 64; ++i) s += h[chunk * (kBins / 64) + i];
  const uint32_t incl = wave::inclusive_scan(s);  // docs in chunks >= chunk
  const uint64_t reach = wave::ballot(incl >= need);
  uint32_t result = 0;
  if (reach) {
    const int src = __builtin_ctzll(reach);  // highest chunk reaching `need`
    const uint32_t above = wave::bcast(incl - s, src);
    const uint32_t c = 63u - uint32_t(src);
    uint32_t cum = above;
    for (int i = int(kBins / 64) - 1; i >= 0; --i) {
      cum += h[c * (kBins / 64) + uint32_t(i)];
      if (cum >= need) { result = c * (kBins / 64) + uint32_t(i); break; }
    }
  }
  if (lane == 0) bstar[unit] = (min_bin && min_bin[unit] > result) ? min_bin[unit] : result;
}

}  // namespace irs_hip
