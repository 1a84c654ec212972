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
//   1. the lead block is decoded: 128 ascending docs, their scores and norms into LDS;
//   2. for every other term, cheapest first: binary search of the block directory for the
//      first block reaching the lead block's first doc, then 64 directory entries per step, a
//      lane each, tested against the docs still alive ("is any of them in (previous last,
//      last]"); only blocks that pass are decoded, and each decoded posting looks its doc up
//      among the lead docs (binary search in LDS): a hit adds the term's score.  A term that
//      leaves no doc alive ends the block;
//   3. docs every term reached are the conjunction: score = the per-term scores summed in
//      cost order (the order the reference's Conjunction sums them); docs at or above the
//      threshold bin become candidates for k_select.
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
// a list's last block).  grid = num_terms * slices, as k_pack_payloads.
template<int LAYOUT>
__global__ void __launch_bounds__(kThreads)
k_block_max(DevSegment seg, uint32_t slices, uint32_t* blk_maxf, uint32_t* blk_minn) {
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t slice = blockIdx.x % slices;
  const DevTerm t = seg.terms[blockIdx.x / slices];
  if (t.docs_count < 2) return;
  for (uint32_t b = slice * kWaves + (threadIdx.x >> 6); b < t.nblk; b += slices * kWaves) {
    const uint64_t e = t.dir_off + b;
    const uint32_t bits = seg.blk_bits[e];
    const uint32_t base = b ? seg.blk_last[e - 1] : kDocMin;
    uint32_t d0, d1, f0, f1;
    decode_block<LAYOUT, true>(seg.doc + t.doc_start + seg.blk_off[e], bits & 0xFFu, bits >> 8,
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

// Doc block `e` of a term, from the packed image when both parts live there (one funnel
// shift + one bit-field extract per value), else from `.doc` (any framing).
template<int LAYOUT>
__device__ __forceinline__ void decode_dir_block(const DevSegment& seg, uint64_t doc_start,
                                                 uint32_t bits, uint32_t off, uint32_t aoff,
                                                 uint32_t base, unsigned lane, uint32_t& d0,
                                                 uint32_t& d1, uint32_t& f0, uint32_t& f1) {
  const uint32_t dbits = bits & 0xFFu, fbits = bits >> 8;
  if (pk_both(dbits, fbits)) {
    const uint8_t* pl = seg.pk + (uint64_t(aoff) << 4);
    uint64_t da, db, fa, fb;
    raw_load_packed<LAYOUT>(pl, dbits, lane, da, db);
    raw_load_packed<LAYOUT>(pl + 16u * dbits, fbits, lane, fa, fb);
    uint32_t x0, x1;
    extract_fast<LAYOUT>(da, db, dbits, lane, x0, x1);
    extract_fast<LAYOUT>(fa, fb, fbits, lane, f0, f1);
    d1 = base + wave::inclusive_scan(x0 + x1);
    d0 = d1 - x1;
  } else {
    decode_block<LAYOUT, true>(seg.doc + doc_start + off, dbits, fbits, base, lane, d0, d1, f0, f1);
  }
}

constexpr uint32_t kConjWaves = 4;  // wavefronts (= lead blocks) per workgroup

// Where the other terms start for every lead item: the binary search of a term's block
// directory for the first block reaching the lead item's first doc — SkipReader::Seek
// (skip_list.hpp:208-249) — done once per (lead item, term) by ONE THREAD of a pre-pass
// (inside k_conj a wavefront would walk the same dependent chain 64 lanes wide).
// One thread per lead item of every conjunction; seek[(item_base + item) * (jt - 1) + i - 1].
__global__ void __launch_bounds__(kThreads)
k_conj_seek(const DevSegment* segs, const DevQuery* queries, const DevTail* tails, uint32_t jt,
            const uint32_t* conj_units, const uint32_t* item_base /*[n_conj + 1]*/,
            uint32_t n_conj, uint32_t* seek) {
  const uint32_t t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= item_base[n_conj]) return;
  uint32_t lo = 0, hi = n_conj;   // the unit whose items hold t: last c with item_base[c] <= t
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (item_base[mid] <= t) lo = mid; else hi = mid;
  }
  const uint32_t unit = conj_units[lo], item = t - item_base[lo];
  const DevQuery qd = queries[unit];
  const DevSegment& seg = segs[qd.seg];
  const DevTail* tl = tails + uint64_t(unit) * jt;
  const DevTail ld = tl[0];
  const uint32_t r_lo = item < ld.nblk ? (item ? seg.blk_last[ld.dir_off + item - 1] + 1u : kDocMin)
                                       : ld.first_doc;
  for (uint32_t i = 1; i < qd.n_terms; ++i) {
    const uint32_t* last = seg.blk_last + tl[i].dir_off;
    uint32_t a = 0, b = tl[i].nblk;  // lower_bound(last, r_lo)
    while (a < b) {
      const uint32_t mid = (a + b) >> 1;
      if (last[mid] < r_lo) a = mid + 1; else b = mid;
    }
    seek[uint64_t(t) * (jt - 1u) + (i - 1u)] = a;
  }
}

struct ConjArgs {
  const DevSegment* segs;
  const DevQuery* queries;
  const DevQTerm* qterms;
  const PhraseWg* wgs;          // {unit, first lead item} per workgroup; pilot pass: {unit, lead
                                // item} per WAVEFRONT (the sampled items only)
  uint32_t n_pilot;             // entries of the pilot list
  const DevTail* tails;         // [unit][jt] (k_plan)
  const uint32_t* bstar;        // threshold bin per unit (0 = none)
  const uint32_t* seek;         // k_conj_seek
  const uint32_t* unit_items;   // [nq] first row of the unit's lead items in `seek` (conj units)
  uint64_t* cands;
  uint32_t* cand_count;
  unsigned long long* hits;
  unsigned long long* touched;  // [unit][2]: `.doc` + norm bytes actually decoded / read (full
                                // pass; per unit: one hot address would serialise the atomics);
                                // null unless the batch counts (irs_hip_batch_profile bit 1)
  uint32_t* hist;               // [unit][kBins], pilot pass only
  uint32_t jt;
  uint32_t cand_cap;
  uint32_t pilot_stride;        // pilot pass: lead items {phase, phase + P, ...}
  uint32_t wand;                // prune lead blocks by block-max bounds
};

// 8 wavefronts per SIMD (a 64-VGPR budget, a few spilled values): the kernel waits on chains
// of dependent loads, so resident wavefronts count for more than registers (AND-3: 6.8 -> 6.0 ms)
#define IRS_CONJ_ATTR RT_WAVES_PER_SIMD(8)
template<int LAYOUT>
__global__ void __launch_bounds__(kConjWaves * 64) IRS_CONJ_ATTR
k_conj(ConjArgs A, uint32_t pilot) {
  __shared__ DevTail s_tl[kConjWaves * kMaxTerms];
  __shared__ DevQTerm s_qt[kConjWaves * kMaxTerms];
  __shared__ uint32_t s_docs[kConjWaves][kBlock];
  __shared__ float s_score[kConjWaves][kBlock];
  __shared__ uint32_t s_norm[kConjWaves][kBlock];
  __shared__ uint32_t s_cnt[kConjWaves][kBlock];    // terms that reached the doc so far
  __shared__ uint32_t s_alive[kConjWaves][kBlock + 1];  // docs alive among the first c
  const uint32_t tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const uint32_t wv = tid >> 6;
  // full pass: a workgroup = kConjWaves consecutive lead items of one unit; pilot pass: every
  // wavefront has its own (unit, item) entry (wavefronts of a workgroup may belong to
  // different units: the per-unit records are then read per wavefront, not staged)
  uint32_t unit, item;
  if (pilot) {
    const uint32_t e = blockIdx.x * kConjWaves + wv;
    const PhraseWg w = A.wgs[e < A.n_pilot ? e : A.n_pilot - 1u];
    unit = w.unit;
    item = e < A.n_pilot ? w.first_item : 0xFFFFFFFFu;
  } else {
    const PhraseWg w = A.wgs[blockIdx.x];
    unit = w.unit;
    item = w.first_item + wv;
  }
  const DevQuery qd = A.queries[unit];
  const uint32_t m = qd.n_terms;
  const DevSegment seg = A.segs[qd.seg];
  // (the term records of this wavefront's unit, in its own LDS rows)
  DevTail* w_tl = s_tl + wv * kMaxTerms;
  DevQTerm* w_qt = s_qt + wv * kMaxTerms;
  if (lane < m) {
    w_tl[lane] = A.tails[uint64_t(unit) * A.jt + lane];
    w_qt[lane] = A.qterms[qd.first_term + lane];
  }
  wave::sync();
  if (m == 0) return;
  const DevTail ld = w_tl[0];   // the host sorted the terms by cost: the cheapest leads
  const uint32_t n_items = ld.nblk + (ld.n ? 1u : 0u);
  if (item >= n_items) return;  // whole wavefront
  const uint32_t bs = pilot ? 0u : A.bstar[unit];
  uint32_t* docs = s_docs[wv];
  float* score = s_score[wv];
  uint32_t* nrm = s_norm[wv];
  uint32_t* cnt = s_cnt[wv];
  uint32_t* alive = s_alive[wv];
  const uint32_t* seek = A.seek + uint64_t(A.unit_items[unit] + item) * (A.jt - 1u);

  // ---- doc range of the lead block (from the directory: nothing decoded yet)
  uint32_t r_lo, r_hi;   // the lead block's docs lie in [r_lo, r_hi]
  uint32_t lead_base = kDocMin;   // what its first delta is relative to (formats_10.cpp:636)
  if (item < ld.nblk) {
    r_hi = seg.blk_last[ld.dir_off + item];
    if (item) lead_base = seg.blk_last[ld.dir_off + item - 1];
    r_lo = item ? lead_base + 1u : kDocMin;
  } else {
    r_lo = ld.first_doc;
    r_hi = ld.last_doc;
  }

  // ---- 0. block-max bound of every doc of this lead block (WandContext: ExecutionContext::wand)
  if (A.wand && bs) {
    float bound = 0.f;
    for (uint32_t i = 0; i < m; ++i) {
      const DevTail tl = w_tl[i];
      const DevQTerm qt = w_qt[i];
      float ub = 0.f;
      if (i == 0 && item < ld.nblk) {
        ub = block_bound(qt, seg.blk_maxf[ld.dir_off + item], seg.blk_minn[ld.dir_off + item]);
      } else if (i == 0) {
        ub = term_bound(qt, seg.terms[tl.term].tf_bound);
      } else {
        // blocks of term i overlapping [r_lo, r_hi]: rows [a, z): from the seek table (the next
        // lead item starts behind r_hi, so its first row bounds this item's last one)
        const uint32_t a = seek[i - 1u];
        uint32_t z = item + 1u < n_items ? seek[(A.jt - 1u) + i - 1u] + 1u : tl.nblk;
        z = z < tl.nblk ? z : tl.nblk;
        for (uint32_t k = a + lane; k < z; k += 64) {
          const float s = block_bound(qt, seg.blk_maxf[tl.dir_off + k], seg.blk_minn[tl.dir_off + k]);
          ub = s > ub ? s : ub;
        }
        // the decoded tail (no block-max entry): the term's global bound
        if (tl.n && tl.first_doc <= r_hi && tl.last_doc >= r_lo) {
          const float s = term_bound(qt, seg.terms[tl.term].tf_bound);
          ub = s > ub ? s : ub;
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
          const float o = __shfl_xor(ub, s, 64);
          ub = o > ub ? o : ub;
        }
      }
      bound += ub;   // the same order the scores are summed in
    }
    // every doc of the block scores <= bound: below the threshold bin none can be a candidate
    // (wanderator: skip_scores_[level] <= threshold_, formats_10.cpp:2521-2528)
    if (score_bin(bound, qd.bin_scale) < bs) return;
  }

  // ---- 1. the lead block: entry index 2*lane + h (block) or lane + 64*h (tail)
  uint32_t n = kBlock;
  uint32_t bytes = 0;   // (wave-uniform) encoded bytes of the blocks this wavefront decodes
  auto block_bytes = [](uint32_t bits) {   // header bytes + payloads (all-equal: ~1 byte of vint)
    const uint32_t db = bits & 0xFFu, fb = bits >> 8;
    return 2u + (db ? 16u * db : 1u) + (fb ? 16u * fb : 1u);
  };
  {
    const DevQTerm qt = w_qt[0];
    uint32_t d[2], f[2], e0, estep;
    if (item < ld.nblk) {
      const uint64_t e = ld.dir_off + item;
      decode_dir_block<LAYOUT>(seg, ld.doc_start, seg.blk_bits[e], seg.blk_off[e], seg.blk_aoff[e],
                               lead_base, lane, d[0], d[1], f[0], f[1]);
      bytes += block_bytes(seg.blk_bits[e]);
      e0 = 2u * lane;
      estep = 1u;
    } else {
      n = ld.n;
      d[0] = lane < n ? seg.tail_docs[ld.tail_row + lane] : 0u;
      d[1] = lane + 64u < n ? seg.tail_docs[ld.tail_row + lane + 64u] : 0u;
      f[0] = lane < n ? seg.tail_freqs[ld.tail_row + lane] : 0u;
      f[1] = lane + 64u < n ? seg.tail_freqs[ld.tail_row + lane + 64u] : 0u;
      e0 = lane;
      estep = 64u;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t idx = e0 + uint32_t(h) * estep;
      const bool on = idx < n;
      const uint32_t nv = on ? norm_value(seg, d[h]) : 1u;
      docs[idx] = on ? d[h] : 0xFFFFFFFFu;
      nrm[idx] = nv;
      score[idx] = on ? score_value(qt, f[h], nv) : 0.f;
      cnt[idx] = on ? 1u : 0u;
    }
  }
  wave::sync();
  const uint32_t dlo = docs[0], dhi = docs[n - 1];

  // ---- 2. the other terms, cheapest first
  for (uint32_t i = 1; i < m; ++i) {
    const DevTail tl = w_tl[i];
    const DevQTerm qt = w_qt[i];
    // alive[c] = docs among the first c that every earlier term reached
    {
      const uint32_t a0 = (2u * lane < n && cnt[2u * lane] == i) ? 1u : 0u;
      const uint32_t a1 = (2u * lane + 1u < n && cnt[2u * lane + 1u] == i) ? 1u : 0u;
      const uint32_t incl = wave::inclusive_scan(a0 + a1);
      if (lane == 0) alive[0] = 0u;
      alive[2u * lane + 1u] = incl - a1;
      alive[2u * lane + 2u] = incl;
      wave::sync();
      if (alive[kBlock] == 0u) {   // no doc reached by every term so far: the block is done
        if (!pilot && A.touched && lane == 0)
          atomicAdd(&A.touched[2u * unit], static_cast<unsigned long long>(bytes));
        return;
      }
    }
    // a decoded posting of term i: is its doc one of the lead docs still alive?
    // (w0, w1] = ranks of the lead docs that can equal it: those inside its block's doc range
    // (a binary search all lanes finish together; a hash table was tried and lost to the
    // longest probe sequence among the 64 lanes)
    auto put = [&](uint32_t doc, uint32_t f, uint32_t w0, uint32_t w1) {
      if (f == 0 || doc < dlo || doc > dhi) return;
      const uint32_t c = count_le(docs, w0, w1, doc);
      if (c > w0 && docs[c - 1] == doc && cnt[c - 1] == i) {
        score[c - 1] += score_value(qt, f, nrm[c - 1]);
        cnt[c - 1] = i + 1u;
      }
    };
    if (tl.nblk) {
      const uint32_t* last = seg.blk_last + tl.dir_off;
      for (uint32_t b0 = seek[i - 1u]; b0 < tl.nblk; b0 += 64) {
        const uint32_t bl = b0 + lane;
        const bool valid = bl < tl.nblk;
        const uint32_t lst = valid ? last[bl] : 0xFFFFFFFFu;
        const uint32_t prv = (valid && bl) ? last[bl - 1] : 0u;  // block holds docs in (prv, lst]
        const bool reach = valid && prv < dhi;
        const uint32_t cp_l = reach ? count_le(docs, 0u, n, prv) : 0u;
        const uint32_t cl_l = reach ? count_le(docs, cp_l, n, lst) : 0u;
        // some lead doc that every earlier term reached lies in (prv, lst]
        const bool want = alive[cl_l] > alive[cp_l];
        BlkDir d{};
        if (want) d = seg.blk_dir[tl.dir_off + bl];
        uint64_t mask = wave::ballot(want);
        const bool more = wave::ballot(valid && !reach) == 0;  // no block started behind dhi yet
        while (mask) {
          const uint32_t k = uint32_t(__builtin_ctzll(mask));
          mask &= mask - 1;
          uint32_t d0, d1, f0, f1;
          decode_dir_block<LAYOUT>(seg, tl.doc_start, wave::read_lane(d.bits, k),
                                   wave::read_lane(d.off, k), wave::read_lane(d.aoff, k),
                                   wave::read_lane(d.prev_last, k), lane, d0, d1, f0, f1);
          bytes += block_bytes(wave::read_lane(d.bits, k));
          const uint32_t w0 = wave::read_lane(cp_l, k), w1 = wave::read_lane(cl_l, k);
          put(d0, f0, w0, w1);
          put(d1, f1, w0, w1);
        }
        if (!more) break;
      }
    }
    if (tl.n && tl.first_doc <= dhi && tl.last_doc >= dlo) {  // vint tail / single doc
      const uint32_t t0 = lane < tl.n ? seg.tail_docs[tl.tail_row + lane] : 0u;
      const uint32_t t1 = lane + 64u < tl.n ? seg.tail_docs[tl.tail_row + lane + 64u] : 0u;
      const uint32_t g0 = lane < tl.n ? seg.tail_freqs[tl.tail_row + lane] : 0u;
      const uint32_t g1 = lane + 64u < tl.n ? seg.tail_freqs[tl.tail_row + lane + 64u] : 0u;
      put(t0, g0, 0u, n);
      put(t1, g1, 0u, n);
    }
    wave::sync();
  }

  // ---- 3. docs every term reached
  if (!pilot && A.touched && lane == 0) {
    // + the norm of every lead doc, where the scorer reads one
    const uint32_t nb = needs_norm(w_qt[0].kind) ? n * seg.norm_width : 0u;
    atomicAdd(&A.touched[2u * unit], static_cast<unsigned long long>(bytes + nb));
  }
  uint32_t my_hits = 0;
  for (uint32_t s = lane; s < n; s += 64) {
    if (cnt[s] != m) continue;
    ++my_hits;
    const float v = score[s];
    const uint32_t bin = score_bin(v, qd.bin_scale);
    if (pilot) {
      atomicAdd(&A.hist[uint64_t(unit) * kBins + bin], 1u);
    } else if (bin >= bs) {
      const uint32_t slot = atomicAdd(&A.cand_count[unit], 1u);
      if (slot < A.cand_cap) A.cands[uint64_t(unit) * A.cand_cap + slot] = make_key(v, docs[s]);
    }
  }
  if (!pilot) {
    my_hits = wave::reduce_add(my_hits);
    if (lane == 0 && my_hits) atomicAdd(&A.hits[unit], static_cast<unsigned long long>(my_hits));
  }
}

// Threshold bin of a unit from the pilot histogram (the rule of k_pilot): one wavefront per unit.
// `items[unit]` = lead items of the unit.
__global__ void __launch_bounds__(64)
k_conj_threshold(const DevQuery* queries, const uint32_t* conj_units, const uint32_t* n_items,
                 const uint32_t* hist, uint32_t stride, uint32_t margin, uint32_t* bstar) {
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
  if (lane == 0) bstar[unit] = result;
}

}  // namespace irs_hip
