// fast.h — plain disjunctions over joined posting streams in TWO passes:
//
//   k_join_fast     every (query, doc tile): accumulate an APPROXIMATE score per doc in packed
//                   16-bit accumulators (two docs per LDS word), count the matching docs exactly,
//                   emit the docs whose approximate score can still reach the threshold bin
//   k_join_rescore  per query: keep the candidates that can still belong to the top k, look their
//                   postings up in the exact entry streams and re-score them EXACTLY — the same
//                   fixed-point arithmetic as join.h / score.h, so the hits are bit-identical
//
// Replaces, like join.h, block_disjunction::refill (disjunction.hpp:1240-1351) + the harness heap
// (index-search.cpp:740-787) for Or filters; what the reference does per posting — a score
// function call and a `+=` into score_buf — is split into a cheap bound and an exact evaluation
// of the few docs that matter, which is what its own WAND mode does with block-max scores
// (disjunction.hpp:1133-1167), only per doc.
//
// Why: PMC of round 3's k_join_score: more than half of its instructions are per
// (query, tile) costs — the accumulator scan of the epilogue and the work split — and both LDS
// accesses of a posting (table gather + accumulator add) conflict on the banks.  A tile of LDS
// holds 12288 exact accumulators but 24576 packed ones: half the tiles, half the epilogue bytes
// per doc, and a posting costs ONE LDS operation because its query-independent factor
// T(tf, norm) was evaluated once, by k_join, for all queries of the batch:
//
//   fast entry  [ T / Tn * 2^16 : 16 | (doc - first doc of its 12288-doc half tile) * 4 : 16 ]
//   per posting c = mul_hi(entry, csq) + 1        csq = c0 * Tn * S16 of the (query, term)
//               acc16[doc] += c                   one v_mul_hi, one v_lshl_add, one v_and, one ds_add
//
// Sums stay below 2^15 (S16 is the query's scale), so the two halves of a word never carry into
// each other and SWAR / v_pk_*_u16 tests work on both at once.  Error: |c - x| < 3 units per posting
// against the real-valued contribution x (u16 floor, the entry's low bits riding along, csq floor,
// mul_hi floor, + 1), i.e. |A - X| < 3 n for a doc of an n-term query.
#pragma once
#include "join.h"

namespace irs_hip {

constexpr uint32_t kFastTile = 2u * kJoinTile;   // docs per packed accumulator tile
constexpr uint32_t kFastChunkTiles = 16;         // tiles per work-queue item (= a join.h chunk's doc range)
constexpr uint32_t kFastErr = 3;                 // |c - x| per posting, in 16-bit units
constexpr uint32_t kFastKeep = 6;                // x n: slack under the k-th approximate score (see k_join_rescore)
constexpr uint32_t kFastMaxSum = 32767;          // a doc's accumulator stays below 2^15
constexpr uint32_t kFastBins = 1024;             // coarse histogram of approximate scores (32 units per bin)
constexpr uint32_t kFastRound = 4096;            // candidates re-scored per round of k_join_rescore

// LDS layout of k_join_fast (byte offsets; the accumulators sit at LDS address 0: an entry's low
// 16 bits ARE the byte address of its doc's word)
struct FastOff {
  static constexpr uint32_t acc = 0;                                      // [kJoinTile] u32 = 2 x u16
  static constexpr uint32_t rng = 4u * kJoinTile;                         // [2 * chunk tiles + 1][kMaxTerms] u32
  static constexpr uint32_t jts = rng + 4u * (2u * kFastChunkTiles + 1u) * kMaxTerms;   // JoinTerm[kMaxTerms]
  static constexpr uint32_t share = jts + uint32_t(sizeof(JoinTerm)) * kMaxTerms;       // [16 waves][4] u32
  static constexpr uint32_t cand = share + 16u * 16u;                     // [2][kJoinCands] u64
  static constexpr uint32_t vars = cand + 8u * 2u * kJoinCands;           // [16] u32
  static constexpr uint32_t end = vars + 64u;
};
static_assert(FastOff::cand % 8u == 0u, "candidate keys are 8-byte aligned");
static_assert(4u * kJoinTile <= 65536u, "an entry's 16 address bits reach every word");

struct FastArgs {
  const DevQuery* queries;
  const JoinTerm* jterms;      // entries / bounds of the EXACT streams; pad[0] = csq
  int64_t fast_delta;          // fast entries of a stream = its exact entries + this many bytes
  const uint32_t* bstar;
  uint64_t* cands;             // [unit][cap] (approximate score << 32) | doc
  uint32_t* cand_count;
  unsigned long long* hits;
  const uint32_t* order;       // as JoinArgs: one work queue per XCD
  uint32_t* work_counter;
  uint32_t base[kJoinQueues + 1];
  uint32_t first[kJoinQueues + 1];
  uint32_t cpq, n_units, nw_log2, cand_cap, chunk_tiles;
};

// what a wavefront takes of a chunk's entries: from fraction f_lo of term j0 to fraction f_hi of
// term j1 (32-bit fixed point; 0xFFFFFFFF = the whole term), every term in between whole
struct FastShare {
  uint32_t j0, j1, f_lo, f_hi;
};
__device__ __forceinline__ uint32_t fast_cut(uint32_t n, uint32_t f) {
  return f == 0xFFFFFFFFu ? n : wave::mul_hi(n, f);
}

// `count` consecutive fast entries from `base` (wave-uniform), all of one half tile: the
// contribution goes to the low (shift 0) or the high (16) half of the doc's word
__device__ __forceinline__ void fast_post4(const unsigned char* lds, const uint32_t (&e)[4], uint32_t csq,
                                           uint32_t shift, uint32_t one) {
  uint32_t c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) c[k] = (wave::mul_hi(e[k], csq) << shift) + one;
#pragma unroll
  for (int k = 0; k < 4; ++k) wave::lds_add(lds, FastOff::acc + (e[k] & 0xFFFCu), c[k]);
}
__device__ __forceinline__ void fast_run(const unsigned char* lds, uint64_t base, uint32_t count,
                                         uint32_t csq, uint32_t shift, unsigned lane) {
  const uint32_t off = lane * 4u;
  const uint32_t one = 1u << shift;
  // 256 entries per step, the NEXT step's four loads in flight while this step's are accumulated:
  // two register sets alternate (written out: a rotating copy would wait for the loads it moves)
  if (count >= 256u) {
    uint32_t p[4], q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = wave::gload_u32(base, off + 256u * uint32_t(k));
    while (count >= 768u) {   // (every load of the loop is unconditional: fixed wait counts)
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = wave::gload_u32(base, off + 1024u + 256u * uint32_t(k));
      wave::keep_all(p);
      fast_post4(lds, p, csq, shift, one);
#pragma unroll
      for (int k = 0; k < 4; ++k) p[k] = wave::gload_u32(base, off + 2048u + 256u * uint32_t(k));
      wave::keep_all(q);
      fast_post4(lds, q, csq, shift, one);
      base += 2048u;
      count -= 512u;
    }
    if (count >= 512u) {
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = wave::gload_u32(base, off + 1024u + 256u * uint32_t(k));
      wave::keep_all(p);
      fast_post4(lds, p, csq, shift, one);
      wave::keep_all(q);
      fast_post4(lds, q, csq, shift, one);
      base += 2048u;
      count -= 512u;
    } else {
      wave::keep_all(p);
      fast_post4(lds, p, csq, shift, one);
      base += 1024u;
      count -= 256u;
    }
  }
  // the tail: slab by slab, lanes past the end masked off
  for (uint32_t s = 0; s < count; s += 64u) {
    if (s + lane < count) {
      const uint32_t e = wave::gload_u32(base, off + 4u * s);
      wave::lds_add(lds, FastOff::acc + (e & 0xFFFCu), (wave::mul_hi(e, csq) << shift) + one);
    }
  }
}

// Persistent workgroups pulling chunks of kFastChunkTiles consecutive tiles of one unit from one
// work queue per XCD (join.h join_pull).  Per chunk the entries of the query's terms are dealt to
// the wavefronts ONCE, as fractions of every term's entries in the chunk (FastShare): per tile a
// wavefront then only scales its fractions to the tile's counts — scalar arithmetic on one or two
// terms — instead of splitting every tile's entries anew.  Per tile: accumulate, barrier, every
// thread reads + clears its 12 words (24 docs), barrier.
__global__ void __launch_bounds__(kTileThreadsMax) IRS_WAVES_PER_SIMD(8)
k_join_fast(const FastArgs* __restrict__ args) {
  RT_DYN_SMEM(smem);
  if (!wave::lds_is_at_zero(smem)) __builtin_trap();
  uint32_t* acc = reinterpret_cast<uint32_t*>(smem + FastOff::acc);
  uint32_t* rng = reinterpret_cast<uint32_t*>(smem + FastOff::rng);
  JoinTerm* jts = reinterpret_cast<JoinTerm*>(smem + FastOff::jts);
  FastShare* share = reinterpret_cast<FastShare*>(smem + FastOff::share);
  uint64_t* lcand = reinterpret_cast<uint64_t*>(smem + FastOff::cand);
  uint32_t* vars = reinterpret_cast<uint32_t*>(smem + FastOff::vars);
  const uint32_t tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const uint32_t wv = wave::uniform(tid >> 6);
  const uint32_t nw = blockDim.x >> 6;
  const uint32_t total_chunks = args->base[kJoinQueues];
  const uint32_t cap = args->cand_cap;

  for (uint32_t i = tid; i < kJoinTile; i += blockDim.x) acc[i] = 0u;
  if (tid < 16u) vars[tid] = 0u;
  __syncthreads();
  auto pull = [&](uint32_t g0, uint32_t first_raw, uint32_t& g_out) -> uint32_t {
    g_out = g0;
    if (first_raw < args->base[g0 + 1u]) return first_raw;
    for (uint32_t k = 1; k < kJoinQueues; ++k) {
      const uint32_t g = (g0 + k) % kJoinQueues;
      if (args->base[g + 1u] == args->base[g]) continue;
      const uint32_t raw = atomicAdd(&args->work_counter[g], 1u);
      if (raw < args->base[g + 1u]) {
        g_out = g;
        return raw;
      }
    }
    return args->base[kJoinQueues];
  };
  if (tid == 0) {
    const uint32_t g0 = blockIdx.x % kJoinQueues;
    uint32_t g = g0;
    vars[kJChunk] = pull(g0, atomicAdd(&args->work_counter[g0], 1u), g);
    vars[kJGroup] = g;
  }
  __syncthreads();
  uint32_t chunk = wave::uniform(vars[kJChunk]);
  uint32_t group = wave::uniform(vars[kJGroup]);
  uint32_t parity = 0;
  uint32_t pend_q = 0, pend_n = 0, pend_base = 0;   // thread 0: the previous chunk's reservation
  __syncthreads();

  while (chunk < total_chunks) {
    uint32_t next_raw = 0;   // (a returning atomic in flight until the hand-over)
    if (tid == 0) next_raw = atomicAdd(&args->work_counter[group], 1u);
    const uint32_t n_units = args->first[group + 1u] - args->first[group];
    const uint32_t local = chunk - args->base[group];
    const uint32_t q = wave::uniform(args->order[args->first[group] + local % n_units]);
    const uint32_t per_chunk = args->chunk_tiles;
    const uint32_t tile0 = (local / n_units) * per_chunk;   // (fast tiles)
    const DevQuery qd = args->queries[q];
    const uint32_t n_half = qd.n_tiles;                      // the unit's kJoinTile tiles
    const uint32_t n_tiles = (n_half + 1u) >> 1;
    const uint32_t ntile = tile0 >= n_tiles ? 0u
                           : ((n_tiles - tile0) < per_chunk ? (n_tiles - tile0) : per_chunk);
    const uint32_t bs = args->bstar[q];
    const uint32_t n_terms = qd.n_terms;
    uint32_t hit_pk = 0;   // matches seen by this lane: low half | high half
    uint64_t* lc = lcand + parity * kJoinCands;
    uint32_t* ncand = vars + kJNc + parity;
    if (ntile) {
      // half-tile boundaries of the chunk, every term: rng[i][j] = bounds_j[2 * tile0 + i]
      // (a segment with an odd number of half tiles: the last boundary once more)
      if (tid < 2u * kMaxTerms) {   // a JoinTerm = two 16-byte halves
        const uint32_t j = tid >> 1;
        uint32_t x = 0, y = 0, z = 0, w = 0;
        if (j < n_terms) {
          const uint32_t* src = reinterpret_cast<const uint32_t*>(args->jterms + qd.first_term) + 4u * tid;
          x = src[0]; y = src[1]; z = src[2]; w = src[3];
        }
        JoinQuad* d = reinterpret_cast<JoinQuad*>(jts);
        d[tid].x = x; d[tid].y = y; d[tid].z = z; d[tid].w = w;
      }
      for (uint32_t e = tid; e < (2u * ntile + 1u) * kMaxTerms; e += blockDim.x) {
        const uint32_t i = e / kMaxTerms, j = e % kMaxTerms;
        uint32_t v = 0;
        if (j < n_terms) {
          uint32_t h = 2u * tile0 + i;
          h = h < n_half ? h : n_half;
          v = reinterpret_cast<const uint32_t*>(args->jterms[qd.first_term + j].bounds)[h];
        }
        rng[e] = v;
      }
      __syncthreads();
      // the wavefronts' shares of the chunk (one thread each)
      if (tid < nw) {
        uint64_t W = 0;
        for (uint32_t j = 0; j < n_terms; ++j)
          W += rng[2u * ntile * kMaxTerms + j] - rng[j];
        const uint64_t lo = W * tid / nw, hi = W * (tid + 1u) / nw;
        FastShare s{1u, 0u, 0u, 0u};   // (j0 > j1: nothing)
        if (hi > lo) {
          uint64_t P = 0;
          bool open = false;
          for (uint32_t j = 0; j < n_terms; ++j) {
            const uint64_t N = rng[2u * ntile * kMaxTerms + j] - rng[j];
            if (!N) continue;
            if (!open && P + N > lo) {
              open = true;
              s.j0 = j;
              s.f_lo = uint32_t(((lo - P) << 32) / N);
            }
            if (open && P < hi) {
              s.j1 = j;
              s.f_hi = (hi - P >= N) ? 0xFFFFFFFFu : uint32_t(((hi - P) << 32) / N);
            }
            P += N;
          }
        }
        share[tid] = s;
      }
      __syncthreads();
      const FastShare sh = share[wv];
      const uint32_t j0 = wave::uniform(sh.j0), j1 = wave::uniform(sh.j1);
      const uint32_t f_lo = wave::uniform(sh.f_lo), f_hi = wave::uniform(sh.f_hi);
      // the threshold in 16-bit units: docs whose exact score reaches bin `bs` have A >= thr
      uint32_t thr = 1u;
      if (bs) {
        const double edge = double(bs) / double(qd.bin_scale) * double(qd.s16) * (1.0 - 1e-6);
        const double t = edge - double(kFastErr * n_terms);
        thr = t > 1.0 ? uint32_t(t) : 1u;
      }
      const uint32_t thr_k = (0x8000u - (thr < 0x8000u ? thr : 0x8000u)) * 0x00010001u;   // half + K: bit 15 set <=> half >= thr
      for (uint32_t u = 0; u < ntile; ++u) {
        // ---- accumulate this wavefront's entries of tile u
        for (uint32_t j = j0; j <= j1; ++j) {   // (wave-uniform; one or two terms as a rule)
          const uint32_t a0 = wave::uniform(rng[(2u * u) * kMaxTerms + j]);
          const uint32_t b1 = wave::uniform(rng[(2u * u + 1u) * kMaxTerms + j]);
          const uint32_t a2 = wave::uniform(rng[(2u * u + 2u) * kMaxTerms + j]);
          const uint32_t n = a2 - a0;
          const uint32_t lo = a0 + (j == j0 ? fast_cut(n, f_lo) : 0u);
          const uint32_t hi = a0 + (j == j1 ? fast_cut(n, f_hi) : n);
          if (hi <= lo) continue;
          const JoinTerm jt = jts[j];
          const uint64_t fent = wave::uniform64(jt.entries) + uint64_t(args->fast_delta);
          const uint32_t csq = wave::uniform(jt.pad[0]);
          const uint32_t mid = b1 < lo ? lo : (b1 > hi ? hi : b1);
          if (mid > lo) fast_run(smem, fent + 4ull * lo, mid - lo, csq, 0u, lane);
          if (hi > mid) fast_run(smem, fent + 4ull * mid, hi - mid, csq, 16u, lane);
        }
        __syncthreads();   // B1: every accumulation of tile u has landed
        // ---- epilogue: 12 words (24 docs) per lane, read AND cleared by LDS exchanges
        const uint32_t doc0 = kDocMin + (tile0 + u) * kFastTile;
        auto candidate = [&](uint32_t word, uint32_t w) {   // rare
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t a = h ? (w >> 16) : (w & 0xFFFFu);
            if (a >= thr) {
              const uint64_t key = (uint64_t(a) << 32) | (doc0 + word + (h ? kJoinTile : 0u));
              const uint32_t slot = atomicAdd(ncand, 1u);
              if (slot < kJoinCands) {
                lc[slot] = key;
              } else {   // rarer: more candidates in one chunk than staging slots
                const uint32_t g = atomicAdd(&args->cand_count[q], 1u);
                if (g < cap) args->cands[uint64_t(q) * cap + g] = key;
              }
            }
          }
        };
        for (uint32_t i = tid * 4u; i < kJoinTile; i += blockDim.x * 4u) {
          uint32_t v[4];
          wave::lds_take4(smem, FastOff::acc + i * 4u, v);
          uint32_t top = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            hit_pk = wave::pk_add_u16(hit_pk, wave::pk_min_u16(v[k], 0x00010001u));
            top = wave::pk_max_u16(top, v[k]);
          }
          if ((wave::pk_add_u16(top, thr_k) & 0x80008000u) != 0u) {
#pragma unroll
            for (int k = 0; k < 4; ++k) candidate(i + uint32_t(k), v[k]);
          }
        }
        __syncthreads();   // B2: accumulators are clear again
      }
    }
    // ---- chunk hand-over (as k_join_score): flush the PREVIOUS chunk's staged candidates,
    // reserve slots for this chunk's, publish the matches
    if (tid == 0) {
      vars[kJPendQ] = pend_q;
      vars[kJPendBase] = pend_base;
      vars[kJPendN] = pend_n;
      uint32_t g = group;
      vars[kJChunk] = pull(group, next_raw, g);
      vars[kJGroup] = g;
    }
    uint32_t my_hits = wave::reduce_add((hit_pk & 0xFFFFu) + (hit_pk >> 16));
    if (lane == 0 && my_hits)
      atomicAdd(&args->hits[q], static_cast<unsigned long long>(my_hits));
    __syncthreads();
    {
      const uint32_t pn = vars[kJPendN];
      if (pn) {
        const uint32_t pq = vars[kJPendQ], gbase = vars[kJPendBase];
        const uint64_t* pl = lcand + (parity ^ 1u) * kJoinCands;
        uint64_t* out = args->cands + uint64_t(pq) * cap;
        for (uint32_t i = tid; i < pn; i += blockDim.x) {
          const uint32_t g = gbase + i;
          if (g < cap) out[g] = pl[i];
        }
      }
    }
    chunk = wave::uniform(vars[kJChunk]);
    group = wave::uniform(vars[kJGroup]);
    if (tid == 0) {
      const uint32_t raw = *ncand;
      pend_n = raw < kJoinCands ? raw : kJoinCands;
      pend_q = q;
      pend_base = pend_n ? atomicAdd(&args->cand_count[q], pend_n) : 0u;
    }
    __syncthreads();   // everyone has read the hand-over words and the previous staging buffer
    parity ^= 1u;
    if (tid == 0) vars[kJNc + parity] = 0u;
  }
  if (tid == 0) {
    vars[kJPendQ] = pend_q;
    vars[kJPendBase] = pend_base;
    vars[kJPendN] = pend_n;
  }
  __syncthreads();
  {
    const uint32_t pn = vars[kJPendN];
    if (pn) {
      const uint32_t pq = vars[kJPendQ], gbase = vars[kJPendBase];
      const uint64_t* pl = lcand + (parity ^ 1u) * kJoinCands;
      uint64_t* out = args->cands + uint64_t(pq) * cap;
      for (uint32_t i = tid; i < pn; i += blockDim.x) {
        const uint32_t g = gbase + i;
        if (g < cap) out[g] = pl[i];
      }
    }
  }
}

// ---------------------------------------------------------------- rescore --

// The fixed-point contribution of one exact entry to its doc's score: what join_post adds
// (table row where every frequency of the term has one, else the general expression).
__device__ __forceinline__ uint32_t join_fixed(const float* caches, uint32_t e, float cs, uint32_t mode) {
  const uint32_t tabofs = mode & kJoinTabMask;
  const int form = join_form(mode);
  if (form == kJTable) {
    const float t = caches[((e & 0xFFFFu) | tabofs) >> 2];
    return static_cast<uint32_t>(wave::fma(cs, t, 1.f));
  }
  const float t = caches[((e & 0x3FCu) | tabofs) >> 2];
  const float tf = static_cast<float>((e >> 10) & kJoinTfMax);
  const float scaled = (form == kJSqrt) ? wave::fast_sqrt(tf) * cs * t
                                        : wave::fma(-cs, wave::fast_rcp(wave::fma(tf, t, 1.f)), cs);
  return static_cast<uint32_t>(scaled) | 1u;
}

struct RescoreArgs {
  const uint32_t* units;       // the launch's units
  const DevQuery* queries;
  const DevQTerm* qterms;
  const JoinTerm* jterms;
  const uint32_t* bstar;
  const uint64_t* acands;      // [unit][cap] k_join_fast's candidates
  const uint32_t* acand_count;
  uint64_t* cands;             // [unit][cap] exact keys for k_select
  uint32_t* cand_count;
  uint32_t cand_cap;
};

// One workgroup per unit.  (1) A coarse histogram of the candidates' approximate scores finds a
// lower bound L of the k-th largest; a doc of the exact top k has A >= L - kFastKeep * n (any k
// docs with A > A_d + 5 n have exact scores above doc d's: |A - X| < 3 n from below, 2 n from
// above; one more n for the fixed-point sums' own rounding) — everything else is dropped unseen.
// (2) The kept docs' postings are looked up in the exact entry streams (binary search inside the
// doc's tile, one thread per (doc, term)) and summed in fixed point exactly as join.h does; docs
// whose exact score reaches the threshold bin become candidates of k_select — the same set, bit
// for bit, the one-pass kernel emits among those that can still matter.
__global__ void __launch_bounds__(kTileThreadsMax)
k_join_rescore(const RescoreArgs* __restrict__ args) {
  RT_DYN_SMEM(smem);
  // LDS: qts | caches (build_tables) | hist / acc | docs
  DevQTerm* qts = reinterpret_cast<DevQTerm*>(smem);
  float* caches = reinterpret_cast<float*>(smem + sizeof(DevQTerm) * kMaxTerms);
  uint32_t* racc = reinterpret_cast<uint32_t*>(caches + 256u * kTableRows);   // [kFastRound] (first: hist[kFastBins])
  uint32_t* rdoc = racc + kFastRound;                                          // [kFastRound]
  __shared__ uint32_t s_n, s_cut, s_out;
  const uint32_t tid = threadIdx.x;
  const uint32_t q = args->units[blockIdx.x];
  const DevQuery qd = args->queries[q];
  const uint32_t cap = args->cand_cap;
  const uint32_t n_raw = args->acand_count[q];
  if (n_raw > cap) {   // the approximate candidates did not fit: k_select reports the overflow
    if (tid == 0) args->cand_count[q] = n_raw;   // (what the host grows the buffers to)
    return;
  }
  const uint32_t n = n_raw;
  const uint64_t* src = args->acands + uint64_t(q) * cap;
  const uint32_t n_terms = qd.n_terms;
  if (tid < n_terms) qts[tid] = args->qterms[qd.first_term + tid];
  for (uint32_t i = tid; i < kFastBins; i += blockDim.x) racc[i] = 0u;
  if (tid == 0) { s_cut = 0u; s_out = 0u; }
  __syncthreads();
  {
    JoinSm sm;
    sm.qts = qts;
    sm.caches = caches;
    build_tables(sm, qd.n_caches, n_terms);
  }
  if (n > qd.k) {
    for (uint32_t i = tid; i < n; i += blockDim.x)
      atomicAdd(&racc[uint32_t(src[i] >> 32) >> 5], 1u);
    __syncthreads();
    if (tid < 64u) {   // suffix search over 1024 bins: lane L owns the 16 bins of chunk 63 - L
      const uint32_t c = 63u - tid;
      uint32_t s = 0;
      for (uint32_t i = 0; i < kFastBins / 64u; ++i) s += racc[c * (kFastBins / 64u) + i];
      const uint32_t incl = wave::inclusive_scan(s);
      const uint64_t reach = wave::ballot(incl >= qd.k);
      if (reach) {
        const int srcl = __builtin_ctzll(reach);
        const uint32_t above = wave::bcast(incl - s, srcl);
        const uint32_t cc = 63u - uint32_t(srcl);
        if (tid == 0) {
          uint32_t cum = above, bin = cc * (kFastBins / 64u);
          for (int i = int(kFastBins / 64u) - 1; i >= 0; --i) {
            cum += racc[cc * (kFastBins / 64u) + uint32_t(i)];
            if (cum >= qd.k) { bin = cc * (kFastBins / 64u) + uint32_t(i); break; }
          }
          const uint32_t edge = bin << 5, slack = kFastKeep * n_terms;
          s_cut = edge > slack ? edge - slack : 0u;
        }
      }
    }
  }
  __syncthreads();
  const uint32_t cut = s_cut;
  const uint32_t bs = args->bstar[q];
  uint64_t* out = args->cands + uint64_t(q) * cap;
  // rounds of kFastRound candidates (of which the kept ones are re-scored)
  for (uint32_t r0 = 0; r0 < n; r0 += kFastRound) {
    __syncthreads();
    if (tid == 0) s_n = 0u;
    __syncthreads();
    const uint32_t r1 = r0 + kFastRound < n ? r0 + kFastRound : n;
    for (uint32_t i = r0 + tid; i < r1; i += blockDim.x) {
      const uint64_t key = src[i];
      if (uint32_t(key >> 32) >= cut) {
        const uint32_t slot = atomicAdd(&s_n, 1u);
        rdoc[slot] = uint32_t(key);
        racc[slot] = 0u;
      }
    }
    __syncthreads();
    const uint32_t kept = s_n;
    // one thread per (kept doc, term)
    for (uint32_t w = tid; w < kept * n_terms; w += blockDim.x) {
      const uint32_t c = w / n_terms, j = w % n_terms;
      const uint32_t doc = rdoc[c];
      const JoinTerm jt = args->jterms[qd.first_term + j];
      const uint32_t tile = (doc - kDocMin) / kJoinTile, idx = (doc - kDocMin) % kJoinTile;
      const uint32_t* bnd = reinterpret_cast<const uint32_t*>(jt.bounds);
      const uint32_t* ent = reinterpret_cast<const uint32_t*>(jt.entries);
      uint32_t a = bnd[tile], b = bnd[tile + 1u];
      while (a < b) {   // lower_bound of idx among the tile's entries (sorted by doc)
        const uint32_t m = (a + b) >> 1;
        if ((ent[m] >> 18) < idx) a = m + 1u; else b = m;
      }
      if (a < bnd[tile + 1u]) {
        const uint32_t e = ent[a];
        if ((e >> 18) == idx) atomicAdd(&racc[c], join_fixed(caches, e, jt.cs, jt.mode));
      }
    }
    __syncthreads();
    for (uint32_t c = tid; c < kept; c += blockDim.x) {
      const uint32_t f = racc[c];
      const float v = f <= kMaxTerms ? 0.f : from_fixed<uint32_t>(f, qd.fx_inv);
      if (f && score_bin(v, qd.bin_scale) >= bs) {
        const uint32_t slot = atomicAdd(&s_out, 1u);
        out[slot] = make_key(v, rdoc[c]);   // (slot < n <= cap)
      }
    }
  }
  __syncthreads();
  if (tid == 0) args->cand_count[q] = s_out;
}

constexpr size_t kRescoreSmem = sizeof(DevQTerm) * kMaxTerms + 4u * 256u * kTableRows + 8u * kFastRound;

}  // namespace irs_hip
