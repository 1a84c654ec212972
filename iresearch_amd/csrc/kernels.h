// kernels.h — the CDNA4 kernels of the query path.  No MFMA on purpose:
// nothing here is a dense contraction; the work is byte/integer streaming
// bounded by HBM bandwidth (DESIGN.md "Kernels").
//
//   k_build_directory  segment open: walk block headers, record per-block
//                      offset / last doc / bit widths (no decoded data kept)
//   k_decode_term      bulk decode of one posting list (bit-exact test surface)
//   k_bit_union        doc bitset of many posting lists (postings_reader::bit_union)
//   k_plan             per (query, term): first block of every doc tile
//   (score.h:          work-item lists per doc tile, k_pilot, k_score)
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

enum : uint32_t {
  kStatusCorrupt = 1u,   // malformed block header / out-of-bounds offset
  kStatusOverflow = 2u,  // candidate buffer exhausted
  kStatusUnderflow = 4u, // an estimated threshold left fewer than k candidates
  kStatusWandFraming = 8u, // k_wand_skip0: skip entries do not line up with the block directory
};

// ------------------------------------------------------------- directory --

// 16-byte units a block occupies in the packed-payload image (DevSegment::pk): blocks whose
// doc part is 1..31-bit packed and whose freq part is 1..31-bit packed or ALL-EQUAL (nothing
// stored for it: rare terms mostly have tf == 1 throughout a block).  0 = not in the image.
__device__ __forceinline__ uint32_t pk_units(uint32_t dbits, uint32_t fbits) {
  return ((dbits - 1u) <= 30u && fbits <= 31u) ? dbits + fbits : 0u;
}
// both parts packed (what the position decoders of phrase.h read from the image)
__device__ __forceinline__ bool pk_both(uint32_t dbits, uint32_t fbits) {
  return (dbits - 1u) <= 30u && (fbits - 1u) <= 30u;
}

// ---- the header chain, by pointer doubling ---------------------------------------------------
// Block headers of the 1_x formats form a chain: the header byte `bits` at p is followed by
// 16 * bits payload bytes (1..32) or, for ALL_EQUAL (0), by one vint (bitpack.hpp:60-69, 159), and
// the next header by that.  Followed step by step the chain costs a dependent LDS read — and, for
// a lone wavefront, some hundred instruction issues — per header, and the longest list of the
// segment sets the duration of the kernel.  Here the chain is data parallel: for EVERY byte p of
// the window f[p] = "where the next header lies if p is one"; composing the table with itself
// gives f^2, f^4, ... (all 256 threads, a window's table in a few passes), and the orbit of the
// start doubles with it: pos[i + 2^k] = f^(2^k)[pos[i]].  log2(headers in the window) rounds.
constexpr uint32_t kChainThreads = 1024;           // 16 wavefronts: a workgroup of a long list has its
                                                   // CU to itself — one wavefront per SIMD issues an
                                                   // instruction every 10-14 cycles, four hide each other
constexpr uint32_t kChainWindow = 8192;            // bytes of the stream staged per window
constexpr uint32_t kChainCap = 1024;               // chain links listed per window at most
constexpr uint32_t kChainEnd = kChainWindow + 1;   // "the block does not end inside the window"
constexpr uint32_t kChainBad = kChainWindow + 2;   // malformed: bits > 32, or past the end of the file
constexpr uint32_t kLinkNone = 0xFFFFFFFEu;        // a link function's verdicts: nothing starts here
constexpr uint32_t kLinkBad = 0xFFFFFFFFu;         //   / malformed
struct ChainTables {
  uint16_t f[2][kChainWindow + 4];   // [0, kChainWindow]: positions (kChainWindow = the byte
                                     // behind a full window); then the two sentinels
  uint16_t pos[2 * kChainCap + 2];   // the orbit of the start
  uint32_t n;
};
struct alignas(16) ChainLine {
  uint64_t lo, hi;
};

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

// The link of the block streams: a header byte and what it stands in front of.
struct block_link {
  const uint8_t* win;
  __device__ __forceinline__ uint32_t operator()(uint32_t p) const {
    const uint32_t bits = win[p];
    if (bits > 32u) return kLinkBad;
    if (bits) return p + 1u + 16u * bits;
    uint32_t len;
    (void)vint_bytes(win + p + 1, &len);
    return p + 1u + len;
  }
};

// All kChainThreads threads of a workgroup.  win[0, lim) holds the stream from a 16-byte boundary
// (readable 64 bytes further), `room` bytes of it lie inside the file; the walk starts at
// `start` < lim and wants `want` (1..kChainCap) more headers.  Returns n = the headers whose
// blocks END inside the window and the file, hdr[i] = (offset << 8) | bits for i < n, and
// *next = the offset the walk goes on from.  *bad: a header above 32 or a block past the end of
// the file was met right behind the n-th header.  Ends with a barrier.
// `link(p)` = where the next link lies if one starts at p (or kLinkBad / kLinkNone); block_link below is the
// block headers', k_wand_skip0 has the skip entries'.
template<typename LINK>
__device__ inline uint32_t chain_orbit(const uint8_t* win, ChainTables& T, uint32_t start,
                                       uint32_t lim, uint64_t room, uint32_t want, uint32_t* hdr,
                                       uint32_t* next, uint32_t* bad, LINK link) {
  const uint32_t tid = threadIdx.x;
  const uint32_t limit = room < lim ? uint32_t(room) : lim;
  const bool inside = lim <= room;   // the file goes on behind the window
  // f^1; the entries behind the staged bytes and the sentinels are fixed points of every power
  for (uint32_t p0 = tid; p0 <= kChainBad; p0 += 8u * kChainThreads) {
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) {
      const uint32_t p = p0 + k * kChainThreads;
      if (p > kChainBad) continue;
      uint32_t v = p == kChainBad ? kChainBad : kChainEnd;
      if (p < lim) {
        const uint32_t nxt = link(p);
        v = nxt == kLinkBad ? kChainBad : nxt == kLinkNone ? kChainEnd
            : nxt <= limit ? nxt : inside ? kChainEnd : kChainBad;
      }
      T.f[0][p] = uint16_t(v);
      if (p >= lim) T.f[1][p] = uint16_t(v);
    }
  }
  if (tid == 0) {
    T.pos[0] = uint16_t(start);
    T.n = 0;
  }
  __syncthreads();
  uint32_t cur = 0, have = 1;   // pos[0, have) is known, f[cur] = f^have
  while (have <= want) {
    const uint16_t* f = T.f[cur];
    for (uint32_t i = tid; i < have; i += kChainThreads) T.pos[i + have] = f[T.pos[i]];
    uint16_t* g = T.f[cur ^ 1u];
    // (eight entries a thread at a time: the two dependent reads of each overlap with the others'
    // — a workgroup of a long list has its SIMDs to itself, nothing else hides the LDS latency)
    for (uint32_t p0 = tid; p0 < lim; p0 += 8u * kChainThreads) {
      uint32_t a[8];
#pragma unroll
      for (uint32_t k = 0; k < 8; ++k) {
        const uint32_t p = p0 + k * kChainThreads;
        a[k] = f[p < lim ? p : kChainEnd];
      }
#pragma unroll
      for (uint32_t k = 0; k < 8; ++k) a[k] = f[a[k]];
#pragma unroll
      for (uint32_t k = 0; k < 8; ++k) {
        const uint32_t p = p0 + k * kChainThreads;
        if (p < lim) g[p] = uint16_t(a[k]);
      }
    }
    __syncthreads();
    cur ^= 1u;
    have *= 2u;
    if (T.pos[have - 1u] > kChainWindow) break;   // the orbit has left the window
  }
  // header i is taken iff its block ends inside: pos[i + 1] is a position (a prefix of the orbit)
  const uint32_t known = have - 1u < want ? have - 1u : want;
  uint32_t mine = 0;
  for (uint32_t i = tid; i < known; i += kChainThreads) {
    const uint32_t at = T.pos[i];
    if (T.pos[i + 1u] <= kChainWindow) {
      hdr[i] = (at << 8) | win[at];
      ++mine;
    }
  }
  mine = wave::reduce_add(mine);
  if ((tid & 63u) == 0 && mine) atomicAdd(&T.n, mine);
  __syncthreads();
  const uint32_t n = T.n;
  *next = T.pos[n];
  *bad = n < want && T.pos[n + 1u] == kChainBad ? 1u : 0u;
  return n;
}

// One workgroup per term walks the term's full blocks front to back: header byte -> payload
// size (bitpack::skip_block32, bitpack.hpp:60-69); the block's last doc is base + sum(deltas).
// Thread 0 then decodes the vint tail (formats_10.cpp:1765-1792) into the per-term tail tables.
// Only the header chain is serial (header -> size -> next header), and the longest list of the
// segment sets the kernel's duration (10 M docs: 78 k blocks), so per window of kDirWindow
// bytes staged in LDS
//   1. all threads list the window's headers (chain_orbit above): (offset, bit width) each;
//   2. the four wavefronts decode the listed blocks' delta sums side by side;
//   3. wavefront 0 turns the sums into last docs (a scan), everyone writes the rows coalesced.
// Round 4 did all of it inside the chain, one wavefront per term: 0.89 us per block, 69.5 ms.
constexpr uint32_t kDirWindow = kChainWindow;
constexpr uint32_t kDirList = 256;     // blocks listed per window at most (2 headers each)
struct alignas(16) DirLine {
  uint64_t lo, hi;
};

template<int LAYOUT>
__global__ void __launch_bounds__(kChainThreads)
k_build_directory(DevSegment seg, DevTerm* terms, uint32_t* blk_off,
                  uint32_t* blk_last, uint16_t* blk_bits, uint32_t* blk_units, BlkDir* blk_dir,
                  uint32_t* blk_term, uint32_t* tail_docs, uint32_t* tail_freqs, uint32_t* status) {
  // (+64: the unpackers read whole 8-byte words, up to 24 bytes past a payload's end)
  __shared__ __attribute__((aligned(16))) uint8_t win[kDirWindow + 64];
  __shared__ ChainTables s_chain;
  __shared__ __attribute__((aligned(16))) uint32_t s_sum[kDirList];   // delta sums, then last docs
  __shared__ uint32_t s_tf[kDirList];    // the block's frequency bound
  __shared__ uint32_t s_hdr[2 * kDirList];   // headers: (offset in the window << 8) | bits
  __shared__ uint64_t s_cur;
  __shared__ uint32_t s_n, s_bad, s_tfb;
  const unsigned tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t term = blockIdx.x;
  if (term >= seg.num_terms) return;
  const DevTerm t = terms[term];
  if (t.docs_count == 0) return;
  if (t.docs_count == 1) {  // single_doc_iterator, formats_10.cpp:1876-1890
    if (tid == 0) {
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
  const uint64_t staged = seg.doc_len + kPadBytes;  // the device copy ends with zero padding
  uint32_t b = 0;
  while (b < t.nblk && !bad) {
    if (cur + 2 > seg.doc_len) { bad = true; break; }
    // the window: from the 16-byte line holding `cur`
    const uint64_t win_lo = cur & ~uint64_t(15);
    uint64_t bytes = staged - win_lo;
    if (bytes > kDirWindow) bytes = kDirWindow;
    bytes &= ~uint64_t(15);
    for (uint32_t o = tid * 16u; o < bytes; o += kChainThreads * 16u)
      *reinterpret_cast<DirLine*>(win + o) = *reinterpret_cast<const DirLine*>(seg.doc + win_lo + o);
    __syncthreads();
    const uint32_t per = seg.has_freq ? 2u : 1u;   // headers per block
    {
      // 1. the chain: the headers whose blocks end inside the window (and the file)
      const uint32_t left = (t.nblk - b) * per;
      const uint32_t cap = per * kDirList;
      uint32_t next, wbad;
      uint32_t nh = chain_orbit(win, s_chain, uint32_t(cur - win_lo), uint32_t(bytes),
                                seg.doc_len - win_lo, left < cap ? left : cap, s_hdr, &next, &wbad,
                                block_link{win});
      if (per == 2u && (nh & 1u)) {   // a doc header without its freq header: the next window's
        --nh;
        next = s_hdr[nh] >> 8;
        wbad = 0;   // (what lies behind the freq header is that window's to judge)
      }
      // a window from `cur` holds at least one whole block of a valid list (<= 1026 bytes)
      if (nh == 0) wbad = 1;
      if (tid == 0) {
        s_n = nh / per;
        s_bad = wbad;
        s_cur = win_lo + next;
      }
    }
    __syncthreads();
    const uint32_t n = s_n;
    // 2. delta sums and frequency bounds of the listed blocks, a wavefront per block
    for (uint32_t i = wv; i < n; i += kChainThreads / 64u) {
      const uint32_t dh = s_hdr[per * i];
      const uint32_t dbits = dh & 0xFFu;
      uint32_t x0, x1;
      (void)read_block_pair<LAYOUT>(win + (dh >> 8), dbits, lane, x0, x1);
      const uint32_t sum = wave::reduce_add(x0 + x1);
      uint32_t tf = 0;
      if (seg.has_freq) {
        const uint32_t fh = s_hdr[per * i + 1u];
        const uint32_t fbits = fh & 0xFFu;
        if (fbits) {
          tf = fbits >= 32 ? 0xFFFFFFFFu : ((1u << fbits) - 1u);
        } else {
          uint32_t len;
          tf = vint_from(wave::load_u64(win + (fh >> 8) + 1), &len);
        }
      }
      if (lane == 0) {
        s_sum[i] = sum;
        s_tf[i] = tf;
      }
    }
    __syncthreads();
    // 3. last docs: base + running sum.  Doc ids ascend and stay inside the segment (a doc
    // block of valid data ends at least 127 docs behind the previous one): a zero sum or a last
    // doc beyond the segment is corrupt data.  Exact in 64 bits from two 32-bit scans (a sum
    // that passed the first test is < 2^32, 256 of them stay below 2^24 per half).
    if (wv == 0) {
      uint32_t v[4], lo = 0, hi = 0, mx = 0;
      bool wbad = false;
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t i = 4u * lane + k;
        v[k] = i < n ? s_sum[i] : 0u;
        if (i < n) {
          if (v[k] == 0 || v[k] > seg.num_docs) wbad = true;
          const uint32_t tf = s_tf[i];
          mx = tf > mx ? tf : mx;
        }
        lo += v[k] & 0xFFFFu;
        hi += v[k] >> 16;
      }
      uint32_t ilo = lo, ihi = hi;
      wave::inclusive_scan2(ilo, ihi);
      uint64_t run = uint64_t(base) + (uint64_t(ihi - hi) << 16) + (ilo - lo);
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t i = 4u * lane + k;
        run += v[k];
        if (i < n) {
          if (run > seg.num_docs) wbad = true;
          s_sum[i] = uint32_t(run);
        }
      }
      mx = wave::reduce_max(mx);
      const bool any_bad = wave::ballot(wbad) != 0;
      if (lane == 0) {
        s_tfb = mx;
        if (any_bad) s_bad = 1;
      }
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += kChainThreads) {
      const uint64_t e = t.dir_off + b + i;
      const uint32_t dh = s_hdr[per * i];
      const uint32_t dbits = dh & 0xFFu, fbits = seg.has_freq ? s_hdr[per * i + 1u] & 0xFFu : 0u;
      const uint32_t bits = dbits | (fbits << 8);
      const uint32_t off = uint32_t(win_lo + (dh >> 8) - t.doc_start);
      blk_off[e] = off;
      blk_last[e] = s_sum[i];
      blk_bits[e] = uint16_t(bits);
      blk_units[e] = pk_units(dbits, fbits);
      blk_term[e] = term;
      BlkDir d;
      d.off = off;
      d.prev_last = i ? s_sum[i - 1] : base;
      d.aoff = 0;   // k_dir_aoff fills it in once the image offsets are known
      d.bits = bits;
      blk_dir[e] = d;
    }
    if (n) base = s_sum[n - 1];
    tfb = s_tfb > tfb ? s_tfb : tfb;
    bad = s_bad != 0;
    cur = s_cur;
    b += n;
    __syncthreads();   // the window and the lists are rewritten next
  }
  if (tid == 0) {
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

// BlkDir::aoff <- blk_aoff (after the exclusive scan turned block sizes into offsets)
__global__ void __launch_bounds__(kThreads)
k_dir_aoff(const uint32_t* blk_aoff, uint64_t n, BlkDir* blk_dir) {
  const uint64_t i = uint64_t(blockIdx.x) * kThreads + threadIdx.x;
  if (i < n) blk_dir[i].aoff = blk_aoff[i];
}

// The passes over every block of the segment (payload image, frequency sums, block maxima,
// posting-order norms) split their work by directory ROW: a wavefront takes rows gw, gw + all
// wavefronts, ...  A Zipfian index holds most of its blocks in a few lists — split by term
// (rounds 1-4: one workgroup per term once there are 2048 terms) the longest list's 78 k blocks
// went through 4 wavefronts while the chip idled: k_pack_payloads copied 550 MB in 15.5 ms.
__host__ __device__ inline uint32_t row_grid(uint64_t rows, uint32_t cus) {
  const uint64_t wgs = (rows + kWaves - 1) / kWaves;
  return uint32_t(wgs < uint64_t(cus) * 16u ? wgs : uint64_t(cus) * 16u);
}
#define IRS_FOR_ROWS(e, rows)                                                        \
  for (uint64_t e = uint64_t(blockIdx.x) * kWaves + (threadIdx.x >> 6); e < (rows);  \
       e += uint64_t(gridDim.x) * kWaves)
// where row e's block starts in `.doc`
__device__ __forceinline__ const uint8_t* row_block(const DevSegment& seg, uint64_t e) {
  return seg.doc + seg.terms[seg.blk_term[e]].doc_start + seg.blk_off[e];
}

// Copies the payloads of the decodable blocks into the packed-payload image.
__global__ void __launch_bounds__(kThreads)
k_pack_payloads(DevSegment seg, uint64_t rows, uint8_t* pk) {
  const unsigned lane = threadIdx.x & 63u;
  IRS_FOR_ROWS(e, rows) {
    const uint32_t bits = seg.blk_bits[e];
    const uint32_t dbits = bits & 0xFFu, fbits = bits >> 8;
    if (!pk_units(dbits, fbits)) continue;
    const uint8_t* blk = row_block(seg, e);
    uint64_t* dst = reinterpret_cast<uint64_t*>(pk + (uint64_t(seg.blk_aoff[e]) << 4));
    // 8-byte pieces: 2*dbits of the doc payload (after its header byte), then
    // 2*fbits of the freq payload (after the second header byte)
    for (uint32_t i = lane; i < 2u * (dbits + fbits); i += 64) {
      const uint8_t* src = i < 2u * dbits ? blk + 1u + 8u * i : blk + 2u + 8u * i;
      dst[i] = wave::load_u64(src);
    }
  }
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
  uint32_t set;    // which of the call's bitsets (irs_hip_bit_union_counts: several; else 0)
  uint32_t pad;
};

template<int LAYOUT>
__global__ void __launch_bounds__(kThreads)
k_bit_union(DevSegment seg, const UnionWg* wgs, uint32_t* sets32, uint64_t n_bits) {
  const unsigned lane = threadIdx.x & 63u;
  const UnionWg wg = wgs[blockIdx.x];
  uint32_t* set32 = sets32 + uint64_t(wg.set) * (n_bits >> 5);
  const DevTerm t = seg.terms[wg.term];
  const uint32_t* dead = seg.dead;   // (deleted docs never enter the set: SegmentReaderImpl::mask)
  auto mark = [&](uint32_t doc) {
    if (doc < n_bits && !(dead && doc_dead(dead, doc))) atomicOr(&set32[doc >> 5], 1u << (doc & 31u));
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

// Population of every bitset of a call (irs_hip_bit_union_counts): one workgroup per set.
__global__ void __launch_bounds__(kThreads)
k_union_counts(const uint32_t* sets32, uint64_t words32, unsigned long long* counts) {
  __shared__ uint32_t s_part[kWaves];
  const uint32_t* set32 = sets32 + uint64_t(blockIdx.x) * words32;
  uint32_t n = 0;
  for (uint64_t i = threadIdx.x; i < words32; i += kThreads) n += uint32_t(__builtin_popcount(set32[i]));
  n = wave::reduce_add(n);
  if ((threadIdx.x & 63u) == 0) s_part[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long sum = 0;
    for (uint32_t w = 0; w < kWaves; ++w) sum += s_part[w];
    counts[blockIdx.x] = sum;
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
  if (qd.first_off == kNoPlan) return;   // joined posting streams: no plan tables
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

// ------------------------------------------------- shared decode helpers --

// legacy `Norm` column (norm.hpp:57-69): one float per doc, 1/sqrt(|doc|)
__device__ __forceinline__ float norm_legacy(const DevSegment& seg, uint32_t doc) {
  float v;
  __builtin_memcpy(&v, seg.norms + 4ull * (doc - seg.norm_min_doc), 4);
  return v;
}

__device__ __forceinline__ uint32_t norm_global(const DevSegment& seg, uint32_t doc) {
  // dense fixed-length column, big-endian values (columnstore2.cpp:736-740, norm.hpp:170-182)
  const uint8_t* p = seg.norms + uint64_t(seg.norm_width) * (doc - seg.norm_min_doc);
  if (seg.norm_width == 2) return (uint32_t(p[0]) << 8) | p[1];
  return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3];
}



// Payload words of values 2*lane, 2*lane+1 of a packed block, 1 <= bits <= 32, with no
// branch on `bits` (any other framing is re-read by the generic path): the
// same address arithmetic as raw_load.  Reads at most 16 bytes past the block.
template<int LAYOUT>
__device__ __forceinline__ void raw_load_packed(const uint8_t* payload, uint32_t bits,
                                                unsigned lane, uint64_t& a, uint64_t& b) {
  if (LAYOUT == kSimd4) {
    const uint32_t k = wave::mul24(lane >> 1, bits) >> 5;
    const uint32_t voff = 16u * k + ((lane & 1u) << 3);
    a = wave::load_u64(payload + voff);
    b = wave::load_u64(payload + voff + 16);
  } else {
    const uint32_t voff = (wave::mul24(lane << 1, bits) >> 5) << 2;
    a = wave::load_u64(payload + voff);
    b = wave::load_u64(payload + voff + 4);
  }
}

// The same through the GLOBAL address space (wave::gload_u64: base = a wave-uniform integer
// address): a payload pointer read out of the DevSegment record is a generic pointer and its
// loads FLAT ones, which also count on lgkmcnt — in a kernel that interleaves them with LDS
// operations (conj.h, phrase.h) every wait for an LDS result then drains the payload loads in
// flight as well.
template<int LAYOUT>
__device__ __forceinline__ void raw_load_packed_g(uint64_t payload, uint32_t bits, unsigned lane,
                                                  uint64_t& a, uint64_t& b) {
  if (LAYOUT == kSimd4) {
    const uint32_t k = wave::mul24(lane >> 1, bits) >> 5;
    const uint32_t voff = 16u * k + ((lane & 1u) << 3);
    a = wave::gload_u64(payload, voff);
    b = wave::gload_u64(payload, voff + 16u);
  } else {
    const uint32_t voff = (wave::mul24(lane << 1, bits) >> 5) << 2;
    a = wave::gload_u64(payload, voff);
    b = wave::gload_u64(payload, voff + 4u);
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
         uint32_t* out_count, uint32_t* status, uint32_t stage_cap, uint32_t sort_cap,
         const uint32_t* bstar, const uint32_t* min_bin /*null: no caller thresholds*/,
         const uint32_t* pruned /*[unit] != 0: block-max pruning skipped blocks or tiles*/,
         const float* min_score /*[unit] the caller's irs::score::Min, null: none*/,
         const uint32_t* group_of /*[unit] != 0: the unit shares its threshold with a group —
                                    "fewer than k" is checked for the group (k_group_check)*/) {
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
  // an estimated threshold (k_pilot) cut off docs that belong to the top k — unless the
  // threshold in force is the caller's own (irs::score::Min): fewer than k docs reach it
  const bool callers = min_bin && min_bin[q] != 0u && bstar[q] == min_bin[q];
  // (hits only counts evaluated docs: where pruning skipped some, "fewer than k" alone says
  // the threshold was too high — unless nothing more exists, which a sound re-run then shows)
  const bool grouped = group_of && group_of[q] != 0u;
  if (n < qd.k && (hits[q] > n || pruned[q]) && !callers && !grouped && tid == 0)
    atomicOr(status, kStatusUnderflow);
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
  // the caller's own threshold is exact: the kernels filtered by score BIN, the few docs that
  // share the threshold's bin but score below it go here (sorted: they are a suffix)
  uint32_t keep = kk;
  if (min_score && min_score[q] > 0.f) {
    if (tid == 0) sh_n = 0;
    __syncthreads();
    const float m = min_score[q];
    uint32_t mine = 0;
    for (uint32_t i = tid; i < kk; i += blockDim.x) mine += key_hit(keys[i]).score >= m ? 1u : 0u;
    if (mine) atomicAdd(&sh_n, mine);
    __syncthreads();
    keep = sh_n;
  }
  for (uint32_t i = tid; i < keep; i += blockDim.x)
    out[uint64_t(q) * k_max + i] = key_hit(keys[i]);
  if (tid == 0) out_count[q] = keep;
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
