// decode.h — one wavefront decodes one 128-posting block: lane L owns values
// 2L and 2L+1.  Replaces, bit-exactly:
//   bitpack::read_block_impl32            core/utils/bitpack.hpp:149-177
//   format_traits::unpack_block           core/formats/formats_10.cpp:107-116
//     -> packed::unpack_block / fastunpack<N>  core/utils/bit_packing.cpp:644-843, 1807
//   format_traits_sse4::unpack_block      core/formats/formats_10.cpp:4138-4142
//     -> simdunpack                       external/simdcomp/src/simdbitpacking.c
//   doc_iterator::next() `doc += delta`   core/formats/formats_10.cpp:2107
// The serial prefix sum of the reference becomes one wavefront scan.
#pragma once
#include "types.h"
#include "wave.h"

namespace irs_hip {

// LEB128 (bytes_io<uint32_t>::vread, core/utils/bytes_utils.hpp:176-206) from
// the low bytes of `x`; *len = encoded length.
__device__ __forceinline__ uint32_t vint_from(uint64_t x, uint32_t* len) {
  uint32_t v = uint32_t(x) & 0x7Fu;
  uint32_t n = 1;
  if (x & 0x80u) {
    v |= (uint32_t(x >> 8) & 0x7Fu) << 7;
    n = 2;
    if (x & 0x8000u) {
      v |= (uint32_t(x >> 16) & 0x7Fu) << 14;
      n = 3;
      if (x & 0x800000u) {
        v |= (uint32_t(x >> 24) & 0x7Fu) << 21;
        n = 4;
        if (x & 0x80000000ull) {
          v |= (uint32_t(x >> 32) & 0x7Fu) << 28;
          n = 5;
        }
      }
    }
  }
  *len = n;
  return v;
}

// Values 2*lane and 2*lane+1 of a packed block payload of `bits` bits/value.
template<int LAYOUT>
__device__ __forceinline__ void unpack_pair(const uint8_t* payload, uint32_t bits,
                                            unsigned lane, uint32_t& v0,
                                            uint32_t& v1) {
  const uint32_t mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
  if (LAYOUT == kSimd4) {
    // value j = 4r + l sits in SSE lane l at bit r*bits of that lane's stream;
    // stream word k of lane l is u32 index 4k + l.  j = 2*lane, 2*lane+1 share
    // r = lane>>1 and are the adjacent lanes l0, l0+1.
    const uint32_t r = lane >> 1, l0 = (lane & 1u) << 1;
    const uint32_t bit = r * bits;
    const uint32_t k = bit >> 5, s = bit & 31u;
    const uint8_t* p = payload + 4u * (4u * k + l0);
    const uint64_t lo = wave::load_u64(p);        // words (k, l0), (k, l0+1)
    const uint64_t hi = wave::load_u64(p + 16);   // words (k+1, l0), (k+1, l0+1)
    const uint64_t a = (uint64_t(uint32_t(hi)) << 32) | uint32_t(lo);
    const uint64_t b = (hi & 0xFFFFFFFF00000000ull) | (lo >> 32);
    v0 = uint32_t(a >> s) & mask;
    v1 = uint32_t(b >> s) & mask;
  } else {
    // one little-endian bitstream, value j at bit j*bits
    const uint32_t bit = 2u * lane * bits;
    const uint32_t w = bit >> 5, s = bit & 31u;
    const uint8_t* p = payload + 4u * w;
    const uint64_t a = wave::load_u64(p);      // words w, w+1
    const uint64_t b = wave::load_u64(p + 4);  // words w+1, w+2
    v0 = uint32_t(a >> s) & mask;
    const uint32_t s1 = s + bits;              // <= 63
    v1 = (s1 < 32u ? uint32_t(a >> s1) : uint32_t(b >> (s1 - 32u))) & mask;
  }
}

// One framed block (header byte + payload): returns the two values of this
// lane and the encoded size.  bits == 0 is the ALL_EQUAL run (bitpack.hpp:159).
template<int LAYOUT>
__device__ __forceinline__ uint32_t read_block_pair(const uint8_t* blk, uint32_t bits,
                                                    unsigned lane, uint32_t& v0,
                                                    uint32_t& v1) {
  if (bits == 0) {
    uint32_t len;
    v0 = v1 = vint_from(wave::load_u64(blk + 1), &len);
    return 1u + len;
  }
  unpack_pair<LAYOUT>(blk + 1, bits, lane, v0, v1);
  return 1u + 16u * bits;
}

// Decodes doc block + freq block `blk` whose predecessor's last doc is `base`.
// d0/d1 receive ABSOLUTE doc ids of postings 2*lane, 2*lane+1.
template<int LAYOUT, bool FREQ>
__device__ __forceinline__ uint32_t decode_block(const uint8_t* blk, uint32_t dbits,
                                                 uint32_t fbits, uint32_t base,
                                                 unsigned lane, uint32_t& d0,
                                                 uint32_t& d1, uint32_t& f0,
                                                 uint32_t& f1) {
  uint32_t x0, x1;
  uint32_t size = read_block_pair<LAYOUT>(blk, dbits, lane, x0, x1);
  const uint32_t incl = wave::inclusive_scan(x0 + x1);
  d1 = base + incl;
  d0 = d1 - x1;
  if (FREQ) {
    size += read_block_pair<LAYOUT>(blk + size, fbits, lane, f0, f1);
  } else {
    f0 = f1 = 1;
  }
  return size;
}

}  // namespace irs_hip
