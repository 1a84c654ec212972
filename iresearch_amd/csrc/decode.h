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

// The two 8-byte words lane `lane` needs from a block payload to extract its
// values 2*lane and 2*lane+1 (issued early so the loads overlap other work).
struct RawPair {
  uint64_t a, b;
};

template<int LAYOUT>
__device__ __forceinline__ RawPair raw_load(const uint8_t* payload, uint32_t bits,
                                            unsigned lane) {
  RawPair r;
  if (bits == 0) {  // ALL_EQUAL: the vint value follows the header byte
    r.a = wave::load_u64(payload);
    r.b = 0;
  } else if (LAYOUT == kSimd4) {
    // value j = 4r + l sits in SSE lane l at bit r*bits of that lane's stream;
    // stream word k of lane l is u32 index 4k + l.  j = 2*lane, 2*lane+1 share
    // r = lane>>1 and are the adjacent lanes l0, l0+1.
    const uint32_t r_ = lane >> 1, l0 = (lane & 1u) << 1;
    const uint32_t k = (r_ * bits) >> 5;
    const uint8_t* p = payload + 4u * (4u * k + l0);
    r.a = wave::load_u64(p);       // words (k, l0), (k, l0+1)
    r.b = wave::load_u64(p + 16);  // words (k+1, l0), (k+1, l0+1)
  } else {
    // one little-endian bitstream, value j at bit j*bits
    const uint32_t w = (2u * lane * bits) >> 5;
    const uint8_t* p = payload + 4u * w;
    r.a = wave::load_u64(p);      // words w, w+1
    r.b = wave::load_u64(p + 4);  // words w+1, w+2
  }
  return r;
}

template<int LAYOUT>
__device__ __forceinline__ void raw_extract(const RawPair& r, uint32_t bits, unsigned lane,
                                            uint32_t& v0, uint32_t& v1) {
  const uint32_t mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
  if (LAYOUT == kSimd4) {
    const uint32_t s = ((lane >> 1) * bits) & 31u;
    const uint64_t a = (uint64_t(uint32_t(r.b)) << 32) | uint32_t(r.a);
    const uint64_t b = (r.b & 0xFFFFFFFF00000000ull) | (r.a >> 32);
    v0 = uint32_t(a >> s) & mask;
    v1 = uint32_t(b >> s) & mask;
  } else {
    const uint32_t s = (2u * lane * bits) & 31u;
    v0 = uint32_t(r.a >> s) & mask;
    const uint32_t s1 = s + bits;  // <= 63
    v1 = (s1 < 32u ? uint32_t(r.a >> s1) : uint32_t(r.b >> (s1 - 32u))) & mask;
  }
}

// Values 2*lane and 2*lane+1 of a packed block payload of `bits` (> 0) bits/value.
template<int LAYOUT>
__device__ __forceinline__ void unpack_pair(const uint8_t* payload, uint32_t bits,
                                            unsigned lane, uint32_t& v0,
                                            uint32_t& v1) {
  const RawPair r = raw_load<LAYOUT>(payload, bits, lane);
  raw_extract<LAYOUT>(r, bits, lane, v0, v1);
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
