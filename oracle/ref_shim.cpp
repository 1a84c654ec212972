// oracle/ref_shim.cpp — TEST INFRASTRUCTURE ONLY.
// C-callable doors into the REFERENCE's own compiled codec (built by
// oracle/Makefile from the sources where they lie under /root/reference; no
// reference source is copied).  Used in the build container to pin
// oracle/postings_oracle.c and to generate tests/golden/codec_*.bin.
#include <cstdint>

#include "utils/bit_packing.hpp"  // /root/reference/core
extern "C" {
#include "simdbitpacking.h"  // /root/reference/external/simdcomp/include
}

extern "C" {

// format_traits::pack_block / unpack_block — formats_10.cpp:96-116
void ref_pack_scalar(const uint32_t* in128, uint32_t bits, uint32_t* out) {
  for (uint32_t sb = 0; sb < 4; ++sb)
    irs::packed::pack_block(in128 + 32 * sb, out + bits * sb, bits);
}
void ref_unpack_scalar(const uint32_t* in, uint32_t bits, uint32_t* out128) {
  for (uint32_t sb = 0; sb < 4; ++sb)
    irs::packed::unpack_block(in + bits * sb, out128 + 32 * sb, bits);
}
uint32_t ref_at_scalar(const uint32_t* in, uint32_t i, uint32_t bits) {
  return irs::packed::at(in, i, bits);
}
// format_traits_sse4::pack_block / unpack_block — formats_10.cpp:4131-4142
void ref_pack_simd4(const uint32_t* in128, uint32_t bits, uint32_t* out) {
  ::simdpackwithoutmask(in128, reinterpret_cast<__m128i*>(out), bits);
}
void ref_unpack_simd4(const uint32_t* in, uint32_t bits, uint32_t* out128) {
  ::simdunpack(reinterpret_cast<const __m128i*>(in), out128, bits);
}
uint32_t ref_maxbits32(const uint32_t* begin, uint32_t n) {
  return irs::packed::maxbits32(begin, begin + n);
}

}  // extern "C"
