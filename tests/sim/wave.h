// tests/sim/wave.h — TEST INFRASTRUCTURE ONLY: portable (shuffle-based) versions of
// iresearch_amd/csrc/hip/wave.h for the CPU fiber emulator.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#define IRS_WAVE 64
#define IRS_WAVES_PER_SIMD(N)

namespace wave {

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

// Inclusive prefix sum across the 64 lanes of a wavefront.
__device__ __forceinline__ uint32_t inclusive_scan(uint32_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(v, d, 64);
    if (lane_id() >= unsigned(d)) v += up;
  }
  return v;
}

__device__ __forceinline__ uint32_t reduce_add(uint32_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ __forceinline__ uint32_t reduce_max(uint32_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const uint32_t o = __shfl_xor(v, d, 64);
    v = o > v ? o : v;
  }
  return v;
}
__device__ __forceinline__ uint32_t bcast(uint32_t v, int src_lane) {
  return __shfl(v, src_lane, 64);
}
__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
// wavefront-level rendezvous: every lane (fiber) arrives before any goes on
__device__ __forceinline__ void sync() { (void)__ballot(1); }

// LDS float accumulate without a returned value -> ds_add_f32
__device__ __forceinline__ void lds_add(float* p, float v) { atomicAdd(p, v); }

// Unaligned little-endian loads from the byte-granular `.doc` stream.
__device__ __forceinline__ uint64_t load_u64(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
__device__ __forceinline__ uint32_t load_u32(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}

// scalar / saddr-global loads of hip/wave.h: plain reads here
template<typename T>
__device__ __forceinline__ T sload(uint64_t addr) { return *reinterpret_cast<const T*>(addr); }
__device__ __forceinline__ uint64_t gload_u64(uint64_t base, uint32_t off) {
  uint64_t v;
  __builtin_memcpy(&v, reinterpret_cast<const uint8_t*>(base) + off, 8);
  return v;
}

// wave-uniform value -> scalar register on the GPU; identity here
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return v; }
__device__ __forceinline__ float uniform_f(float v) { return v; }
__device__ __forceinline__ uint64_t uniform64(uint64_t v) { return v; }
__device__ __forceinline__ void inclusive_scan2(uint32_t& a, uint32_t& b) {
  a = inclusive_scan(a);
  b = inclusive_scan(b);
}
__device__ __forceinline__ uint32_t read_lane(uint32_t v, uint32_t k) { return __shfl(v, int(k), 64); }
__device__ __forceinline__ uint32_t write_lane(uint32_t v, uint32_t x, uint32_t k) {
  return lane_id() == k ? x : v;
}
__device__ __forceinline__ float read_lane_f(float v, uint32_t k) { return __shfl(v, int(k), 64); }
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t s) {
  return uint32_t(((uint64_t(hi) << 32) | lo) >> (s & 31u));
}
__device__ __forceinline__ uint32_t bfe(uint32_t x, uint32_t bits) { return x & ((1u << bits) - 1u); }
__device__ __forceinline__ uint64_t undef64() { return 0; }
__device__ __forceinline__ void undef4(uint32_t (&v)[4]) { v[0] = v[1] = v[2] = v[3] = 0xDEADBEEFu; }
__device__ __forceinline__ uint32_t mul_hi(uint32_t a, uint32_t b) { return uint32_t((uint64_t(a) * b) >> 32); }
template<typename T>
__device__ __forceinline__ void count_nonzero4(uint32_t& acc, T a, T b, T c, T d) {
  acc += uint32_t(a != 0) + uint32_t(b != 0) + uint32_t(c != 0) + uint32_t(d != 0);
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
  const uint32_t lo = (a & 0xFFFFu) < (b & 0xFFFFu) ? (a & 0xFFFFu) : (b & 0xFFFFu);
  const uint32_t hi = (a >> 16) < (b >> 16) ? (a >> 16) : (b >> 16);
  return (hi << 16) | lo;
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
  const uint32_t lo = (a & 0xFFFFu) > (b & 0xFFFFu) ? (a & 0xFFFFu) : (b & 0xFFFFu);
  const uint32_t hi = (a >> 16) > (b >> 16) ? (a >> 16) : (b >> 16);
  return (hi << 16) | lo;
}
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) {
  return (((a >> 16) + (b >> 16)) << 16) | ((a + b) & 0xFFFFu);
}
__device__ __forceinline__ void count_nonzero_halves4(uint32_t& acc, uint32_t a, uint32_t b,
                                                      uint32_t c, uint32_t d) {
  const uint32_t one = 0x00010001u;
  acc = pk_add_u16(acc, pk_add_u16(pk_add_u16(pk_min_u16(a, one), pk_min_u16(b, one)),
                                   pk_add_u16(pk_min_u16(c, one), pk_min_u16(d, one))));
}
__device__ __forceinline__ uint32_t opaque(uint32_t v) { return v; }
__device__ __forceinline__ uint64_t opaque64(uint64_t v) { return v; }

// LDS "by absolute address" (hip/wave.h): here simply base + offset
__device__ __forceinline__ bool lds_is_at_zero(const unsigned char*) { return true; }
__device__ __forceinline__ uint32_t lds_u8(const unsigned char* base, uint32_t off) { return base[off]; }
__device__ __forceinline__ float lds_f32(const unsigned char* base, uint32_t off) {
  float v;
  __builtin_memcpy(&v, base + off, 4);
  return v;
}
__device__ __forceinline__ void lds_add(const unsigned char* base, uint32_t off, uint32_t v) {
  atomicAdd(reinterpret_cast<uint32_t*>(const_cast<unsigned char*>(base) + off), v);
}
__device__ __forceinline__ void lds_add(const unsigned char* base, uint32_t off, unsigned long long v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(const_cast<unsigned char*>(base) + off), v);
}
__device__ __forceinline__ void lds_read4(const unsigned char* base, uint32_t off, uint32_t (&v)[4]) {
  __builtin_memcpy(v, base + off, 16);
}
__device__ __forceinline__ void lds_zero4(unsigned char* base, uint32_t off) {
  __builtin_memset(base + off, 0, 16);
}
__device__ __forceinline__ void lds_take4(unsigned char* base, uint32_t off, uint32_t (&v)[4]) {
  __builtin_memcpy(v, base + off, 16);
  __builtin_memset(base + off, 0, 16);
}
__device__ __forceinline__ void lds_take4x2(unsigned char* base, uint32_t off0, uint32_t off1,
                                            uint32_t (&a)[4], uint32_t (&b)[4]) {
  lds_take4(base, off0, a);
  lds_take4(base, off1, b);
}
__device__ __forceinline__ void lds_take4x3(unsigned char* base, uint32_t off0, uint32_t off1, uint32_t off2,
                                            uint32_t (&a)[4], uint32_t (&b)[4], uint32_t (&c)[4]) {
  lds_take4(base, off0, a);
  lds_take4(base, off1, b);
  lds_take4(base, off2, c);
}
__device__ __forceinline__ uint32_t gload_u32(uint64_t base, uint32_t off) {
  uint32_t v;
  __builtin_memcpy(&v, reinterpret_cast<const uint8_t*>(base) + off, 4);
  return v;
}
__device__ __forceinline__ void gload_u32x4(uint64_t base, uint32_t off, uint32_t (&v)[4]) {
  __builtin_memcpy(v, reinterpret_cast<const uint8_t*>(base) + off, 16);
}
__device__ __forceinline__ void gload_u32x4_at(uint64_t addr, uint32_t (&v)[4]) {
  __builtin_memcpy(v, reinterpret_cast<const uint8_t*>(addr), 16);
}
__device__ __forceinline__ void keep64(uint64_t&) {}
template<int N> __device__ __forceinline__ void keep_all(uint32_t (&)[N]) {}
template<int N> __device__ __forceinline__ void keep_all_f(float (&)[N]) {}
__device__ __forceinline__ void keep(uint32_t&) {}
__device__ __forceinline__ void keep_f(float&) {}
__device__ __forceinline__ void keep_acc(uint32_t&) {}
__device__ __forceinline__ void keep_acc(unsigned long long&) {}
// v_rcp_f32 on the GPU (<= 1 ulp); exact division here
__device__ __forceinline__ float fast_rcp(float v) { return 1.0f / v; }
__device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float fast_sqrt(float v) { return __builtin_sqrtf(v); }

}  // namespace wave
