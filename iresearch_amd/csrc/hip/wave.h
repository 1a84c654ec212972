// wave.h — wavefront (64-lane) primitives for gfx950 (CDNA4).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#define IRS_WAVE 64
// register budget of a kernel: at least N wavefronts per SIMD (512 / N VGPRs each)
#define IRS_WAVES_PER_SIMD(N) __attribute__((amdgpu_waves_per_eu(N)))

namespace wave {

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

// DPP move with zero fill for lanes whose source is out of range / masked off.
template<int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_zero(uint32_t v) {
  return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, ROW_MASK, 0xF, true));
}

// Inclusive prefix sum across the 64 lanes of a wavefront, entirely in the
// VALU cross-lane network (no LDS traffic): row_shr 1/2/4/8 inside each row of
// 16 lanes, then row_bcast:15 (rows 1,3) and row_bcast:31 (rows 2,3).
__device__ __forceinline__ uint32_t inclusive_scan(uint32_t v) {
  v += dpp_zero<0x111, 0xF>(v);  // row_shr:1
  v += dpp_zero<0x112, 0xF>(v);  // row_shr:2
  v += dpp_zero<0x114, 0xF>(v);  // row_shr:4
  v += dpp_zero<0x118, 0xF>(v);  // row_shr:8
  v += dpp_zero<0x142, 0xA>(v);  // row_bcast:15 -> rows 1 and 3
  v += dpp_zero<0x143, 0xC>(v);  // row_bcast:31 -> rows 2 and 3
  return v;
}

// Two independent scans interleaved: each DPP step of one chain fills the
// wait states of the other.
__device__ __forceinline__ void inclusive_scan2(uint32_t& a, uint32_t& b) {
  a += dpp_zero<0x111, 0xF>(a);
  b += dpp_zero<0x111, 0xF>(b);
  a += dpp_zero<0x112, 0xF>(a);
  b += dpp_zero<0x112, 0xF>(b);
  a += dpp_zero<0x114, 0xF>(a);
  b += dpp_zero<0x114, 0xF>(b);
  a += dpp_zero<0x118, 0xF>(a);
  b += dpp_zero<0x118, 0xF>(b);
  a += dpp_zero<0x142, 0xA>(a);
  b += dpp_zero<0x142, 0xA>(b);
  a += dpp_zero<0x143, 0xC>(a);
  b += dpp_zero<0x143, 0xC>(b);
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
// LDS written by some lanes of this wavefront becomes readable by its other lanes
// (DS operations of one wavefront execute in order; this stops the compiler reordering).
__device__ __forceinline__ void sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A value known to be identical in every lane -> SGPR, so that branches on it
// are scalar branches instead of exec-mask juggling.
__device__ __forceinline__ uint32_t uniform(uint32_t v) {
  return uint32_t(__builtin_amdgcn_readfirstlane(int(v)));
}
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
  return (uint64_t(uniform(uint32_t(v >> 32))) << 32) | uniform(uint32_t(v));
}
__device__ __forceinline__ float uniform_f(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// Lane `k` (wave-uniform index) of a per-lane value -> SGPR (v_readlane_b32).
__device__ __forceinline__ uint32_t read_lane(uint32_t v, uint32_t k) {
  return uint32_t(__builtin_amdgcn_readlane(int(v), int(k)));
}
// `v` with lane `k` (wave-uniform index) replaced by the wave-uniform `x` (v_writelane_b32).
// (this clang has no __builtin_amdgcn_writelane: the intrinsic is bound by its name; the
// compiler moves the lane select into m0 itself)
extern "C" __device__ int irs_llvm_writelane(int, int, int) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ uint32_t write_lane(uint32_t v, uint32_t x, uint32_t k) {
  return uint32_t(irs_llvm_writelane(int(x), int(k), int(v)));
}
__device__ __forceinline__ float read_lane_f(float v, uint32_t k) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), int(k)));
}

// Cheap integer helpers that map to single full-rate instructions.
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
// ({hi, lo} >> s) & 0xffffffff, s in 0..31 (v_alignbit_b32)
__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t s) {
  return __builtin_amdgcn_alignbit(hi, lo, s);
}
// low `bits` bits of x, 1 <= bits <= 31 (v_bfe_u32)
__device__ __forceinline__ uint32_t bfe(uint32_t x, uint32_t bits) {
  return __builtin_amdgcn_ubfe(x, 0u, bits);
}

// high 32 bits of a 32 x 32 bit product (v_mul_hi_u32)
__device__ __forceinline__ uint32_t mul_hi(uint32_t a, uint32_t b) { return __umulhi(a, b); }

// Optimisation barrier: the value must exist in a VGPR at this program point.
__device__ __forceinline__ void keep(uint32_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void keep_f(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void keep_acc(uint32_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void keep_acc(unsigned long long& v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ void keep_all(uint32_t (&v)[2]) { asm volatile("" : "+v"(v[0]), "+v"(v[1])); }
__device__ __forceinline__ void keep_all(uint32_t (&v)[4]) {
  asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}
__device__ __forceinline__ void keep_all_f(float (&v)[2]) { asm volatile("" : "+v"(v[0]), "+v"(v[1])); }
__device__ __forceinline__ void keep_all_f(float (&v)[4]) {
  asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}

// acc += number of non-zero values among a..d: v_min_u32(1, x) pairs folded by v_add3_u32 in
// one block (two temporaries) — written out because the optimiser turns min(x, 1) back into
// compare + add-with-carry, which costs a VCC round trip with wait states per value on gfx950.
__device__ __forceinline__ void count_nonzero4(uint32_t& acc, uint32_t a, uint32_t b, uint32_t c,
                                               uint32_t d) {
  uint32_t t0, t1;
  asm("v_min_u32 %1, 1, %3\n\tv_min_u32 %2, 1, %4\n\tv_add3_u32 %0, %0, %1, %2\n\t"
      "v_min_u32 %1, 1, %5\n\tv_min_u32 %2, 1, %6\n\tv_add3_u32 %0, %0, %1, %2"
      : "+v"(acc), "=&v"(t0), "=&v"(t1)
      : "v"(a), "v"(b), "v"(c), "v"(d));
}
__device__ __forceinline__ void count_nonzero4(uint32_t& acc, unsigned long long a,
                                               unsigned long long b, unsigned long long c,
                                               unsigned long long d) {
  acc += (a != 0) + (b != 0) + (c != 0) + (d != 0);
}

// Two 16-bit halves per register (k_join_score's paired tiles): v_pk_min_u16 / v_pk_max_u16 /
// v_pk_add_u16 — one instruction for both halves (vector types, not inline assembly: the compiler
// then knows the instructions' hazards and may keep uniform operands in SGPRs).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a),
                                                                __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a),
                                                                __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b));
}
// acc (two 16-bit counters) += the non-zero low halves / high halves among a..d: v_pk_min_u16
// against 1|1 folded by v_pk_add_u16, in one block — written out because the optimiser turns
// min(x, 1) into compare + select per half (a VCC round trip with wait states each).  (A packed
// result needs one wait state before a dependent read — the compiler puts s_nop 0 between its own
// dependent v_pk_* — hence the order of the block and its one s_nop.)
__device__ __forceinline__ void count_nonzero_halves4(uint32_t& acc, uint32_t a, uint32_t b,
                                                      uint32_t c, uint32_t d) {
  uint32_t t0, t1;
  const uint32_t one = 0x00010001u;
  asm("v_pk_min_u16 %1, %3, %7\n\tv_pk_min_u16 %2, %4, %7\n\tv_pk_add_u16 %0, %0, %1\n\t"
      "v_pk_min_u16 %1, %5, %7\n\tv_pk_add_u16 %0, %0, %2\n\tv_pk_min_u16 %2, %6, %7\n\t"
      "v_pk_add_u16 %0, %0, %1\n\ts_nop 0\n\tv_pk_add_u16 %0, %0, %2"
      : "+v"(acc), "=&v"(t0), "=&v"(t1)
      : "v"(a), "v"(b), "v"(c), "v"(d), "s"(one));
}

// A wave-uniform value the optimiser may not reason about (stays in an SGPR).
__device__ __forceinline__ uint32_t opaque(uint32_t v) {
  asm volatile("" : "+s"(v));
  return v;
}

__device__ __forceinline__ uint64_t opaque64(uint64_t v) {
  asm volatile("" : "+s"(v));
  return v;
}

// A register whose content does not matter (no instruction is emitted).
__device__ __forceinline__ uint64_t undef64() {
  uint64_t v;
  asm volatile("" : "=v"(v));
  return v;
}

// Four registers whose content does not matter (no instruction is emitted).
__device__ __forceinline__ void undef4(uint32_t (&v)[4]) {
  asm volatile("" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
}

// v_rcp_f32: <= 1 ulp
__device__ __forceinline__ float fast_rcp(float v) { return __builtin_amdgcn_rcpf(v); }

// v_sqrt_f32: <= 1 ulp
__device__ __forceinline__ float fast_sqrt(float v) { return __builtin_amdgcn_sqrtf(v); }

// a*b + c in one rounding (v_fma_f32), independent of -ffp-contract
__device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ---- LDS by absolute address ------------------------------------------------
// The dynamic LDS block of a kernel WITHOUT static __shared__ variables starts at LDS
// address 0, but the compiler only learns the address of `extern __shared__` after
// instruction selection and then leaves a `v_add 0` in front of every access.  On the
// hot path the tile arrays are therefore addressed by their compile-time byte offsets
// (TileSmemT layout) through address-space-3 pointers built from integers: the carve
// offset lands in the instruction's immediate field and no address arithmetic is left.
// `base` is only used by the CPU emulator twin of this header; lds_is_at_zero() lets a
// kernel verify the assumption once.
#define IRS_LDS __attribute__((address_space(3)))
__device__ __forceinline__ bool lds_is_at_zero(const unsigned char* smem) {
  return uint32_t(uintptr_t((IRS_LDS const unsigned char*)smem)) == 0u;
}
__device__ __forceinline__ uint32_t lds_u8(const unsigned char*, uint32_t off) {
  return *(const IRS_LDS uint8_t*)(uintptr_t)off;
}
__device__ __forceinline__ float lds_f32(const unsigned char*, uint32_t off) {
  return *(const IRS_LDS float*)(uintptr_t)off;
}
__device__ __forceinline__ void lds_add(const unsigned char*, uint32_t off, uint32_t v) {
  __hip_atomic_fetch_add((IRS_LDS uint32_t*)(uintptr_t)off, v, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_add(const unsigned char*, uint32_t off, unsigned long long v) {
  __hip_atomic_fetch_add((IRS_LDS unsigned long long*)(uintptr_t)off, v, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
}

// 16 bytes of LDS at a 16-byte aligned offset: ds_read_b128 / ds_write_b128
__device__ __forceinline__ void lds_read4(const unsigned char*, uint32_t off, uint32_t (&v)[4]) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 x = *(const IRS_LDS u32x4*)(uintptr_t)off;
  v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
}
// Read 16 bytes of LDS and leave zeros behind, in ONE pass of the LDS pipeline
// (ds_wrxchg2_rtn_b64: an atomic exchange of two adjacent 8-byte words) instead of a 16-byte
// read followed by a 16-byte write.  take4x2: two of them in flight behind one wait.
__device__ __forceinline__ void lds_take4(unsigned char*, uint32_t off, uint32_t (&v)[4]) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 x;
  uint64_t z = 0;
  asm volatile("ds_wrxchg2_rtn_b64 %0, %1, %2, %2 offset1:1\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(x) : "v"(off), "v"(z) : "memory");
  v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
}
__device__ __forceinline__ void lds_take4x2(unsigned char*, uint32_t off0, uint32_t off1,
                                            uint32_t (&a)[4], uint32_t (&b)[4]) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 x, y;
  uint64_t z = 0;
  asm volatile("ds_wrxchg2_rtn_b64 %0, %2, %4, %4 offset1:1\n\t"
               "ds_wrxchg2_rtn_b64 %1, %3, %4, %4 offset1:1\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(x), "=&v"(y) : "v"(off0), "v"(off1), "v"(z) : "memory");
  a[0] = x[0]; a[1] = x[1]; a[2] = x[2]; a[3] = x[3];
  b[0] = y[0]; b[1] = y[1]; b[2] = y[2]; b[3] = y[3];
}
__device__ __forceinline__ void lds_take4x3(unsigned char*, uint32_t off0, uint32_t off1, uint32_t off2,
                                            uint32_t (&a)[4], uint32_t (&b)[4], uint32_t (&c)[4]) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 x, y, w;
  uint64_t z = 0;
  asm volatile("ds_wrxchg2_rtn_b64 %0, %3, %6, %6 offset1:1\n\t"
               "ds_wrxchg2_rtn_b64 %1, %4, %6, %6 offset1:1\n\t"
               "ds_wrxchg2_rtn_b64 %2, %5, %6, %6 offset1:1\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(x), "=&v"(y), "=&v"(w) : "v"(off0), "v"(off1), "v"(off2), "v"(z) : "memory");
  a[0] = x[0]; a[1] = x[1]; a[2] = x[2]; a[3] = x[3];
  b[0] = y[0]; b[1] = y[1]; b[2] = y[2]; b[3] = y[3];
  c[0] = w[0]; c[1] = w[1]; c[2] = w[2]; c[3] = w[3];
}
// (ds_write2_b64 from ONE zeroed register pair: a ds_write_b128 would pin four zero registers
// for the whole kernel)
__device__ __forceinline__ void lds_zero4(unsigned char*, uint32_t off) {
  uint64_t z = 0;
  asm volatile("" : "+v"(z));
  *(IRS_LDS uint64_t*)(uintptr_t)off = z;
  *(IRS_LDS uint64_t*)(uintptr_t)(off + 8u) = z;
}

// LDS float accumulate without a returned value -> ds_add_f32
__device__ __forceinline__ void lds_add(float* p, float v) { atomicAdd(p, v); }

// ---- loads that are NOT flat ---------------------------------------------------
// A pointer read out of a struct in memory is a generic pointer: the compiler emits
// FLAT loads for it, which count on BOTH vmcnt and lgkmcnt — every wait for an LDS
// result then also drains the global loads in flight.  The hot paths therefore go
// through explicit address spaces:
//   sload<T>(a)      a record at a wave-uniform ADDRESS (an integer: a pointer would let the
//                    compiler trace it back to a global-memory kernel argument and fall back
//                    to vector loads) that nobody writes during the kernel -> constant
//                    address space -> s_load_dwordxN into SGPRs
//   gload_u64(b, o)  8 bytes at (wave-uniform 64-bit base) + (per-lane 32-bit offset)
//                    -> global_load_dwordx2 v, v_off, s[base] (saddr form, vmcnt only)
#define IRS_CONST __attribute__((address_space(4)))
#define IRS_GLOBAL __attribute__((address_space(1)))
template<typename T>
__device__ __forceinline__ T sload(uint64_t addr) {
  T v;
  __builtin_memcpy(&v, (const IRS_CONST T*)addr, sizeof(T));
  return v;
}
__device__ __forceinline__ uint64_t gload_u64(uint64_t base, uint32_t off) {
  typedef uint64_t __attribute__((aligned(4))) u64a4;
  return *(const IRS_GLOBAL u64a4*)((const IRS_GLOBAL uint8_t*)base + off);
}

// 4 bytes at (wave-uniform 64-bit base) + (per-lane 32-bit offset): global_load_dword, saddr form
__device__ __forceinline__ uint32_t gload_u32(uint64_t base, uint32_t off) {
  return *(const IRS_GLOBAL uint32_t*)((const IRS_GLOBAL uint8_t*)base + off);
}

// 16 bytes at (wave-uniform 64-bit base) + (per-lane 32-bit offset), 4-byte aligned:
// global_load_dwordx4, saddr form
__device__ __forceinline__ void gload_u32x4(uint64_t base, uint32_t off, uint32_t (&v)[4]) {
  typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
  const u32x4a4 x = *(const IRS_GLOBAL u32x4a4*)((const IRS_GLOBAL uint8_t*)base + off);
  v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
}

// ... at a per-lane address
__device__ __forceinline__ void gload_u32x4_at(uint64_t addr, uint32_t (&v)[4]) {
  typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
  const u32x4a4 x = *(const IRS_GLOBAL u32x4a4*)addr;
  v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
}
__device__ __forceinline__ void keep64(uint64_t& v) { asm volatile("" : "+v"(v)); }

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

}  // namespace wave
