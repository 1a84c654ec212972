// Micro-benchmark (dev tool): what does the per-tile skeleton of a doc-tile kernel cost on gfx950?
// 1024-thread workgroups, 2 per CU (56 KB of LDS each), every iteration = one "tile":
//   mode 0  two barriers only
//   mode 1  + read-and-clear of 48 KB by ds_wrxchg2_rtn_b64 (3 per lane, one at a time)
//   mode 2  + read-and-clear by ds_read_b128 + ds_write_b128 (3 + 3 per lane)
//   mode 3  mode 1 with two exchanges in flight
//   mode 4  mode 1 + 14 ds_add_u32 per lane-slab (random addresses) in front of the first barrier
//   mode 5  mode 4 with conflict-free addresses
// Prints ns per tile per workgroup and the shader clock it implies for s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define LDS __attribute__((address_space(3)))
template<int MODE>
__global__ void __launch_bounds__(1024) k(uint32_t* out, int iters, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* acc = reinterpret_cast<uint32_t*>(smem);
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < 12288; i += 1024) acc[i] = 0;
  __syncthreads();
  uint32_t sum = 0;
  uint32_t rnd = tid * 2654435761u + blockIdx.x;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 4) {
#pragma unroll
      for (int s = 0; s < 14; ++s) {
        rnd = rnd * 1664525u + 1013904223u;
        const uint32_t a = MODE == 5 ? (((rnd >> 8) & 0xBF00u) | ((tid & 63u) * 4u)) : ((rnd >> 8) % 12288u) * 4u;
        __hip_atomic_fetch_add((LDS uint32_t*)(uintptr_t)a, rnd & 0xFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    __syncthreads();
    if (MODE == 1 || MODE >= 4) {
      for (uint32_t i = tid * 4u; i < 12288u; i += 4096u) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 x; uint64_t z = 0;
        asm volatile("ds_wrxchg2_rtn_b64 %0, %1, %2, %2 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x) : "v"(i * 4u), "v"(z) : "memory");
        sum += x[0] + x[1] + x[2] + x[3];
      }
    } else if (MODE == 2) {
      for (uint32_t i = tid * 4u; i < 12288u; i += 4096u) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 x = *(LDS u32x4*)(uintptr_t)(i * 4u);
        u32x4 z = {0, 0, 0, 0};
        *(LDS u32x4*)(uintptr_t)(i * 4u) = z;
        sum += x[0] + x[1] + x[2] + x[3];
      }
    } else if (MODE == 3) {
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      u32x4 x, y, w; uint64_t z = 0;
      const uint32_t i = tid * 16u;
      asm volatile("ds_wrxchg2_rtn_b64 %0, %3, %6, %6 offset1:1\n\tds_wrxchg2_rtn_b64 %1, %4, %6, %6 offset1:1\n\t"
                   "ds_wrxchg2_rtn_b64 %2, %5, %6, %6 offset1:1\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(x), "=&v"(y), "=&v"(w) : "v"(i), "v"(i + 16384u), "v"(i + 32768u), "v"(z) : "memory");
      sum += x[0] + y[1] + w[2];
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 1024 + tid] = sum;
  if (tid == 0 && blockIdx.x == 0) clk[MODE] = t1 - t0;
}
int main() {
  const int iters = 4000;
  uint32_t* out; unsigned long long* clk;
  hipMalloc(&out, 512 * 1024 * 4); hipMalloc(&clk, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t smem = 56 * 1024;
#define RUN(M) { hipFuncSetAttribute((const void*)k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(k<M>, dim3(512), dim3(1024), smem, 0, out, iters, clk); \
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
      unsigned long long c; hipMemcpy(&c, clk + M, 8, hipMemcpyDeviceToHost); \
      if (rep) printf("mode %d: %.3f ms total, %.0f ns per tile per workgroup, s_memtime %.0f ticks per tile (%.2f GHz)\n", M, ms, ms * 1e6 / iters, double(c) / iters, double(c) / (ms * 1e6)); } }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
  return 0;
}
