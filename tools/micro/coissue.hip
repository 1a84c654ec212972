// Micro-benchmark (dev tool): does SALU work co-issue with VALU work of other
// wavefronts on a gfx950 SIMD?  Three loops of the same VALU work: plain, with one
// scalar add per vector op, with one s_waitcnt-free LDS read per 8 vector ops.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template<int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, int sx) {
  float a = threadIdx.x, b = 1.0001f, c = 0.5f, d = 2.f;
  int s0 = sx, s1 = sx + 1, s2 = sx + 2, s3 = sx + 3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
      if (MODE == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) : : "scc");
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d) : "v"(b), "v"(c));
      if (MODE == 1) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s1) : : "scc");
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(d));
      if (MODE == 1) asm volatile("s_xor_b32 %0, %0, %1" : "+s"(s2) : "s"(s0) : "scc");
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d) : "v"(b), "v"(a));
      if (MODE == 1) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s3) : "s"(s1) : "scc");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + d + float(s0 + s1 + s2 + s3);
}

int main(int argc, char** argv) {
  const int iters = 20000;
  float* out;
  hipMalloc(&out, 4 << 20);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps = 1; wps <= 8; wps *= 2) {        // wavefronts per SIMD
    const int blocks = 256 * wps;                // 256 CUs, 4 waves per block
    for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, rep);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) printf("error: %s\n", hipGetErrorString(err));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) {
          const double valu = double(blocks) * 4 * iters * 64;   // wave-instructions
          printf("waves/SIMD %d  mode %s  %.3f ms  VALU %.2f G wave-instr/s  (%.2f per SIMD-cycle @2.4GHz)\n",
                 wps, mode ? "valu+salu" : "valu     ", ms, valu / ms / 1e6,
                 valu / (ms * 1e-3) / (1024 * 2.4e9));
        }
      }
    }
  }
  return 0;
}
