// Micro-benchmark (dev tool): what does rocprofv3's FETCH_SIZE report for a coalesced streaming
// read of a KNOWN byte count on gfx950 — with 4-byte loads per lane (global_load_dword: what
// k_join_score / k_join_fast issue) and with 16-byte loads per lane (the case the MI355X guide
// calibrated: reports exactly half)?  A 4 GB buffer (16x the Infinity Cache) is read once per kernel.
//   rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib     (tools/gpu/fetch_calib.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(256) read4(const uint32_t* p, uint64_t n, uint32_t* out) {
  uint32_t s = 0;
  for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) s += p[i];
  if (s == 0x12345678u) out[0] = s;
}
__global__ void __launch_bounds__(256) read16(const uint4* p, uint64_t n, uint32_t* out) {
  uint32_t s = 0;
  for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) {
    const uint4 v = p[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 0x12345678u) out[0] = s;
}
int main() {
  const uint64_t bytes = 4ull << 30;
  void* buf; uint32_t* out;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
  (void)hipMemset(buf, 1, bytes);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    float ms;
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(read4, dim3(256 * 8), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 4, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    printf("read4  : %.3f ms, %.1f GB/s, %llu bytes read\n", ms, bytes / ms / 1e6, (unsigned long long)bytes);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(read16, dim3(256 * 8), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    printf("read16 : %.3f ms, %.1f GB/s, %llu bytes read\n", ms, bytes / ms / 1e6, (unsigned long long)bytes);
  }
  return 0;
}
