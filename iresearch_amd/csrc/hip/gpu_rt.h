// gpu_rt.h — the HIP runtime surface the host side of libirs_hip uses.
// (tests/sim/ carries a same-named header that runs the same host code and
// kernels on a CPU fiber emulator for the CPU-only test tier; the product is
// only ever built against THIS file.)
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstring>

namespace rt {

using stream_t = hipStream_t;
using event_t = hipEvent_t;

inline bool ok(hipError_t e) { return e == hipSuccess; }

inline int device_count() {
  int n = 0;
  return ok(hipGetDeviceCount(&n)) ? n : 0;
}
inline bool set_device(int dev) { return ok(hipSetDevice(dev)); }
inline bool device_arch(int dev, char* buf, size_t cap) {
  hipDeviceProp_t p;
  if (!ok(hipGetDeviceProperties(&p, dev))) return false;
  std::strncpy(buf, p.gcnArchName, cap);
  if (cap) buf[cap - 1] = 0;
  return true;
}
inline int device_cus(int dev) {
  hipDeviceProp_t p;
  return ok(hipGetDeviceProperties(&p, dev)) ? p.multiProcessorCount : 0;
}
inline void* dmalloc(size_t n) {
  void* p = nullptr;
  return ok(hipMalloc(&p, n ? n : 1)) ? p : nullptr;
}
inline void dfree(void* p) {
  if (p) (void)hipFree(p);
}
// page-locked host memory: device copies into it run at full PCIe speed
inline void* hmalloc(size_t n) {
  void* p = nullptr;
  return ok(hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault)) ? p : nullptr;
}
inline void hfree(void* p) {
  if (p) (void)hipHostFree(p);
}
inline bool h2d(void* d, const void* h, size_t n, stream_t s) {
  return n == 0 || ok(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s));
}
inline bool d2h(void* h, const void* d, size_t n, stream_t s) {
  return n == 0 || ok(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s));
}
inline bool d2d(void* dst, const void* src, size_t n, stream_t s) {
  return n == 0 || ok(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s));
}
inline bool dmemset(void* d, int v, size_t n, stream_t s) {
  return n == 0 || ok(hipMemsetAsync(d, v, n, s));
}
inline bool sync(stream_t s) { return ok(hipStreamSynchronize(s)); }
inline bool last_error_ok() { return ok(hipGetLastError()); }

// Dynamic LDS beyond the default limit needs an explicit opt-in per kernel.
inline bool allow_dynamic_smem(const void* fn, size_t bytes) {
  if (bytes <= 32 * 1024) return true;
  return ok(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
}

inline bool event_create(event_t* e) { return ok(hipEventCreate(e)); }
inline void event_destroy(event_t e) { (void)hipEventDestroy(e); }
inline bool event_record(event_t e, stream_t s) { return ok(hipEventRecord(e, s)); }
inline bool event_elapsed(float* ms, event_t a, event_t b) {
  return ok(hipEventElapsedTime(ms, a, b));
}

}  // namespace rt

// One spelling for a kernel launch on both builds.
#define RT_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (shmem), (stream), __VA_ARGS__)

// Dynamic LDS carve base, 16-byte aligned (cdna guide G17).
#define RT_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]

