// gpu_rt.h — the HIP runtime surface the host side of libirs_hip uses.
// (tests/sim/ carries a same-named header that runs the same host code and
// kernels on a CPU fiber emulator for the CPU-only test tier; the product is
// only ever built against THIS file.)
#pragma once
#include <dlfcn.h>
#include <link.h>
#include <cstdio>
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
inline int current_device() {
  int d = 0;
  return ok(hipGetDevice(&d)) ? d : 0;
}
// bytes of freed device / page-locked memory the library keeps for reuse per device (irs_hip.hip
// `pool`): hipMalloc / hipFree / hipHostMalloc cost 0.1-1 ms each and hipFree synchronises the
// device, so a batch's buffers are recycled instead
inline size_t pool_cap_bytes() { return size_t(64) << 30; }
inline void poison(void*, size_t) {}   // (the CPU test tier marks recycled memory)
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
// a stream that does not synchronise with the null stream (copies that overlap other streams' kernels)
inline bool stream_create(stream_t* s) { return ok(hipStreamCreateWithFlags(s, hipStreamNonBlocking)); }
inline bool last_error_ok() { return ok(hipGetLastError()); }

// Dynamic LDS beyond the default limit needs an explicit opt-in per kernel.
inline bool allow_dynamic_smem(const void* fn, size_t bytes) {
  if (bytes <= 32 * 1024) return true;
  return ok(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
}

inline bool event_sync(event_t e) { return ok(hipEventSynchronize(e)); }
inline bool event_create(event_t* e) { return ok(hipEventCreate(e)); }
inline void event_destroy(event_t e) { (void)hipEventDestroy(e); }
inline bool event_record(event_t e, stream_t s) { return ok(hipEventRecord(e, s)); }
// work queued on `s` from here on starts after `e`
inline bool stream_wait(stream_t s, event_t e) { return ok(hipStreamWaitEvent(s, e, 0)); }
inline bool event_elapsed(float* ms, event_t a, event_t b) {
  return ok(hipEventElapsedTime(ms, a, b));
}

// ---- RCCL (the collective of the multi-GPU top-k exchange) -----------------------------
// Bound at first use with dlopen: a process that never creates a communicator never loads
// librccl.  Only the four entry points the exchange needs.
namespace comm {

constexpr size_t kIdBytes = 128;   // NCCL_UNIQUE_ID_BYTES
struct UniqueId {
  char internal[kIdBytes];
};
using handle_t = void*;            // ncclComm_t

struct Api {
  char path[512] = {0};   // the library the entry points were bound from ("" = none)
  bool preloaded = false; // ... which the process had mapped already (e.g. torch's bundled one)
  int (*get_unique_id)(UniqueId*) = nullptr;
  int (*init_rank)(handle_t*, int, UniqueId, int) = nullptr;
  int (*all_gather)(const void*, void*, size_t, int /*ncclDataType_t*/, handle_t, hipStream_t) = nullptr;
  int (*all_reduce)(const void*, void*, size_t, int /*ncclDataType_t*/, int /*ncclRedOp_t*/, handle_t,
                    hipStream_t) = nullptr;
  int (*destroy)(handle_t) = nullptr;
  bool ok = false;
};
// An RCCL this process has mapped already — a host that also runs torch.distributed has torch's
// bundled torch/lib/librccl.so in its address space; a second copy next to it would mean two
// sets of RCCL globals (bootstrap threads, topology caches, IPC handle tables) in one process.
inline int find_mapped_rccl(struct dl_phdr_info* info, size_t, void* out) {
  const char* name = info->dlpi_name;
  if (name && std::strstr(name, "librccl.so")) {
    std::snprintf(static_cast<char*>(out), 512, "%s", name);
    return 1;
  }
  return 0;
}
inline const Api& api() {   // bound once per process, at first use
  static const Api a = [] {
    Api x;
    void* h = nullptr;
    char mapped[512] = {0};
    if (dl_iterate_phdr(find_mapped_rccl, mapped) && mapped[0]) {
      h = dlopen(mapped, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);   // (only takes a reference)
      if (h) {
        x.preloaded = true;
        std::snprintf(x.path, sizeof x.path, "%s", mapped);
      }
    }
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      if (h) break;
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h) std::snprintf(x.path, sizeof x.path, "%s", name);
    }
    if (!h) return x;
    x.get_unique_id = reinterpret_cast<decltype(x.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    x.init_rank = reinterpret_cast<decltype(x.init_rank)>(dlsym(h, "ncclCommInitRank"));
    x.all_gather = reinterpret_cast<decltype(x.all_gather)>(dlsym(h, "ncclAllGather"));
    x.all_reduce = reinterpret_cast<decltype(x.all_reduce)>(dlsym(h, "ncclAllReduce"));
    x.destroy = reinterpret_cast<decltype(x.destroy)>(dlsym(h, "ncclCommDestroy"));
    x.ok = x.get_unique_id && x.init_rank && x.all_gather && x.all_reduce && x.destroy;
    return x;
  }();
  return a;
}

inline bool unique_id(void* id128) {
  const Api& a = api();
  return a.ok && a.get_unique_id(static_cast<UniqueId*>(id128)) == 0;
}
inline bool init_rank(handle_t* out, int nranks, const void* id128, int rank) {
  const Api& a = api();
  if (!a.ok) return false;
  UniqueId id;
  std::memcpy(&id, id128, kIdBytes);
  return a.init_rank(out, nranks, id, rank) == 0;
}
inline bool all_gather(handle_t c, const void* send, void* recv, size_t bytes, stream_t s) {
  const Api& a = api();
  return a.ok && a.all_gather(send, recv, bytes, 0 /*ncclInt8*/, c, s) == 0;
}
// in place: every rank's `count` 32-bit counters summed, the sums on every rank
inline bool all_reduce_u32(handle_t c, void* buf, size_t count, stream_t s) {
  const Api& a = api();
  return a.ok && a.all_reduce(buf, buf, count, 3 /*ncclUint32*/, 0 /*ncclSum*/, c, s) == 0;
}
inline void destroy(handle_t c) {
  const Api& a = api();
  if (a.ok && c) (void)a.destroy(c);
}
// which library the collective runs on: its path, prefixed "mapped:" when the process had it
// loaded before this library asked (diagnostics: bench.py prints it)
inline bool library(char* buf, size_t cap) {
  const Api& a = api();
  if (!a.ok) return false;
  std::snprintf(buf, cap, "%s%s", a.preloaded ? "mapped:" : "", a.path);
  return true;
}

}  // namespace comm

}  // namespace rt

// One spelling for a kernel launch on both builds.
#define RT_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (shmem), (stream), __VA_ARGS__)

// A kernel compiled for exactly n resident wavefronts per SIMD (its register budget follows).
#define RT_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))

// Dynamic LDS carve base, 16-byte aligned (cdna guide G17).
#define RT_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]

