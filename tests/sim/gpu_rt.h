// tests/sim/gpu_rt.h — TEST INFRASTRUCTURE ONLY: the rt:: surface of
// iresearch_amd/csrc/hip/gpu_rt.h on top of the CPU fiber emulator.
#pragma once
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "hip_sim.h"

namespace rt {

using stream_t = void*;
using event_t = double;

inline int device_count() { return 1; }
inline bool set_device(int dev) { return dev == 0; }
inline bool device_arch(int, char* buf, size_t cap) {
  std::strncpy(buf, "gfx950-sim", cap);
  if (cap) buf[cap - 1] = 0;
  return true;
}
inline int device_cus(int) { return 3; }  // a few persistent workgroups
inline void* dmalloc(size_t n) {
  void* p = std::malloc(n ? n : 1);
  if (p) std::memset(p, 0xCD, n ? n : 1);  // hipMalloc does not zero: poison
  return p;
}
inline void dfree(void* p) { std::free(p); }
inline void* hmalloc(size_t n) { return std::malloc(n ? n : 1); }
inline void hfree(void* p) { std::free(p); }
inline bool h2d(void* d, const void* h, size_t n, stream_t) {
  if (n) std::memcpy(d, h, n);
  return true;
}
inline bool d2h(void* h, const void* d, size_t n, stream_t) {
  if (n) std::memcpy(h, d, n);
  return true;
}
inline bool d2d(void* dst, const void* src, size_t n, stream_t) {
  if (n) std::memcpy(dst, src, n);
  return true;
}
inline bool dmemset(void* d, int v, size_t n, stream_t) {
  if (n) std::memset(d, v, n);
  return true;
}
inline bool sync(stream_t) { return true; }
inline bool last_error_ok() { return true; }
inline bool allow_dynamic_smem(const void*, size_t bytes) { return bytes <= sim::kMaxSmem; }

inline bool event_create(event_t* e) {
  *e = 0;
  return true;
}
inline void event_destroy(event_t) {}
inline double now_ms() {
  return std::chrono::duration<double, std::milli>(
           std::chrono::steady_clock::now().time_since_epoch())
    .count();
}
inline bool event_record(event_t& e, stream_t) {
  e = now_ms();
  return true;
}
inline bool event_elapsed(float* ms, event_t a, event_t b) {
  *ms = float(b - a);
  return true;
}

}  // namespace rt

namespace sim {
template<typename K, typename... A>
inline void launch_kernel(unsigned grid, unsigned block, size_t shmem, K kernel, A... args) {
  launch(grid, block, shmem, [=]() { kernel(args...); });
}
}  // namespace sim

#define RT_LAUNCH(kernel, grid, block, shmem, stream, ...)                       \
  do {                                                                           \
    (void)(stream);                                                              \
    sim::launch_kernel(unsigned(grid), unsigned(block), size_t(shmem), kernel,   \
                       __VA_ARGS__);                                             \
  } while (0)

#define RT_DYN_SMEM(name) unsigned char* name = sim::dyn_smem()
