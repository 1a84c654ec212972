// tests/sim/gpu_rt.h — TEST INFRASTRUCTURE ONLY: the rt:: surface of
// iresearch_amd/csrc/hip/gpu_rt.h on top of the CPU fiber emulator.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
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
inline int current_device() { return 0; }
inline size_t pool_cap_bytes() { return size_t(256) << 20; }
inline void poison(void* p, size_t n) {   // recycled memory is not zero either
#ifndef IRS_SIM_NO_POISON
  std::memset(p, 0xCD, n);
#else
  (void)p; (void)n;
#endif
}
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
inline bool stream_create(stream_t* s) { *s = nullptr; return true; }
inline bool last_error_ok() { return true; }
inline bool allow_dynamic_smem(const void*, size_t bytes) { return bytes <= sim::kMaxSmem; }

inline bool event_sync(event_t) { return true; }
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
inline bool stream_wait(stream_t, const event_t&) { return true; }   // (everything is synchronous)
inline bool event_elapsed(float* ms, event_t a, event_t b) {
  *ms = float(b - a);
  return true;
}

// ---- communicator of the CPU test tier: ranks are PROCESSES sharing one file under /dev/shm
// (header with a generation barrier, then one slot per rank).  Semantics of the RCCL calls
// of hip/gpu_rt.h, nothing of their performance.
namespace comm {

constexpr size_t kIdBytes = 128;
using handle_t = void*;
constexpr size_t kSlotBytes = 8u << 20;   // largest per-rank message of the tests

struct Shared {
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
  uint32_t nranks;
  uint32_t pad;
};
struct Comm {
  Shared* sh = nullptr;
  unsigned char* slots = nullptr;
  size_t map_bytes = 0;
  int nranks = 0, rank = 0;
  char path[160];
};

inline bool shm_map(const char* path, size_t bytes, bool create, void** out) {
  int fd = -1;
  if (create) {
    fd = ::open(path, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ::ftruncate(fd, off_t(bytes)) != 0) return false;   // (new file: zero-filled)
  } else {
    for (int tries = 0; tries < 20000 && fd < 0; ++tries) {   // up to ~20 s
      fd = ::open(path, O_RDWR);
      struct stat st;
      if (fd >= 0 && (::fstat(fd, &st) != 0 || size_t(st.st_size) < bytes)) {
        ::close(fd);
        fd = -1;
      }
      if (fd < 0) ::usleep(1000);
    }
    if (fd < 0) return false;
  }
  void* m = ::mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  ::close(fd);
  if (m == MAP_FAILED) return false;
  *out = m;
  return true;
}

inline bool unique_id(void* id128) {
  std::memset(id128, 0, kIdBytes);
  static std::atomic<unsigned> counter{0};
  std::snprintf(static_cast<char*>(id128), kIdBytes, "/dev/shm/irs_sim_comm_%ld_%u_%ld",
                long(::getpid()), counter.fetch_add(1),
                long(std::chrono::steady_clock::now().time_since_epoch().count() & 0xFFFFFF));
  return true;
}
inline void barrier(Comm* c) {
  const uint32_t gen = c->sh->generation.load();
  if (c->sh->arrived.fetch_add(1) + 1 == uint32_t(c->nranks)) {
    c->sh->arrived.store(0);
    c->sh->generation.fetch_add(1);
  } else {
    while (c->sh->generation.load() == gen) ::usleep(50);
  }
}
inline bool init_rank(handle_t* out, int nranks, const void* id128, int rank) {
  if (nranks < 1 || rank < 0 || rank >= nranks) return false;
  Comm* c = new Comm;
  c->nranks = nranks;
  c->rank = rank;
  std::memcpy(c->path, id128, kIdBytes);
  c->path[kIdBytes - 1] = 0;
  c->map_bytes = 4096 + size_t(nranks) * kSlotBytes;
  void* m = nullptr;
  // rank 0 creates (and zero-fills) the file, the others wait for it to appear
  if (!shm_map(c->path, c->map_bytes, rank == 0, &m)) {
    delete c;
    return false;
  }
  c->sh = static_cast<Shared*>(m);
  c->slots = static_cast<unsigned char*>(m) + 4096;
  if (rank == 0) c->sh->nranks = uint32_t(nranks);
  barrier(c);
  *out = c;
  return true;
}
inline bool all_gather(handle_t h, const void* send, void* recv, size_t bytes, stream_t) {
  Comm* c = static_cast<Comm*>(h);
  if (bytes > kSlotBytes) return false;
  std::memcpy(c->slots + size_t(c->rank) * kSlotBytes, send, bytes);
  barrier(c);
  for (int r = 0; r < c->nranks; ++r)
    std::memcpy(static_cast<unsigned char*>(recv) + size_t(r) * bytes, c->slots + size_t(r) * kSlotBytes, bytes);
  barrier(c);
  return true;
}
inline bool all_reduce_u32(handle_t h, void* buf, size_t count, stream_t) {
  Comm* c = static_cast<Comm*>(h);
  if (count * 4 > kSlotBytes) return false;
  std::memcpy(c->slots + size_t(c->rank) * kSlotBytes, buf, count * 4);
  barrier(c);
  uint32_t* out = static_cast<uint32_t*>(buf);
  for (size_t i = 0; i < count; ++i) out[i] = 0;
  for (int r = 0; r < c->nranks; ++r) {
    const uint32_t* in = reinterpret_cast<const uint32_t*>(c->slots + size_t(r) * kSlotBytes);
    for (size_t i = 0; i < count; ++i) out[i] += in[i];
  }
  barrier(c);
  return true;
}
inline bool library(char* buf, size_t cap) {
  std::snprintf(buf, cap, "shared-memory stand-in (tests/sim)");
  return true;
}
inline void destroy(handle_t h) {
  Comm* c = static_cast<Comm*>(h);
  if (!c) return;
  barrier(c);   // nobody still reads the slots
  ::munmap(c->sh, c->map_bytes);
  if (c->rank == 0) ::unlink(c->path);
  delete c;
}

}  // namespace comm

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

#define RT_WAVES_PER_SIMD(n)

#define RT_DYN_SMEM(name) unsigned char* name = sim::dyn_smem()
