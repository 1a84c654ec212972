// tests/sim/hip_sim.h — TEST INFRASTRUCTURE ONLY.  Never part of the product.
//
// A functional emulator of the HIP execution model, just large enough to run
// iresearch_amd/csrc/{kernels,decode}.h and irs_hip.hip UNMODIFIED on the CPU
// in the build container (which has no GPU): every GPU thread is a fiber with
// its own stack; a workgroup's fibers run on one OS thread, round-robin;
// __syncthreads() and the 64-lane wavefront operations are rendezvous points
// between fibers.  It models semantics (including 64-wide wavefronts), not
// timing.  The real library (libirs_hip.so) is never built from this file and
// the package never loads the emulated one (tests/test_layout.py checks).
#pragma once
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

extern "C" void irs_sim_switch(void** save_sp, void* load_sp);

namespace sim {

constexpr unsigned kMaxThreads = 1024;
constexpr size_t kStackBytes = 64 * 1024;
constexpr size_t kMaxSmem = 160 * 1024;

struct Worker {
  void* sp[kMaxThreads];
  bool done[kMaxThreads];
  void* sched_sp = nullptr;
  unsigned n = 0, cur = 0, done_count = 0;
  const std::function<void()>* body = nullptr;
  // block barrier
  unsigned arrived = 0, gen = 0;
  // wave barriers
  unsigned w_arrived[kMaxThreads / 64] = {0}, w_gen[kMaxThreads / 64] = {0};
  uint64_t xbuf[kMaxThreads];
  char* stacks = nullptr;
  unsigned char* smem = nullptr;
  unsigned block_idx = 0, grid_dim = 0;
  unsigned long progress = 0;
};

inline thread_local Worker* W = nullptr;

[[noreturn]] inline void die(const char* msg) {
  std::fprintf(stderr, "hip_sim: %s\n", msg);
  std::abort();
}

inline void yield() { irs_sim_switch(&W->sp[W->cur], W->sched_sp); }

inline void fiber_entry() {
  Worker* w = W;
  (*w->body)();
  w = W;
  w->done[w->cur] = true;
  ++w->done_count;
  ++w->progress;
  for (;;) yield();
}

inline void run_block(Worker* w, unsigned nthreads, const std::function<void()>& body) {
  if (nthreads == 0 || nthreads > kMaxThreads) die("bad block size");
  if (!w->stacks) {
    w->stacks = static_cast<char*>(std::aligned_alloc(64, kStackBytes * kMaxThreads));
    w->smem = static_cast<unsigned char*>(std::aligned_alloc(64, kMaxSmem));
    if (!w->stacks || !w->smem) die("out of memory");
  }
  w->n = nthreads;
  w->body = &body;
  w->done_count = 0;
  w->arrived = 0;
  w->gen = 0;
  std::memset(w->w_arrived, 0, sizeof w->w_arrived);
  std::memset(w->w_gen, 0, sizeof w->w_gen);
  for (unsigned i = 0; i < nthreads; ++i) {
    w->done[i] = false;
    char* top = w->stacks + kStackBytes * (i + 1);
    void** s = reinterpret_cast<void**>(reinterpret_cast<uintptr_t>(top) & ~uintptr_t(15));
    *--s = nullptr;                                   // fake return address
    *--s = reinterpret_cast<void*>(&fiber_entry);     // popped by `ret`
    for (int r = 0; r < 6; ++r) *--s = nullptr;       // rbp rbx r12 r13 r14 r15
    w->sp[i] = s;
  }
  while (w->done_count < nthreads) {
    const unsigned long before = w->progress;
    for (unsigned i = 0; i < nthreads; ++i) {
      if (w->done[i]) continue;
      w->cur = i;
      irs_sim_switch(&w->sched_sp, w->sp[i]);
    }
    if (w->progress == before) die("deadlock: divergent barrier / wavefront operation");
  }
}

inline void sync_block() {
  Worker* w = W;
  ++w->progress;
  const unsigned g = w->gen;
  if (++w->arrived == w->n) {
    w->arrived = 0;
    ++w->gen;
  } else {
    while (w->gen == g) yield();
  }
}

inline void sync_wave() {
  Worker* w = W;
  ++w->progress;
  const unsigned wv = w->cur >> 6;
  const unsigned lanes = (w->n - 64 * wv) < 64 ? (w->n - 64 * wv) : 64;
  const unsigned g = w->w_gen[wv];
  if (++w->w_arrived[wv] == lanes) {
    w->w_arrived[wv] = 0;
    ++w->w_gen[wv];
  } else {
    while (w->w_gen[wv] == g) yield();
  }
}

struct Idx {
  struct X {
    operator unsigned() const { return W->cur; }
  };
  X x;
};
struct BIdx {
  struct X {
    operator unsigned() const { return W->block_idx; }
  };
  X x;
};
struct BDim {
  struct X {
    operator unsigned() const { return W->n; }
  };
  X x;
};
struct GDim {
  struct X {
    operator unsigned() const { return W->grid_dim; }
  };
  X x;
};

// persistent pool: each OS thread executes whole workgroups
struct Pool {
  std::vector<std::thread> threads;
  std::mutex m;
  std::condition_variable cv, cv_done;
  unsigned long epoch = 0;
  unsigned active = 0;
  bool stop = false;
  // current launch
  unsigned grid = 0, block = 0;
  const std::function<void()>* body = nullptr;
  std::atomic<unsigned> next{0};

  Pool() {
    unsigned n = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("IRS_SIM_THREADS")) n = unsigned(std::atoi(e));
    if (n == 0) n = 1;
    if (n > 16) n = 16;
    for (unsigned i = 0; i < n; ++i) threads.emplace_back([this] { loop(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> l(m);
      stop = true;
    }
    cv.notify_all();
    for (auto& t : threads) t.join();
  }
  void loop() {
    Worker* w = new Worker;
    W = w;
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return stop || epoch != seen; });
        if (stop) break;
        seen = epoch;
      }
      for (;;) {
        const unsigned b = next.fetch_add(1);
        if (b >= grid) break;
        w->block_idx = b;
        w->grid_dim = grid;
        run_block(w, block, *body);
      }
      {
        std::lock_guard<std::mutex> l(m);
        if (--active == 0) cv_done.notify_all();
      }
    }
    std::free(w->stacks);
    std::free(w->smem);
    delete w;
  }
  // (one launch at a time: the product queues runs from a worker thread of its own while the
  // caller's thread may launch too — on a GPU the streams sort that out, here a lock does)
  std::mutex launch_m;
  void launch(unsigned g, unsigned b, const std::function<void()>& fn) {
    std::lock_guard<std::mutex> one(launch_m);
    std::unique_lock<std::mutex> l(m);
    grid = g;
    block = b;
    body = &fn;
    next = 0;
    active = unsigned(threads.size());
    ++epoch;
    cv.notify_all();
    cv_done.wait(l, [&] { return active == 0; });
  }
};

inline Pool& pool() {
  static Pool p;
  return p;
}

inline void launch(unsigned grid, unsigned block, size_t shmem, const std::function<void()>& fn) {
  if (shmem > kMaxSmem) die("dynamic LDS request exceeds 160 KiB");
  if (grid == 0) return;
  pool().launch(grid, block, fn);
}

inline unsigned char* dyn_smem() { return W->smem; }

}  // namespace sim

#define threadIdx (sim::Idx{})
#define blockIdx (sim::BIdx{})
#define blockDim (sim::BDim{})
#define gridDim (sim::GDim{})

inline void __syncthreads() { sim::sync_block(); }

// ---- wavefront (64 lanes) cross-lane operations -----------------------------
template<typename T>
inline T sim_exchange(T v, unsigned src_lane_abs, bool take) {
  static_assert(sizeof(T) <= 8, "");
  sim::Worker* w = sim::W;
  uint64_t raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  w->xbuf[w->cur] = raw;
  sim::sync_wave();
  T out = v;
  if (take) {
    raw = w->xbuf[src_lane_abs];
    std::memcpy(&out, &raw, sizeof(T));
  }
  sim::sync_wave();
  return out;
}
template<typename T>
inline T __shfl_up(T v, unsigned delta, int = 64) {
  const unsigned tid = sim::W->cur, lane = tid & 63u;
  return sim_exchange(v, tid - delta, lane >= delta);
}
template<typename T>
inline T __shfl_down(T v, unsigned delta, int = 64) {
  const unsigned tid = sim::W->cur, lane = tid & 63u;
  const unsigned lanes = (sim::W->n - (tid & ~63u)) < 64 ? (sim::W->n - (tid & ~63u)) : 64;
  return sim_exchange(v, tid + delta, lane + delta < lanes);
}
template<typename T>
inline T __shfl_xor(T v, int mask, int = 64) {
  const unsigned tid = sim::W->cur;
  return sim_exchange(v, (tid & ~63u) | ((tid ^ unsigned(mask)) & 63u), true);
}
template<typename T>
inline T __shfl(T v, int src, int = 64) {
  const unsigned tid = sim::W->cur;
  return sim_exchange(v, (tid & ~63u) | (unsigned(src) & 63u), true);
}
inline unsigned long long __ballot(int pred) {
  sim::Worker* w = sim::W;
  const unsigned tid = w->cur, base = tid & ~63u;
  const unsigned lanes = (w->n - base) < 64 ? (w->n - base) : 64;
  w->xbuf[tid] = pred ? 1u : 0u;
  sim::sync_wave();
  unsigned long long m = 0;
  for (unsigned l = 0; l < lanes; ++l) m |= (unsigned long long)(w->xbuf[base + l] & 1u) << l;
  sim::sync_wave();
  return m;
}

inline unsigned __float_as_uint(float f) {
  unsigned u;
  std::memcpy(&u, &f, 4);
  return u;
}
inline float __uint_as_float(unsigned u) {
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// ---- atomics (global memory is shared between OS threads) --------------------
inline unsigned atomicAdd(unsigned* p, unsigned v) {
  return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
  return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
inline float atomicAdd(float* p, float v) {
  uint32_t* ip = reinterpret_cast<uint32_t*>(p);
  uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED);
  for (;;) {
    float f;
    std::memcpy(&f, &old, 4);
    f += v;
    uint32_t nv;
    std::memcpy(&nv, &f, 4);
    if (__atomic_compare_exchange_n(ip, &old, nv, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
      float r;
      std::memcpy(&r, &old, 4);
      return r;
    }
  }
}
inline unsigned atomicMin(unsigned* p, unsigned v) {
  unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v < old &&
         !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
  return old;
}
inline unsigned atomicMax(unsigned* p, unsigned v) {
  unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v > old &&
         !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
  return old;
}
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) {
  unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v > old &&
         !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
  return old;
}
inline unsigned atomicCAS(unsigned* p, unsigned expected, unsigned desired) {
  __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
  return expected;
}
inline unsigned atomicOr(unsigned* p, unsigned v) {
  return __atomic_fetch_or(p, v, __ATOMIC_RELAXED);
}
