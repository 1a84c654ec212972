// irs_hip.hip — host side of the C ABI declared in include/irs_hip.h:
// staging into HBM, kernel orchestration on a HIP stream, result hand-back.
// Nothing here touches oracle/; there is no CPU execution path.
#include "irs_hip.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "gpu_rt.h"
#include "kernels.h"
#include "phrase.h"
#include "score.h"
#include "conj.h"
#include "join.h"

using namespace irs_hip;

namespace {

constexpr uint32_t kDefaultStride = 64;
constexpr uint32_t kPilotMargin = 3;   // estimated threshold: aim at margin * k candidates
constexpr uint32_t kDefaultWgThreads = 512;  // 8 wavefronts share one tile (measured best)

// Freed device / page-locked memory is kept per device and handed out again (size classes of
// 1/8 of a power of two): hipMalloc, hipFree and hipHostMalloc cost 0.1 - 1 ms apiece and hipFree
// synchronises the device — a batch that is created, run once and destroyed (the normal life of
// a batch) would spend more time in the allocator than in its kernels.  Whoever returns a block
// has made sure no queued work still touches it (irs_hip_batch_destroy waits for the batch's own
// events).  irs_hip_device_trim() gives everything back to the runtime.
namespace pool {
constexpr int kMaxDevices = 16;
struct Bin {
  std::mutex m;
  std::multimap<size_t, void*> blocks;   // capacity -> block
  size_t cached = 0;
};
inline Bin& bin(int device, bool pinned) {
  static Bin bins[2][kMaxDevices];
  return bins[pinned ? 1 : 0][device >= 0 && device < kMaxDevices ? device : 0];
}
inline size_t size_class(size_t n) {
  size_t step = 4096;
  while (step * 16 <= n) step <<= 1;   // step = 2^floor(log2 n) / 8 for n >= 64 KB
  return (std::max<size_t>(n, 1) + step - 1) / step * step;
}
// What stays with the library per device: device memory up to rt::pool_cap_bytes() — the buffers
// of the batches a pipelined serving loop has alive (a config-5 step holds two batches of ~6 GB, and
// three steps overlap; a cap of 16 GB was tried: blocks then go back to the runtime, hipFree
// synchronises the device and the steps stall — 45 -> 150 ms) —, page-locked HOST memory up to 4 GB:
// a batch pins a few MB of tables and its results (8 MB for 1000 x top-1000), and pinned pages are
// taken from every process on the node (8 ranks x the old 64 GB default was the whole host).
// IRS_HIP_POOL_MB / IRS_HIP_PINNED_POOL_MB override.
inline size_t cap_bytes(bool pinned) {
  if (const char* e = std::getenv(pinned ? "IRS_HIP_PINNED_POOL_MB" : "IRS_HIP_POOL_MB"))
    return size_t(std::atoll(e)) << 20;
  return pinned ? std::min<size_t>(rt::pool_cap_bytes(), size_t(4) << 30) : rt::pool_cap_bytes();
}
// A closing segment's memory goes back to the runtime, not into the pool: it is hundreds of MB in
// sizes no batch asks for (irs_hip_segment_close sets this around its destructor).
inline thread_local bool tl_free_now = false;
inline void release_all(int device, bool pinned) {
  Bin& b = bin(device, pinned);
  std::lock_guard<std::mutex> lock(b.m);
  for (auto& kv : b.blocks) pinned ? rt::hfree(kv.second) : rt::dfree(kv.second);
  b.blocks.clear();
  b.cached = 0;
}
// `*cap` = the block's capacity (what give() wants back)
inline void* take(int device, bool pinned, size_t bytes, size_t* cap) {
  const size_t want = size_class(bytes);
  Bin& b = bin(device, pinned);
  {
    std::lock_guard<std::mutex> lock(b.m);
    auto it = b.blocks.lower_bound(want);
    if (it != b.blocks.end() && it->first <= want + want / 4) {
      void* p = it->second;
      *cap = it->first;
      b.cached -= it->first;
      b.blocks.erase(it);
      rt::poison(p, *cap);
      return p;
    }
  }
  void* p = pinned ? rt::hmalloc(want) : rt::dmalloc(want);
  if (!p) {   // out of memory with blocks of other sizes lying around: give them back first
    release_all(device, pinned);
    p = pinned ? rt::hmalloc(want) : rt::dmalloc(want);
  }
  *cap = p ? want : 0;
  return p;
}
inline void give(int device, bool pinned, void* p, size_t cap) {
  if (!p) return;
  Bin& b = bin(device, pinned);
  if (!tl_free_now) {
    std::lock_guard<std::mutex> lock(b.m);
    if (b.cached + cap <= cap_bytes(pinned)) {
      b.blocks.emplace(cap, p);
      b.cached += cap;
      return;
    }
  }
  pinned ? rt::hfree(p) : rt::dfree(p);
}
}  // namespace pool

template<bool PINNED>
struct PoolBuf {  // owning allocation out of the pool of the device that was current at alloc()
  void* p = nullptr;
  size_t n = 0;     // bytes asked for
  size_t cap = 0;   // the block's capacity
  int device = 0;
  bool owned = true;   // false: a view into another PoolBuf (view())
  PoolBuf() = default;
  PoolBuf(const PoolBuf&) = delete;
  PoolBuf& operator=(const PoolBuf&) = delete;
  PoolBuf(PoolBuf&& o) noexcept : p(o.p), n(o.n), cap(o.cap), device(o.device), owned(o.owned) {
    o.p = nullptr;
    o.n = o.cap = 0;
  }
  ~PoolBuf() { release(); }
  bool alloc(size_t bytes) {
    if (p && owned && bytes <= cap && pool::size_class(bytes) == cap) {   // the same block would come back
      n = bytes;
      return true;
    }
    release();
    device = rt::current_device();
    p = pool::take(device, PINNED, bytes, &cap);
    n = p ? bytes : 0;
    return p != nullptr;
  }
  void release() {
    if (owned) pool::give(device, PINNED, p, cap);
    p = nullptr;
    n = cap = 0;
    owned = true;
  }
  // `bytes` at `ptr` inside a block somebody else owns (and outlives this view)
  void view(void* ptr, size_t bytes) {
    release();
    p = ptr;
    n = bytes;
    owned = false;
  }
  template<typename T>
  T* as() const { return static_cast<T*>(p); }
};
using DevBuf = PoolBuf<false>;
using PinBuf = PoolBuf<true>;

// Host -> device uploads of a batch: the bytes are built in (or copied into) page-locked memory
// and go out with asynchronous copies on the stream of the batch's next run — no copy from
// pageable memory (the runtime stages those synchronously), no stream synchronisation, so a
// caller's host thread prepares batch i + 1 while the device still executes batch i.
struct Stager {
  struct Piece { void* dst; const void* src; size_t n; };
  std::vector<PinBuf> chunks;
  size_t used = 0;   // of chunks.back()
  std::vector<Piece> pending;
  // n bytes of page-locked memory that will be copied to `dst`: the caller fills them before
  // the next flush()
  void* put(void* dst, size_t n) {
    if (!n) return nullptr;
    const size_t need = (n + 63) & ~size_t(63);
    if (chunks.empty() || used + need > chunks.back().n) {
      PinBuf c;
      if (!c.alloc(std::max<size_t>(need, size_t(1) << 20))) return nullptr;
      chunks.push_back(std::move(c));
      used = 0;
    }
    void* at = chunks.back().as<uint8_t>() + used;
    used += need;
    pending.push_back(Piece{dst, at, n});
    return at;
  }
  bool copy(void* dst, const void* src, size_t n) {
    if (!n) return true;
    void* at = put(dst, n);
    if (!at) return false;
    std::memcpy(at, src, n);
    return true;
  }
  bool flush(rt::stream_t st) {
    bool ok = true;
    for (const Piece& p : pending) ok = ok && rt::h2d(p.dst, p.src, p.n, st);
    pending.clear();
    return ok;
  }
};

// One copy stream per device for the tables of a batch's FIRST run: queued on the caller's stream
// they would start only when the previous batch's kernels are through (0.2 ms of idle compute per
// step for the headline batch's 5 MB); on their own stream they travel while those kernels run,
// and the run waits for them by an event.
rt::stream_t copy_stream(int device, int which) {
  static std::mutex m;
  static std::map<int, rt::stream_t> streams;
  std::lock_guard<std::mutex> lock(m);
  const int key = device * 2 + which;
  auto it = streams.find(key);
  if (it != streams.end()) return it->second;
  rt::stream_t s = nullptr;
  if (!rt::stream_create(&s)) s = nullptr;   // (null: the caller keeps its own stream)
  streams[key] = s;
  return s;
}
rt::stream_t upload_stream(int device) { return copy_stream(device, 0); }
// ... and one for results on their way to page-locked host memory (irs_hip_batch_results_to_host):
// the copy of batch i travels while the kernels of batch i + 1 run
rt::stream_t download_stream(int device) { return copy_stream(device, 1); }

// IRS_HIP_TRACE=1: host-side stage times on stderr (what a batch costs before its first kernel)
struct HostTrace {
  const char* what;
  std::chrono::steady_clock::time_point t0;
  explicit HostTrace(const char* w) : what(w), t0(std::chrono::steady_clock::now()) {}
  ~HostTrace() {
    static const bool on = std::getenv("IRS_HIP_TRACE") != nullptr;
    if (on)
      std::fprintf(stderr, "[irs_hip] %s: %.1f us\n", what,
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
};

bool device_usable(int device) {
  if (device < 0 || device >= rt::device_count()) return false;
  char arch[64] = {0};
  if (!rt::device_arch(device, arch, sizeof arch)) return false;
  // gfx950 only (the sim runtime of the CPU test tier reports "gfx950-sim")
  if (std::strncmp(arch, "gfx950", 6) != 0) return false;
  return rt::set_device(device);
}

// format_utils::check_header for `.doc` (format_utils.cpp:74-105,
// formats_10.cpp:325-326, 3356-3361); returns header length or 0.
size_t check_header(const uint8_t* f, uint64_t len, const char* name, int32_t* version) {
  const size_t nlen = std::strlen(name);
  if (len < 4 + 1 + nlen + 4 + 16) return 0;
  const uint32_t magic = (uint32_t(f[0]) << 24) | (uint32_t(f[1]) << 16) |
                         (uint32_t(f[2]) << 8) | f[3];
  if (magic != 0x3fd76c17u) return 0;
  if (f[4] != nlen || std::memcmp(f + 5, name, nlen) != 0) return 0;
  const uint8_t* v = f + 5 + nlen;
  *version = int32_t((uint32_t(v[0]) << 24) | (uint32_t(v[1]) << 16) |
                     (uint32_t(v[2]) << 8) | v[3]);
  return 5 + nlen + 4;
}
size_t check_doc_header(const uint8_t* f, uint64_t len, int32_t* version) {
  return check_header(f, len, "iresearch_10_postings_documents", version);
}
// `.pos`: formats_10.cpp:327-328, 3369-3381
size_t check_pos_header(const uint8_t* f, uint64_t len, int32_t* version) {
  return check_header(f, len, "iresearch_10_postings_positions", version);
}

}  // namespace

struct irs_hip_segment {
  int device = 0;
  DevSegment dev{};
  DevBuf d_doc, d_norms, d_terms, d_blk_off, d_blk_last, d_blk_bits, d_status;
  DevBuf d_blk_aoff, d_pk;     // packed-payload image (DevSegment::pk) and its offsets
  DevBuf d_blk_dir;            // BlkDir per block
  DevBuf d_blk_term;           // the row's term
  DevBuf d_tail_docs, d_tail_freqs;  // decoded vint tails, [num_terms][128]
  // positions (fields with POS): `.pos` bytes, per-term records, pos block directory,
  // positions in front of every doc block, decoded position tails
  DevBuf d_pos, d_pterms, d_pblk_off, d_pblk_bits, d_blk_pos, d_ptail;
  std::vector<DevPosTerm> pterms;
  std::vector<DevTerm> terms;  // host mirror incl. the fields the dir kernel filled
  uint64_t total_blocks = 0;
  uint64_t device_bytes = 0;
  uint32_t cus = 1;  // compute units of the device (persistent grid sizing)
  // Candidate slots per unit that batches on this segment turned out to need (recover_overflow):
  // scores that tie by the thousand at the threshold bin (TF-IDF without norms: every doc with
  // the same frequencies) cannot be cut by a bin threshold; later batches start from what the
  // earlier ones learned instead of overflowing and running twice each.  Decays never; bounded
  // by default_cand_cap's ceiling.
  std::atomic<uint32_t> cand_cap_hint{0};
  // block-max data (WAND), built on first use: the one thing that changes after open
  std::mutex wand_mutex;
  bool wand_ready = false;
  // the norm byte of every posting in posting order (k_posting_norms), built on the segment's
  // first joined batch; 1-byte Norm2 columns only
  DevBuf d_pnorm, d_tail_norms;
  bool pnorm_ready = false;
  DevBuf d_dead;                   // the DocumentMask as a bitmap (DevSegment::dead)
  uint64_t live_docs = 0;          // num_docs - deleted docs
  DevBuf d_blk_maxf, d_blk_minn;
  std::vector<uint64_t> skip_at;   // per term: absolute offset of its skip data (0: none)
  bool has_pos = false;
  uint64_t wand_from_index = 0;    // blocks whose (max freq, min norm) came from the index's wand data
  uint32_t wand_type = 0;          // IRS_HIP_WAND_* of the scorer that wrote the wand data
};

struct irs_hip_comm {
  int device = 0;
  int n_ranks = 1, rank = 0;
  rt::comm::handle_t h = nullptr;
};

struct irs_hip_batch {
  irs_hip_segment* seg = nullptr;          // segs[0]: device, CU count
  std::vector<irs_hip_segment*> segs;      // a batch spans one or more segments of one device
  uint32_t nq_user = 0;                    // queries per segment
  uint32_t nq = 0 /* execution units = segments x queries */, jt = 0, k_max = 0;
  uint32_t tile = 0 /* 0 = pick by accumulator width */, stride = kDefaultStride, cand_cap = 0;
  bool estimate = true;  // k_pilot picks an estimated threshold (falls back to the sound one)
  uint32_t reruns = 0;   // recoveries so far (underflow or overflow re-runs)
  uint32_t n_tiles = 0;     // of the segment with the FEWEST tiles (pilot stride, recovery)
  uint32_t max_tiles = 0;   // ... with the most (chunk ids per unit)
  uint32_t stride_eff = 1;  // pilot stride actually used (>= 2 pilot tiles per segment when possible)
  uint32_t wg_threads = kDefaultWgThreads;  // threads per pilot/score workgroup
  bool any_and = false;    // some unit counts matches per doc in the tile kernels (min-match)
  bool wand = false;       // irs_hip_batch_set_wand
  // units by the kernels that execute them: doc tiles (Or, min-match) / lead blocks (And)
  // (all_tile_units: every doc-tile unit, fixed at create; ensure_scratch deals them to
  // join_units — plain disjunctions run as joined posting streams, join.h — and tile_units —
  // the rest, score.h's work items)
  // (all_conj_units: every conjunction, fixed at create; ensure_scratch deals them to
  // join_units — accumulators with match counts, join.h — and conj_units — block driven, conj.h)
  std::vector<uint32_t> all_tile_units, tile_units, join_units, conj_units, all_conj_units;
  std::vector<uint8_t> count_precise;   // [unit] match counts may share its 32-bit accumulators
  std::vector<uint32_t> conj_items;   // lead items of every conj unit
  uint32_t n_conj_wgs = 0;
  DevBuf d_tile_units, d_conj_units, d_conj_items, d_conj_hist;
  DevBuf d_conj_item_base, d_conj_unit_items, d_conj_seek, d_conj_recs;   // k_conj_seek
  DevBuf d_conj_lg;   // [conj unit] log2 of the pieces a lead block is cut into (ConjItem)
  DevBuf d_conj_item_hits;   // [lead item] matches (ConjArgs::item_hits, k_conj_hits)
  DevBuf d_lead_of;   // by_phrase: slot of every unit's lead term
  DevBuf d_min_bin;   // [unit] score bin of the caller's irs::score::Min (irs_hip_batch_set_min_scores)
  DevBuf d_min_score; // [unit] ... and the score itself (k_select's exact filter)
  bool has_min = false;
  // the caller's scores themselves: their bins are worked out when a run's tables go out — AFTER
  // ensure_scratch, which may change a unit's bin_scale (build_groups across ranks)
  std::vector<float> min_scores;   // [unit]
  bool min_dirty = false;
  uint32_t conj_total_items = 0;
  DevBuf d_conj_pilot;             // the lead items the pilot pass samples, {unit, item} each
  uint32_t n_conj_pilot = 0, conj_pilot_stride = 0;
  bool phrase = false;  // a batch of by_phrase queries (k_phrase instead of k_pilot + k_score)
  PinBuf h_pin;                // page-locked staging for irs_hip_batch_results
  uint32_t n_phrase_wgs = 0;   // k_phrase workgroups: kPhraseWaves lead blocks each
  bool acc32 = true;   // 32-bit fixed-point accumulators are precise enough for every query
  bool scratch_ready = false;
  std::vector<DevQuery> queries;
  std::vector<DevQTerm> qterms;
  DevBuf d_segs, d_queries, d_qterms, d_first, d_tails, d_bstar, d_cands, d_cand_count, d_hits,
    d_out, d_out_count, d_status, d_work;
  // work-item lists of the doc tiles (score.h): per-tile item offsets (+ scan scratch) and
  // the 32-byte records themselves
  DevBuf d_tile_off, d_scan_parts, d_items, d_score_args, d_tile_ub;
  DevBuf d_pruned;    // [unit] u32: block-max pruning skipped something of the unit in this run
  DevBuf d_zeroed;    // owns d_status, d_bstar, d_cand_count, d_hits, d_touched, d_pruned (views): one fill per run
  DevBuf d_touched;   // [unit][2] u64: bytes decoded / positions read by the block-driven kernels
  ScoreArgs score_args{}, score_args_sent{};
  bool score_args_valid = false;
  uint32_t total_tiles = 0;    // doc tiles of all units
  uint32_t score_threads = 0;  // threads per k_pilot / k_score workgroup (power of two x 64)
  uint32_t nw_log2 = 3;        // log2(wavefronts per such workgroup)
  uint64_t alg_bytes = 0, postings = 0;
  // joined posting streams (join.h): every distinct (segment, term) of the batch decoded once
  // per run.  path_pref: irs_hip_batch_set_path (0 auto, 1 work items, 2 joined streams).
  int path_pref = 0;
  uint32_t tile_asked = 0;     // irs_hip_batch_configure's tile (0: ensure_scratch picks one per deal)
  bool joined = false;
  DevBuf d_streams, d_join_wgs, d_jterms, d_entries, d_bounds, d_join_args, d_join_units,
    d_join_order;
  uint32_t join_max_tiles = 0;
  uint32_t n_streams = 0, n_join_wgs = 0;
  uint32_t join_threads = 1024, join_nw_log2 = 4;   // threads per k_join_pilot / k_join_score workgroup
  uint64_t join_entries = 0;
  // one threshold per query for its units on the batch's segments (irs_hip_batch_set_shared_threshold)
  bool shared_threshold = false;
  bool pairs_allowed = true;   // irs_hip_batch_set_paired_tiles (0: never; 1: by size; 2: whatever the size)
  bool pairs_forced = false;
  bool pairs_used = false;     // ... and whether the last run's plain disjunctions took them
  uint32_t n_groups = 0;       // groups in force this run (0: none)
  DevBuf d_group_of;           // [unit] group + 1, 0: a threshold of its own
  DevBuf d_group_members;      // [nq_user][n_segs] unit or 0xFFFFFFFF
  DevBuf d_group_hist;         // [nq_user][kBins + 2]
  DevBuf d_group_sums;         // [nq_user][kGroupSumWords] + 2 status counters (k_group_sums)
  // ... across ranks (irs_hip_batch_set_comm): the group histograms and the group sums are summed
  // over the communicator's ranks inside every run
  irs_hip_comm* comm = nullptr;
  DevBuf d_agree;   // one word: the ranks' vote before a collective re-run (all_ranks_can)
  std::vector<double> group_upper;   // [unit] a score bound that is the same on every segment, 0: none
  JoinArgs join_args[2]{};   // plain disjunctions / units with match counts
  JoinArgs join_args_sent[2]{};   // ... as the device last got them
  bool join_args_valid[2] = {false, false};
  uint32_t n_join_plain = 0; // join_units in d_join_order: the plain ones first
  uint32_t join_first[2][kJoinQueues + 1]{};   // [launch] the queues' first slots in d_join_order
  DevBuf d_join_ctr;                           // [launch][kJoinQueues] work counters
  uint32_t join_ctr_init[2][kJoinQueues]{};
  bool profile = false;
  bool count_touched = false;   // irs_hip_batch_profile bit 1: the kernels count what they decode
  bool events_ready = false;
  rt::event_t ev[2 * IRS_HIP_K_COUNT];
  // what verify_run waits for: the batch's OWN last run (not whatever else the caller has
  // queued on the stream since), and the status word that run left in page-locked memory
  rt::event_t ev_done{};
  bool ev_done_ready = false;
  // irs_hip_batch_plan: the planning stage of the NEXT run was queued ahead (on another stream)
  rt::event_t ev_planned{};
  bool ev_planned_ready = false;
  bool planned = false;        // ... and the next run may use it (same geometry)
  // a plan stage is queued on some stream and may still be running — whether or not the next run
  // will use its tables (`planned` is dropped by every setter that re-deals the units; the kernels
  // it queued keep reading and writing the batch's buffers until ev_planned)
  bool plan_pending = false;
  uint32_t* h_status = nullptr;
  PinBuf h_status_buf;
  rt::stream_t stream = nullptr;
  bool ran = false;
  // host -> device tables of the batch: built in page-locked memory, sent with the next run
  Stager up;
  bool slack_zeroed = false;   // the readable slack behind d_entries
  // the last copy OUT of the batch's buffers queued by irs_hip_batch_results_to_device (destroy
  // waits for it and for ev_done — never for the stream, which may hold other batches' work)
  rt::event_t ev_used{};
  bool ev_used_ready = false, ev_used_pending = false;
  rt::event_t ev_up{};         // the first run's table uploads (the device's copy stream)
  bool ev_up_ready = false;
  // irs_hip_batch_results_to_host: hits, counts and totals in page-locked memory of the batch
  PinBuf h_res;
  rt::event_t ev_host{};
  bool ev_host_ready = false, host_pending = false;
  // irs_hip_batch_run hands the host half of a run (units dealt, streams and work lists built,
  // uploads and launches queued: ~1 ms for 1000 queries) to the device's worker thread and returns;
  // every other entry point waits here for it first.  async_rc: what that run returned.
  std::mutex am;
  std::condition_variable acv;
  bool async_pending = false;
  int async_rc = 0;
  int async_pref = -1;   // irs_hip_batch_set_async: -1 the process default (IRS_HIP_ASYNC_RUN), 0 / 1
};

namespace {

// Packed-payload image: block sizes (left in blk_aoff by the directory kernel) ->
// exclusive prefix sum in place -> one copy pass.  Everything on the device.
// In-place exclusive prefix sum of n u32 values on the device; *total = their sum, which
// must fit 32 bits (the scanned values are offsets kept as u32).
int scan_exclusive(uint32_t* d_values, uint64_t n, uint64_t* total, bool may_wrap = false) {
  *total = 0;
  if (!n) return IRS_HIP_OK;
  const uint32_t parts = uint32_t((n + kScanChunk - 1) / kScanChunk);
  DevBuf totals;
  if (!totals.alloc((uint64_t(parts) + 1) * 8)) return IRS_HIP_ENOMEM;
  RT_LAUNCH(k_scan_totals, parts, kThreads, 0, nullptr, d_values, n, totals.as<uint64_t>());
  RT_LAUNCH(k_scan_parts, 1, 64, 0, nullptr, totals.as<uint64_t>(), parts);
  if (!rt::last_error_ok() || !rt::d2h(total, totals.as<uint64_t>() + parts, 8, nullptr) ||
      !rt::sync(nullptr))
    return IRS_HIP_EHIP;
  // (offsets kept as u32; sums that are only ever used as DIFFERENCES may wrap mod 2^32)
  if (*total > 0xFFFFFFFFull && !may_wrap) return IRS_HIP_EUNSUPPORTED;
  RT_LAUNCH(k_scan_apply, parts, kThreads, 0, nullptr, d_values, n, totals.as<uint64_t>());
  if (!rt::last_error_ok() || !rt::sync(nullptr)) return IRS_HIP_EHIP;
  return IRS_HIP_OK;
}

int build_packed_image(irs_hip_segment* s) {
  const uint64_t n = s->total_blocks;
  uint64_t total_units = 0;  // offsets are u32 units of 16 bytes (64 GB)
  if (const int rc = scan_exclusive(s->d_blk_aoff.as<uint32_t>(), n, &total_units)) return rc;
  if (n) {
    RT_LAUNCH(k_dir_aoff, uint32_t((n + kThreads - 1) / kThreads), kThreads, 0, nullptr,
              s->d_blk_aoff.as<uint32_t>(), n, s->d_blk_dir.as<BlkDir>());
    if (!rt::last_error_ok()) return IRS_HIP_EHIP;
  }
  const uint64_t bytes = total_units * 16;
  if (!s->d_pk.alloc(bytes + kPadBytes)) return IRS_HIP_ENOMEM;
  if (!rt::dmemset(s->d_pk.as<uint8_t>() + bytes, 0, kPadBytes, nullptr)) return IRS_HIP_EHIP;
  s->dev.pk = s->d_pk.as<uint8_t>();
  if (total_units && n) {
    RT_LAUNCH(k_pack_payloads, row_grid(n, s->cus), kThreads, 0, nullptr, s->dev, n,
              s->d_pk.as<uint8_t>());
    if (!rt::last_error_ok() || !rt::sync(nullptr)) return IRS_HIP_EHIP;
  }
  return IRS_HIP_OK;
}

template<int LAYOUT>
int build_directory(irs_hip_segment* s) {
  const uint32_t grid = s->dev.num_terms;   // a workgroup per term
  if (!rt::dmemset(s->d_status.p, 0, 4, nullptr)) return IRS_HIP_EHIP;
  if (grid) {
    RT_LAUNCH((k_build_directory<LAYOUT>), grid, kChainThreads, 0, nullptr, s->dev,
              s->d_terms.as<DevTerm>(), s->d_blk_off.as<uint32_t>(),
              s->d_blk_last.as<uint32_t>(), s->d_blk_bits.as<uint16_t>(),
              s->d_blk_aoff.as<uint32_t>(), s->d_blk_dir.as<BlkDir>(),
              s->d_blk_term.as<uint32_t>(),
              s->d_tail_docs.as<uint32_t>(), s->d_tail_freqs.as<uint32_t>(),
              s->d_status.as<uint32_t>());
  }
  if (!rt::last_error_ok()) return IRS_HIP_EHIP;
  uint32_t status = 0;
  if (!rt::d2h(&status, s->d_status.p, 4, nullptr) ||
      !rt::d2h(s->terms.data(), s->d_terms.p, s->terms.size() * sizeof(DevTerm), nullptr) ||
      !rt::sync(nullptr))
    return IRS_HIP_EHIP;
  if (status & kStatusCorrupt) return IRS_HIP_ECORRUPT;
  for (const DevTerm& t : s->terms) {
    if (t.docs_count && (t.last_doc > s->dev.num_docs || t.last_doc < kDocMin))
      return IRS_HIP_ECORRUPT;
  }
  return build_packed_image(s);
}

// Positions: frequency sums per doc block -> exclusive scan (blk_pos), then the pos block
// directory and the decoded position tails.  `pos_end` = term_meta::pos_end per term.
template<int LAYOUT>
int build_positions(irs_hip_segment* s, const std::vector<uint64_t>& pos_end) {
  const uint64_t n = s->total_blocks;
  if (!rt::dmemset(s->d_blk_pos.p, 0, s->d_blk_pos.n, nullptr)) return IRS_HIP_EHIP;
  if (n && s->dev.num_terms) {
    RT_LAUNCH((k_freq_sums<LAYOUT>), row_grid(n, s->cus), kThreads, 0, nullptr, s->dev,
              n, s->d_blk_pos.as<uint32_t>());
    if (!rt::last_error_ok() || !rt::sync(nullptr)) return IRS_HIP_EHIP;
  }
  uint64_t total = 0;
  // (position numbers are per term: differences of blk_pos, every term's total < 2^32)
  if (const int rc = scan_exclusive(s->d_blk_pos.as<uint32_t>(), n, &total, true)) return rc;
  const uint32_t total32 = uint32_t(total);  // sentinel row: everything in front of "row n"
  if (!rt::h2d(s->d_blk_pos.as<uint32_t>() + n, &total32, 4, nullptr)) return IRS_HIP_EHIP;
  DevBuf d_pos_end;
  if (!d_pos_end.alloc(std::max<size_t>(1, pos_end.size()) * 8)) return IRS_HIP_ENOMEM;
  if (!rt::h2d(d_pos_end.p, pos_end.data(), pos_end.size() * 8, nullptr) ||
      !rt::dmemset(s->d_status.p, 0, 4, nullptr))
    return IRS_HIP_EHIP;
  const uint32_t grid = (s->dev.num_terms + kWaves - 1) / kWaves;
  if (grid) {
    // a term_meta::freq that disagrees with the decoded frequencies would send the position
    // kernels past their buffers: refuse the segment (IRS_HIP_ECORRUPT)
    RT_LAUNCH(k_check_freq_totals, (s->dev.num_terms + kThreads - 1) / kThreads, kThreads, 0,
              nullptr, s->dev, s->d_pterms.as<DevPosTerm>(), s->d_status.as<uint32_t>());
    RT_LAUNCH(k_pos_directory, s->dev.num_terms, kChainThreads, 0, nullptr, s->dev, s->d_pterms.as<DevPosTerm>(),
              s->d_pblk_off.as<uint32_t>(), s->d_pblk_bits.as<uint8_t>(),
              s->d_ptail.as<uint32_t>(), d_pos_end.as<uint64_t>(), s->d_status.as<uint32_t>());
  }
  uint32_t status = 0;
  if (!rt::last_error_ok() || !rt::d2h(&status, s->d_status.p, 4, nullptr) ||
      !rt::d2h(s->pterms.data(), s->d_pterms.p, s->pterms.size() * sizeof(DevPosTerm), nullptr) ||
      !rt::sync(nullptr))
    return IRS_HIP_EHIP;
  return (status & kStatusCorrupt) ? IRS_HIP_ECORRUPT : IRS_HIP_OK;
}

template<typename K>
bool big_smem(K kernel, size_t bytes) {
  return rt::allow_dynamic_smem(reinterpret_cast<const void*>(kernel), bytes);
}

// Candidate slots per query.  An estimated threshold aims at kPilotMargin * k
// candidates; the sound one admits about k * (pilot stride).
static const uint32_t* min_bins(const irs_hip_batch* b) {
  return b->has_min ? b->d_min_bin.as<uint32_t>() : nullptr;
}

static uint32_t default_cand_cap(const irs_hip_batch* b) {
  const uint64_t per_k = b->estimate ? 16ull : 4ull * b->stride_eff;
  uint64_t learned = 0;   // (block-driven units only: tile units cut ties by their per-tile staging)
  if (b->phrase || !b->all_conj_units.empty())
    for (const irs_hip_segment* sg : b->segs) learned = std::max<uint64_t>(learned, sg->cand_cap_hint.load());
  return uint32_t(std::min<uint64_t>(std::max<uint64_t>({per_k * b->k_max, 16384, learned}), 262144));
}

// launch helpers: one instantiation per (accumulator width, layout, tile, AND)
template<typename ACC, int LAYOUT, int TILE, bool AND>
bool launch_pilot(irs_hip_batch* b, rt::stream_t st) {
  const size_t smem = tile_smem_bytes<ACC, TILE, AND>() + kBins * sizeof(uint32_t);
  auto kern = k_pilot<ACC, LAYOUT, TILE, AND>;
  if (!big_smem(kern, smem)) return false;
  RT_LAUNCH(kern, uint32_t(b->tile_units.size()), b->score_threads, smem, st,
            b->d_tile_units.as<uint32_t>(), b->d_segs.as<DevSegment>(),
            b->d_queries.as<DevQuery>(), b->d_qterms.as<DevQTerm>(), b->stride_eff,
            b->nw_log2, b->d_tile_off.as<uint32_t>(), reinterpret_cast<uint64_t>(b->d_items.p),
            b->d_bstar.as<uint32_t>(), b->estimate ? kPilotMargin : 0u, min_bins(b));
  return rt::last_error_ok();
}

template<typename ACC, int LAYOUT, int TILE, bool AND>
bool launch_score(irs_hip_batch* b, rt::stream_t st) {
  const size_t smem = score_smem_bytes<ACC, TILE, AND>();
  auto kern = k_score<ACC, LAYOUT, TILE, AND>;
  if (!big_smem(kern, smem)) return false;
  // persistent grid: as many workgroups as stay resident on the chip at once
  const uint32_t waves = b->score_threads / 64;
  uint32_t per_cu = uint32_t((160u * 1024u) / smem);
  per_cu = std::max<uint32_t>(1, std::min<uint32_t>(per_cu, 16u / waves));  // 128 VGPRs: 4 waves/SIMD
  const uint32_t cpq = (b->max_tiles + kChunkTiles - 1) / kChunkTiles;  // chunk ids per unit
  const uint32_t n_units = uint32_t(b->tile_units.size());
  const uint64_t chunks = uint64_t(n_units) * cpq;
  if (chunks > 0xFFFF0000ull) return false;
  const uint32_t grid = uint32_t(std::min<uint64_t>(chunks, uint64_t(b->seg->cus) * per_cu));
  ScoreArgs& a = b->score_args;   // read by the kernel from device memory (score.h)
  a.segs = b->d_segs.as<DevSegment>();
  a.queries = b->d_queries.as<DevQuery>();
  a.qterms = b->d_qterms.as<DevQTerm>();
  a.tile_off = b->d_tile_off.as<uint32_t>();
  a.items = reinterpret_cast<uint64_t>(b->d_items.p);
  a.bstar = b->d_bstar.as<uint32_t>();
  a.cands = b->d_cands.as<uint64_t>();
  a.cand_count = b->d_cand_count.as<uint32_t>();
  a.hits = b->d_hits.as<unsigned long long>();
  a.work_counter = b->d_work.as<uint32_t>();
  a.tile_ub = b->wand ? b->d_tile_ub.as<float>() : nullptr;
  a.pruned = b->d_pruned.as<uint32_t>();
  a.cpq = cpq;
  a.n_units = n_units;
  a.nw_log2 = b->nw_log2;
  a.cand_cap = b->cand_cap;
  // (the arguments only change with the batch's geometry or a regrown candidate buffer)
  if (std::memcmp(&a, &b->score_args_sent, sizeof a) != 0 || !b->score_args_valid) {
    if (!b->up.copy(b->d_score_args.p, &a, sizeof a) || !b->up.flush(st)) return false;
    std::memcpy(&b->score_args_sent, &a, sizeof a);
    b->score_args_valid = true;
  }
  if (!rt::dmemset(b->d_work.p, 0, 4, st)) return false;
  RT_LAUNCH(kern, grid, b->score_threads, smem, st, reinterpret_cast<uint64_t>(b->d_score_args.p));
  return rt::last_error_ok();
}

// LDS byte offset of the table rows in the tile kernels' layout (what k_items_fill writes
// into the work items' `tab` field)
template<typename ACC, int TILE>
uint32_t caches_off_and(const irs_hip_batch* b) {
  return b->any_and ? TileOff<ACC, TILE, true>::caches : TileOff<ACC, TILE, false>::caches;
}
template<typename ACC>
uint32_t caches_off_tile(const irs_hip_batch* b) {
  switch (b->tile) {
    case 12288: return caches_off_and<ACC, 12288>(b);
    case 8192: return caches_off_and<ACC, 8192>(b);
    case 6144: return caches_off_and<ACC, 6144>(b);
    default: return caches_off_and<ACC, 4096>(b);
  }
}
uint32_t caches_off(const irs_hip_batch* b) {
  return b->acc32 ? caches_off_tile<uint32_t>(b) : caches_off_tile<unsigned long long>(b);
}

template<typename ACC, int LAYOUT, int TILE>
bool launch_pilot_and(irs_hip_batch* b, rt::stream_t st) {
  return b->any_and ? launch_pilot<ACC, LAYOUT, TILE, true>(b, st)
                    : launch_pilot<ACC, LAYOUT, TILE, false>(b, st);
}
template<typename ACC, int LAYOUT, int TILE>
bool launch_score_and(irs_hip_batch* b, rt::stream_t st) {
  return b->any_and ? launch_score<ACC, LAYOUT, TILE, true>(b, st)
                    : launch_score<ACC, LAYOUT, TILE, false>(b, st);
}
template<typename ACC, int LAYOUT>
bool launch_pilot_tile(irs_hip_batch* b, rt::stream_t st) {
  switch (b->tile) {
    case 12288: return launch_pilot_and<ACC, LAYOUT, 12288>(b, st);
    case 8192: return launch_pilot_and<ACC, LAYOUT, 8192>(b, st);
    case 6144: return launch_pilot_and<ACC, LAYOUT, 6144>(b, st);
    default: return launch_pilot_and<ACC, LAYOUT, 4096>(b, st);
  }
}
template<typename ACC, int LAYOUT>
bool launch_score_tile(irs_hip_batch* b, rt::stream_t st) {
  switch (b->tile) {
    case 12288: return launch_score_and<ACC, LAYOUT, 12288>(b, st);
    case 8192: return launch_score_and<ACC, LAYOUT, 8192>(b, st);
    case 6144: return launch_score_and<ACC, LAYOUT, 6144>(b, st);
    default: return launch_score_and<ACC, LAYOUT, 4096>(b, st);
  }
}
template<int LAYOUT>
bool launch_pilot_acc(irs_hip_batch* b, rt::stream_t st) {
  return b->acc32 ? launch_pilot_tile<uint32_t, LAYOUT>(b, st)
                  : launch_pilot_tile<unsigned long long, LAYOUT>(b, st);
}
template<int LAYOUT>
bool launch_score_acc(irs_hip_batch* b, rt::stream_t st) {
  return b->acc32 ? launch_score_tile<uint32_t, LAYOUT>(b, st)
                  : launch_score_tile<unsigned long long, LAYOUT>(b, st);
}

bool ensure_pilot_list(irs_hip_batch* b, uint32_t stride, rt::stream_t st);

// Conjunctions: [pilot pass over every P-th lead block -> threshold bins] -> full pass.
template<int LAYOUT>
bool launch_conj(irs_hip_batch* b, rt::stream_t st) {
  if (b->n_conj_wgs == 0) return true;
  ConjArgs a{};
  a.segs = b->d_segs.as<DevSegment>();
  a.queries = b->d_queries.as<DevQuery>();
  a.qterms = b->d_qterms.as<DevQTerm>();
  a.wgs = nullptr;
  a.n_items = b->conj_total_items;
  a.tails = b->d_tails.as<DevTail>();
  a.bstar = b->d_bstar.as<uint32_t>();
  a.cands = b->d_cands.as<uint64_t>();
  a.cand_count = b->d_cand_count.as<uint32_t>();
  a.hits = b->d_hits.as<unsigned long long>();
  a.hist = b->d_conj_hist.as<uint32_t>();
  a.touched = b->count_touched ? b->d_touched.as<unsigned long long>() : nullptr;
  a.seek = b->d_conj_seek.as<uint32_t>();
  a.recs = b->d_conj_recs.as<ConjItem>();
  a.unit_items = b->d_conj_unit_items.as<uint32_t>();
  a.jt = b->jt;
  a.cand_cap = b->cand_cap;
  a.pilot_stride = b->stride_eff;
  a.wand = b->wand ? 1u : 0u;
  a.pruned = b->d_pruned.as<uint32_t>();
  a.item_hits = b->d_conj_item_hits.as<uint32_t>();
  if (!ensure_pilot_list(b, a.pilot_stride, st)) return false;
  if (!rt::dmemset(b->d_conj_hist.p, 0, b->d_conj_hist.n, st) ||
      !rt::dmemset(b->d_conj_item_hits.p, 0, b->d_conj_item_hits.n, st))
    return false;
  RT_LAUNCH(k_conj_seek, (b->conj_total_items + kThreads - 1) / kThreads, kThreads, 0, st,
            b->d_segs.as<DevSegment>(), b->d_queries.as<DevQuery>(), b->d_tails.as<DevTail>(),
            b->jt, b->d_conj_units.as<uint32_t>(), b->d_conj_item_base.as<uint32_t>(),
            uint32_t(b->conj_units.size()), static_cast<const uint32_t*>(nullptr),
            b->d_conj_lg.as<uint32_t>(), b->d_conj_seek.as<uint32_t>(), b->d_conj_recs.as<ConjItem>());
  if (b->n_conj_pilot) {
    ConjArgs p = a;
    p.wgs = b->d_conj_pilot.as<PhraseWg>();
    p.n_pilot = b->n_conj_pilot;
    RT_LAUNCH((k_conj<LAYOUT>), (b->n_conj_pilot + kConjWaves - 1) / kConjWaves, kConjWaves * 64, 0,
              st, p, 1u);
  }
  RT_LAUNCH(k_conj_threshold, uint32_t(b->conj_units.size()), 64, 0, st,
            b->d_queries.as<DevQuery>(), b->d_conj_units.as<uint32_t>(),
            b->d_conj_items.as<uint32_t>(), b->d_conj_hist.as<uint32_t>(), a.pilot_stride,
            b->estimate ? kPilotMargin : 0u, b->d_bstar.as<uint32_t>(), min_bins(b));
  RT_LAUNCH((k_conj<LAYOUT>), (b->conj_total_items + kConjWaves - 1) / kConjWaves, kConjWaves * 64,
            0, st, a, 0u);
  RT_LAUNCH(k_conj_hits, uint32_t(b->conj_units.size()), 64, 0, st, b->d_conj_units.as<uint32_t>(),
            b->d_conj_item_base.as<uint32_t>(), b->d_conj_item_hits.as<uint32_t>(),
            b->d_hits.as<unsigned long long>());
  return rt::last_error_ok();
}

// Work-item lists of every (unit, doc tile): count -> exclusive scan (all on the device, no
// host round trip: the buffer is sized by an upper bound) -> fill.
bool launch_items(irs_hip_batch* b, rt::stream_t st) {
  uint32_t* off = b->d_tile_off.as<uint32_t>();
  const uint64_t n = uint64_t(b->total_tiles) + 1;   // [total_tiles] = 0 -> the grand total
  if (!rt::dmemset(off + b->total_tiles, 0, 4, st)) return false;
  const uint32_t tb = (b->max_tiles + kThreads - 1) / kThreads;
  RT_LAUNCH(k_items_count, b->nq * tb, kThreads, 0, st, b->d_queries.as<DevQuery>(), b->jt,
            b->tile, tb, b->d_first.as<uint32_t>(), b->d_tails.as<DevTail>(), off);
  const uint32_t parts = uint32_t((n + kScanChunk - 1) / kScanChunk);
  uint64_t* totals = b->d_scan_parts.as<uint64_t>();
  RT_LAUNCH(k_scan_totals, parts, kThreads, 0, st, off, n, totals);
  RT_LAUNCH(k_scan_parts, 1, 64, 0, st, totals, parts);
  RT_LAUNCH(k_scan_apply, parts, kThreads, 0, st, off, n, totals);
  const uint32_t tb4 = (b->max_tiles + kWaves - 1) / kWaves;
  RT_LAUNCH(k_items_fill, b->nq * tb4, kThreads, 0, st, b->d_segs.as<DevSegment>(),
            b->d_queries.as<DevQuery>(), b->d_qterms.as<DevQTerm>(), b->jt, b->tile, tb4,
            b->nw_log2, caches_off(b), b->d_first.as<uint32_t>(), b->d_tails.as<DevTail>(), off,
            b->total_tiles, b->d_items.as<ItemG>(), b->wand ? b->d_tile_ub.as<float>() : nullptr);
  return rt::last_error_ok();
}

// The pilot pass's work list of a block-driven batch (And / by_phrase): lead items
// {phase, phase + P, ...} of every unit in conj_units.
bool ensure_pilot_list(irs_hip_batch* b, uint32_t stride, rt::stream_t st) {
  if (b->conj_pilot_stride == stride) return true;
  std::vector<PhraseWg> pl;
  for (size_t c = 0; c < b->conj_units.size(); ++c) {
    const uint32_t u = b->conj_units[c];
    for (uint32_t it = (u * 7u) % stride; it < b->conj_items[c]; it += stride)
      pl.push_back(PhraseWg{u, it});
  }
  // (the list being replaced may still be read by a run in flight: recoveries come here)
  if ((b->d_conj_pilot.p && !rt::sync(st)) ||
      !b->d_conj_pilot.alloc(std::max<size_t>(1, pl.size()) * sizeof(PhraseWg)) ||
      !b->up.copy(b->d_conj_pilot.p, pl.data(), pl.size() * sizeof(PhraseWg)) || !b->up.flush(st))
    return false;
  b->n_conj_pilot = uint32_t(pl.size());
  b->conj_pilot_stride = stride;
  return true;
}

// by_phrase: lead-item records + start blocks -> pilot pass over every P-th lead block ->
// threshold bins -> full pass.
template<int LAYOUT, int MT>
bool launch_phrase(irs_hip_batch* b, rt::stream_t st) {
  if (b->n_phrase_wgs == 0) return true;  // no query has all its terms in its segment
  const uint32_t stride = b->stride_eff;
  if (!ensure_pilot_list(b, stride, st) || !rt::dmemset(b->d_conj_hist.p, 0, b->d_conj_hist.n, st))
    return false;
  ConjArgs a{};
  a.segs = b->d_segs.as<DevSegment>();
  a.queries = b->d_queries.as<DevQuery>();
  a.qterms = b->d_qterms.as<DevQTerm>();
  a.wgs = nullptr;
  a.n_items = b->conj_total_items;
  a.tails = b->d_tails.as<DevTail>();
  a.bstar = b->d_bstar.as<uint32_t>();
  a.cands = b->d_cands.as<uint64_t>();
  a.cand_count = b->d_cand_count.as<uint32_t>();
  a.hits = b->d_hits.as<unsigned long long>();
  a.hist = b->d_conj_hist.as<uint32_t>();
  a.touched = b->count_touched ? b->d_touched.as<unsigned long long>() : nullptr;
  a.seek = b->d_conj_seek.as<uint32_t>();
  a.recs = b->d_conj_recs.as<ConjItem>();
  a.unit_items = b->d_conj_unit_items.as<uint32_t>();
  a.lead_of = b->d_lead_of.as<uint32_t>();
  a.item_hits = b->d_conj_item_hits.as<uint32_t>();
  a.jt = b->jt;
  a.cand_cap = b->cand_cap;
  a.pilot_stride = stride;
  if (!rt::dmemset(b->d_conj_item_hits.p, 0, b->d_conj_item_hits.n, st)) return false;
  RT_LAUNCH(k_conj_seek, (b->conj_total_items + kThreads - 1) / kThreads, kThreads, 0, st,
            b->d_segs.as<DevSegment>(), b->d_queries.as<DevQuery>(), b->d_tails.as<DevTail>(),
            b->jt, b->d_conj_units.as<uint32_t>(), b->d_conj_item_base.as<uint32_t>(),
            uint32_t(b->conj_units.size()), b->d_lead_of.as<uint32_t>(),
            static_cast<const uint32_t*>(nullptr), b->d_conj_seek.as<uint32_t>(),
            b->d_conj_recs.as<ConjItem>());
  if (b->n_conj_pilot) {
    ConjArgs p = a;
    p.wgs = b->d_conj_pilot.as<PhraseWg>();
    p.n_pilot = b->n_conj_pilot;
    p.touched = nullptr;
    if (MT == 2) {
      RT_LAUNCH(k_phrase2<LAYOUT>, (b->n_conj_pilot + kPhraseWaves - 1) / kPhraseWaves,
                kPhraseWaves * 64, 0, st, p, 1u);
    } else {
      RT_LAUNCH((k_phrase<LAYOUT, MT>), (b->n_conj_pilot + kPhraseWaves - 1) / kPhraseWaves,
                kPhraseWaves * 64, 0, st, p, 1u);
    }
  }
  RT_LAUNCH(k_conj_threshold, uint32_t(b->conj_units.size()), 64, 0, st,
            b->d_queries.as<DevQuery>(), b->d_conj_units.as<uint32_t>(),
            b->d_conj_items.as<uint32_t>(), b->d_conj_hist.as<uint32_t>(), stride,
            b->estimate ? kPilotMargin : 0u, b->d_bstar.as<uint32_t>(), min_bins(b));
  if (MT == 2) {
    RT_LAUNCH(k_phrase2<LAYOUT>, b->n_phrase_wgs, kPhraseWaves * 64, 0, st, a, 0u);
  } else {
    RT_LAUNCH((k_phrase<LAYOUT, MT>), b->n_phrase_wgs, kPhraseWaves * 64, 0, st, a, 0u);
  }
  RT_LAUNCH(k_conj_hits, uint32_t(b->conj_units.size()), 64, 0, st, b->d_conj_units.as<uint32_t>(),
            b->d_conj_item_base.as<uint32_t>(), b->d_conj_item_hits.as<uint32_t>(),
            b->d_hits.as<unsigned long long>());
  return rt::last_error_ok();
}
template<int LAYOUT>
bool launch_phrase_terms(irs_hip_batch* b, rt::stream_t st) {
  if (b->jt <= 2) return launch_phrase<LAYOUT, 2>(b, st);
  if (b->jt <= 4) return launch_phrase<LAYOUT, 4>(b, st);
  return launch_phrase<LAYOUT, int(kPhraseMaxTerms)>(b, st);
}

// Block-max data of a segment (conj.h k_block_max), built once, on first use.
static bool launch_block_max(irs_hip_segment* s) {
  const uint64_t rows = s->total_blocks;
  if (s->dev.layout == kSimd4) {
    RT_LAUNCH((k_block_max<kSimd4>), row_grid(rows, s->cus), kThreads, 0, nullptr, s->dev,
              rows, s->d_blk_maxf.as<uint32_t>(), s->d_blk_minn.as<uint32_t>());
  } else {
    RT_LAUNCH((k_block_max<kScalar>), row_grid(rows, s->cus), kThreads, 0, nullptr, s->dev,
              rows, s->d_blk_maxf.as<uint32_t>(), s->d_blk_minn.as<uint32_t>());
  }
  return rt::last_error_ok() && rt::sync(nullptr);
}

int prepare_posting_norms(irs_hip_segment* s) {
  std::lock_guard<std::mutex> lock(s->wand_mutex);
  if (s->pnorm_ready) return IRS_HIP_OK;
  const DevSegment& d = s->dev;
  if (d.norms && d.norm_width == 1u && !d.norm_legacy) {
    const uint64_t rows = s->total_blocks, tails = s->d_tail_docs.n / 4;
    if (!s->d_pnorm.alloc((rows + 1) * kBlock) || !s->d_tail_norms.alloc(tails + 1))
      return IRS_HIP_ENOMEM;
    if (rows && d.num_terms) {
      if (d.layout == kSimd4) {
        RT_LAUNCH((k_posting_norms<kSimd4>), row_grid(rows, s->cus), kThreads, 0, nullptr, d, rows,
                  s->d_pnorm.as<uint8_t>());
      } else {
        RT_LAUNCH((k_posting_norms<kScalar>), row_grid(rows, s->cus), kThreads, 0, nullptr, d, rows,
                  s->d_pnorm.as<uint8_t>());
      }
    }
    if (tails) {
      RT_LAUNCH(k_tail_norms, uint32_t((tails + kThreads - 1) / kThreads), kThreads, 0, nullptr, d,
                tails, s->d_tail_norms.as<uint8_t>());
    }
    if (!rt::last_error_ok() || !rt::sync(nullptr)) return IRS_HIP_EHIP;
    s->device_bytes += s->d_pnorm.n + s->d_tail_norms.n;
    s->dev.pnorm = s->d_pnorm.as<uint8_t>();
    s->dev.tail_norms = s->d_tail_norms.as<uint8_t>();
  }
  s->pnorm_ready = true;
  return IRS_HIP_OK;
}

int prepare_blockmax(irs_hip_segment* s) {
  std::lock_guard<std::mutex> lock(s->wand_mutex);
  if (s->wand_ready) return IRS_HIP_OK;
  const uint64_t n = s->total_blocks;
  if (!s->d_blk_maxf.alloc((n + 1) * 4) || !s->d_blk_minn.alloc((n + 1) * 4)) return IRS_HIP_ENOMEM;
  if (n && s->dev.num_terms) {
    // derived from the postings: every block of every index gets a pair
    if (!launch_block_max(s)) return IRS_HIP_EHIP;
    // a field indexed with scorers carries the pairs itself (skip level 0): those are used —
    // when they bound EVERY score function: a MaxFreq or MinNorm payload.  A DivNorm payload
    // is the (freq, norm) of the doc with the largest ratio, no bound for BM25 or a MaxFreq
    // scorer (the reference refuses the combination: Scorer::compatible, scorer.cpp:46-49)
    if (!s->skip_at.empty() &&
        (s->wand_type == IRS_HIP_WAND_MAX_FREQ || s->wand_type == IRS_HIP_WAND_MIN_NORM)) {
      DevBuf d_at, d_taken;
      if (!d_at.alloc(s->skip_at.size() * 8) || !d_taken.alloc(8)) return IRS_HIP_ENOMEM;
      uint32_t status = 0;
      unsigned long long taken = 0;
      if (!rt::h2d(d_at.p, s->skip_at.data(), s->skip_at.size() * 8, nullptr) ||
          !rt::dmemset(d_taken.p, 0, 8, nullptr) || !rt::dmemset(s->d_status.p, 0, 4, nullptr))
        return IRS_HIP_EHIP;
      RT_LAUNCH(k_wand_skip0, s->dev.num_terms, kChainThreads, 0, nullptr,
                s->dev, d_at.as<uint64_t>(), s->has_pos ? 1u : 0u, s->d_blk_maxf.as<uint32_t>(),
                s->d_blk_minn.as<uint32_t>(), d_taken.as<unsigned long long>(),
                s->d_status.as<uint32_t>());
      if (!rt::last_error_ok() || !rt::d2h(&status, s->d_status.p, 4, nullptr) ||
          !rt::d2h(&taken, d_taken.p, 8, nullptr) || !rt::sync(nullptr))
        return IRS_HIP_EHIP;
      if (status & kStatusCorrupt) return IRS_HIP_ECORRUPT;
      if (status & kStatusWandFraming) {
        // entries that do not line up with the block directory (e.g. a field with positions
        // opened without its `.pos`): nothing of the walk is trusted, the derived pairs stand
        if (!launch_block_max(s)) return IRS_HIP_EHIP;
        taken = 0;
      }
      s->wand_from_index = taken;
    }
  }
  s->dev.blk_maxf = s->d_blk_maxf.as<uint32_t>();
  s->dev.blk_minn = s->d_blk_minn.as<uint32_t>();
  s->device_bytes += s->d_blk_maxf.n + s->d_blk_minn.n;
  s->wand_ready = true;
  return IRS_HIP_OK;
}

// ---- joined posting streams (join.h) ----------------------------------------------------
// Can the batch's doc-tile units run as joined streams?  (Anything else keeps score.h's work
// items: per-doc match counters, Max / Min merged scores, scorers outside the table family,
// 64-bit accumulators, a frequency that does not fit an entry.)
bool join_allowed(const irs_hip_batch* b) {   // batch level
  if (b->path_pref == IRS_HIP_PATH_ITEMS) return false;
  if (const char* e = std::getenv("IRS_HIP_JOIN")) {   // tuning / test knob
    if (std::atoi(e) == 0 && b->path_pref != IRS_HIP_PATH_JOINED) return false;
  }
  // (a unit on joined streams runs exhaustively under ExecutionContext::wand: the top k is the
  // exhaustive one by construction; pruning stays with the block-driven / work-item kernels)
  return !b->phrase && b->acc32;
}
bool join_counts_allowed() {   // tuning / test knob
  const char* e = std::getenv("IRS_HIP_JOIN_COUNTS");
  return !e || std::atoi(e) != 0;
}
// Conjunction as joined streams or block driven?  Measured on 10 M docs (tools/sweep.py --op and,
// GPU time summed over the chip, picoseconds): joined = 4200 per doc tile of the unit (barriers,
// epilogue: the part that does not depend on the postings) + 1.5 per posting of its terms
// (k_join_score 0.3 + a share of k_join's decode); block driven = 2300 + 400 x terms per
// 128-posting block of the rarest term: its decode plus a seek and a block decode in every other
// term.  The conjunctions of two frequent terms are the ones that join.
// Returns the picoseconds saved by joining (<= 0: block driven is cheaper).
int64_t join_and_saving(const irs_hip_batch* b, const DevQuery& dq) {
  const irs_hip_segment* sg = b->segs[dq.seg];
  uint64_t sum = 0, lead = ~0ull;
  for (uint32_t j = 0; j < dq.n_terms; ++j) {
    const uint64_t df = sg->terms[b->qterms[dq.first_term + j].term].docs_count;
    sum += df;
    lead = std::min(lead, df);
  }
  const uint64_t tiles = sg->dev.num_docs / kJoinTile + 1;
  const uint64_t lead_blocks = lead / kBlock + 1;
  return int64_t(lead_blocks * (2300ull + 400ull * dq.n_terms)) -
         int64_t(4200ull * tiles + (3ull * sum) / 2);
}
// ... and the launches of the joined kernels themselves (k_join, the pilot, one more score
// kernel) only pay when the conjunctions that would join save more than that together
constexpr int64_t kJoinAndLaunchCost = 500000000;   // 0.5 ms
int join_and_forced(const irs_hip_batch* b) {   // -1: decide by cost
  if (b->path_pref == IRS_HIP_PATH_JOINED) return 1;   // (forced: wherever it is possible)
  if (const char* e = std::getenv("IRS_HIP_JOIN_AND")) return std::atoi(e) != 0;   // tuning / test knob
  return -1;
}
// Plain disjunctions as joined streams or as work items?  Measured on one MI355X, BM25, 10 M docs
// (tools/cost_sweep.py, profiles/r04_sweeps.txt; picoseconds of step time):
//   joined:     2.9 per posting of every DISTINCT stream (k_join: decode + 4 B written)
//             + 0.47 per posting a query references + 3800 per (unit, doc tile)
//   work items: 1.14 per referenced posting + 6200 per (unit, doc tile)
// A stream pays for itself when it is shared (the headline batch: 5.7 G referenced postings on
// 0.31 G distinct ones) or when there are many units (the per-tile cost is lower): joining wins
// iff  2.9 D < 0.67 R + 2400 T.  128 queries x 8 terms without one shared term: 1.51 ms as work
// items against 1.60 joined; on a corpus of 1000-word docs 0.75 against 1.35.
bool join_or_pays(const irs_hip_batch* b, const std::vector<uint32_t>& units) {
  if (units.empty()) return false;
  uint64_t refs = 0, distinct = 0, tiles = 0;
  std::vector<std::vector<uint8_t>> seen(b->segs.size());
  for (uint32_t u : units) {
    const DevQuery& dq = b->queries[u];
    const irs_hip_segment* sg = b->segs[dq.seg];
    tiles += sg->dev.num_docs / kJoinTile + 1;
    if (seen[dq.seg].empty()) seen[dq.seg].assign(sg->dev.num_terms, 0);
    for (uint32_t j = 0; j < dq.n_terms; ++j) {
      const uint32_t term = b->qterms[dq.first_term + j].term;
      const uint64_t df = sg->terms[term].docs_count;
      refs += df;
      if (!seen[dq.seg][term]) {
        seen[dq.seg][term] = 1;
        distinct += df;
      }
    }
  }
  return 29ull * distinct < (67ull * refs) / 10ull + 24000ull * tiles;
}
int join_or_forced(const irs_hip_batch* b) {   // -1: decide by cost
  if (b->path_pref == IRS_HIP_PATH_JOINED) return 1;
  if (const char* e = std::getenv("IRS_HIP_JOIN_OR")) return std::atoi(e) != 0;   // tuning / test knob
  return -1;
}
bool unit_counts_matches(const DevQuery& dq) {   // min-match / the kMin disjunction of two
  return (dq.op & 0xFF) == 1 || query_min_both(dq.op);
}
bool unit_joinable(const irs_hip_batch* b, uint32_t u) {
  const DevQuery& dq = b->queries[u];
  if (query_min_both(dq.op) || query_merge(dq.op) != kScoreSum) return false;
  if ((dq.op & 0xFF) != 0) {
    // min-match / conjunction: the match count rides in the accumulator's low bits (join.h
    // COUNT) where that costs no precision that matters
    if (!dq.n_terms || !b->count_precise[u] || !join_counts_allowed()) return false;
  }
  const irs_hip_segment* sg = b->segs[dq.seg];
  for (uint32_t j = 0; j < dq.n_terms; ++j) {
    const DevQTerm& qt = b->qterms[dq.first_term + j];
    if (!table_kind(qt.kind) || qt.cache_id >= kMaxCaches) return false;
    if (sg->terms[qt.term].tf_bound > kJoinTfMax) return false;
  }
  return true;
}

// The batch's distinct (segment, term) streams, k_join's work list and the per-(unit, term)
// records of k_join_score.  Static per batch: built once, the kernels refill the entries and
// boundaries in every run.
bool build_streams(irs_hip_batch* b) {
  struct WgRef { uint32_t stream, first; };   // a k_join workgroup before its record is made
  std::unique_ptr<HostTrace> tr(new HostTrace("  streams: distinct terms"));
  auto lap = [&](const char* what) { tr.reset(); tr.reset(new HostTrace(what)); };
  std::vector<StreamRec> streams;
  std::vector<WgRef> wgs;
  std::vector<JoinTerm> jterms(b->qterms.size());
  // A stream = a distinct (segment, term, scorer signature) of the joined units: the signature
  // — (kind, norm_const, norm_length) — is normally ONE per batch.  stream_of[unit term] by an
  // open-addressing table: the streams come out in first-use order.
  struct Sig { int32_t kind; float nc, nl; };
  std::vector<Sig> sigs;
  std::vector<uint32_t> stream_of(b->qterms.size(), 0xFFFFFFFFu);
  std::vector<uint8_t> stream_sig;
  {
    size_t slots = 64;
    size_t n_keys = 0;
    for (uint32_t u : b->join_units) n_keys += b->queries[u].n_terms;
    while (slots < 2 * n_keys + 2) slots <<= 1;
    std::vector<uint64_t> hkey(slots, ~0ull);
    std::vector<uint32_t> hval(slots, 0);
    for (uint32_t u : b->join_units) {
      const DevQuery& dq = b->queries[u];
      for (uint32_t j = 0; j < dq.n_terms; ++j) {
        const DevQTerm& qt = b->qterms[dq.first_term + j];
        uint32_t sg_id = 0;
        for (; sg_id < sigs.size(); ++sg_id)
          if (sigs[sg_id].kind == qt.kind && sigs[sg_id].nc == qt.norm_const && sigs[sg_id].nl == qt.norm_length) break;
        if (sg_id == sigs.size()) {
          if (sigs.size() >= 255) return false;
          sigs.push_back(Sig{qt.kind, qt.norm_const, qt.norm_length});
        }
        if (dq.seg >= (1u << 24)) return false;
        const uint64_t key = (uint64_t(sg_id) << 56) | (uint64_t(dq.seg) << 32) | qt.term;
        size_t h = size_t((key * 0x9E3779B97F4A7C15ull) >> 32) & (slots - 1);
        while (hkey[h] != ~0ull && hkey[h] != key) h = (h + 1) & (slots - 1);
        if (hkey[h] == ~0ull) {
          hkey[h] = key;
          hval[h] = uint32_t(streams.size());
          StreamRec r{};
          r.seg = dq.seg;
          r.term = qt.term;
          r.n = b->segs[dq.seg]->terms[qt.term].docs_count;
          streams.push_back(r);
          stream_sig.push_back(uint8_t(sg_id));
        }
        stream_of[dq.first_term + j] = hval[h];
      }
    }
  }
  uint64_t entries = 0, bounds = 0;
  std::vector<uint64_t> ent_off, bnd_off;
  for (size_t si = 0; si < streams.size(); ++si) {
    const irs_hip_segment* sg = b->segs[streams[si].seg];
    const DevTerm& t = sg->terms[streams[si].term];
    ent_off.push_back(entries);
    bnd_off.push_back(bounds);
    const uint32_t n_tiles = (sg->dev.num_docs + kJoinTile - 1) / kJoinTile;
    const uint32_t nb = t.nblk + ((t.docs_count == 1 || t.tail_n) ? 1u : 0u);
    for (uint32_t first = 0; first < nb; first += kJoinBlocks)
      wgs.push_back(WgRef{uint32_t(si), first});
    entries += t.docs_count;
    bounds += uint64_t(n_tiles) + 1;
  }
  if (wgs.size() > 0x7FFFFFFFull) return false;
  lap("  streams: work list order");
  // k_join reads a norm byte per posting: launched term after term, the workgroups in flight
  // would touch the whole norm column at once (10 MB at 10 M docs against 4 MB of L2 per XCD).
  // Ordered by where in the doc space a workgroup's blocks lie — estimated as its position
  // inside its list — the ones in flight share a doc range, i.e. norm cache lines.
  {
    // (a counting sort over 1024 positions per segment: this runs once per batch on the host,
    // in front of the batch's first kernel)
    constexpr uint32_t kPos = 1024;
    std::vector<uint32_t> key(wgs.size()), start(b->segs.size() * kPos + 1, 0);
    for (size_t i = 0; i < wgs.size(); ++i) {
      const StreamRec& sr = streams[wgs[i].stream];
      const DevTerm& t = b->segs[sr.seg]->terms[sr.term];
      const uint32_t nb = t.nblk + ((t.docs_count == 1 || t.tail_n) ? 1u : 0u);
      const uint64_t at = (uint64_t(2u * wgs[i].first + kJoinBlocks) * kPos) / (2ull * (nb + kJoinBlocks));
      key[i] = sr.seg * kPos + uint32_t(std::min<uint64_t>(at, kPos - 1));
      ++start[key[i] + 1];
    }
    for (size_t k = 1; k < start.size(); ++k) start[k] += start[k - 1];
    std::vector<WgRef> sorted(wgs.size());
    for (size_t i = 0; i < wgs.size(); ++i) sorted[start[key[i]]++] = wgs[i];
    wgs.swap(sorted);
  }
  lap("  streams: buffers");
  if (!b->d_entries.alloc((entries + kJoinSlack) * 4) || !b->d_bounds.alloc((bounds + 1) * 4) ||
      !b->d_streams.alloc(std::max<size_t>(1, streams.size()) * sizeof(StreamRec)) ||
      !b->d_join_wgs.alloc(std::max<size_t>(1, wgs.size()) * sizeof(JoinWg)) ||
      !b->d_jterms.alloc(jterms.size() * sizeof(JoinTerm)) ||
      !b->d_join_args.alloc(2 * sizeof(JoinArgs)) ||
      !b->d_join_units.alloc(b->join_units.size() * 4) ||
      !b->d_join_order.alloc(b->join_units.size() * 4))
    return false;
  // k_join_score's queues (JoinArgs): per launch — the plain disjunctions, then the units with
  // match counts — the units sorted by (segment, heaviest term) and cut into kJoinQueues runs of
  // about equal work, one queue per XCD: the workgroups that share an L2 work on units that share
  // their longest stream (and the same doc range: chunk-major within a queue).
  lap("  streams: queue order");
  std::vector<uint32_t> order;
  {
    struct Item { uint64_t work; uint64_t key; uint32_t unit; };
    b->n_join_plain = 0;
    for (uint32_t part = 0; part < 2; ++part) {
      std::vector<Item> items;
      for (uint32_t u : b->join_units) {
        const DevQuery& dq = b->queries[u];
        if ((query_need(dq.op) > 1u) != (part == 1u)) continue;
        uint64_t w = 0, top = 0, top_term = 0;
        for (uint32_t j = 0; j < dq.n_terms; ++j) {
          const uint32_t term = b->qterms[dq.first_term + j].term;
          const uint64_t df = b->segs[dq.seg]->terms[term].docs_count;
          w += df;
          if (df > top) { top = df; top_term = term; }
        }
        items.push_back({w, (uint64_t(dq.seg) << 32) | top_term, u});
      }
      if (part == 0) b->n_join_plain = uint32_t(items.size());
      std::stable_sort(items.begin(), items.end(),
                       [](const Item& x, const Item& y) { return x.key < y.key; });
      uint64_t total = 0;
      for (const Item& it : items) total += it.work + 1;
      uint32_t (&first)[kJoinQueues + 1] = b->join_first[part];
      size_t at = 0;
      uint64_t done = 0;
      for (uint32_t g = 0; g < kJoinQueues; ++g) {
        first[g] = uint32_t(order.size());
        const uint64_t goal = total * (g + 1) / kJoinQueues;
        const size_t from = at;
        while (at < items.size() && (done < goal || g + 1 == kJoinQueues)) done += items[at++].work + 1;
        // (round 6: the units of a queue by decreasing work instead — longest processing time first
        // within every chunk round — changes nothing: 0.956 ms either way for a 1.25 M-doc share)
        for (size_t i = from; i < at; ++i) order.push_back(items[i].unit);
      }
      first[kJoinQueues] = uint32_t(order.size());
    }
  }
  for (size_t i = 0; i < streams.size(); ++i) {
    streams[i].entries = reinterpret_cast<uint64_t>(b->d_entries.as<uint32_t>() + ent_off[i]);
    streams[i].bounds = reinterpret_cast<uint64_t>(b->d_bounds.as<uint32_t>() + bnd_off[i]);
    streams[i].n_tiles = (b->segs[streams[i].seg]->dev.num_docs + kJoinTile - 1) / kJoinTile;
    const Sig& sig = sigs[stream_sig[i]];
    streams[i].kind = sig.kind;
    streams[i].nc = sig.nc;
    streams[i].nl = sig.nl;
  }
  for (irs_hip_segment* sg : b->segs)
    if (prepare_posting_norms(sg) != IRS_HIP_OK) return false;
  lap("  streams: k_join records");
  // the workgroups' records (JoinWg: everything k_join reads before its first payload byte)
  JoinWg* wg_recs = static_cast<JoinWg*>(b->up.put(b->d_join_wgs.p, wgs.size() * sizeof(JoinWg)));
  if (!wg_recs && !wgs.empty()) return false;
  for (size_t i = 0; i < wgs.size(); ++i) {
    const StreamRec& sr = streams[wgs[i].stream];
    const irs_hip_segment* sg = b->segs[sr.seg];
    const DevSegment& ds = sg->dev;
    const DevTerm& t = sg->terms[sr.term];
    JoinWg& w = wg_recs[i];
    w.entries = sr.entries;
    w.bounds = sr.bounds;
    w.doc = reinterpret_cast<uint64_t>(ds.doc) + t.doc_start;
    w.dir = reinterpret_cast<uint64_t>(ds.blk_dir + t.dir_off);
    const bool tiny = sg->d_pnorm.p != nullptr;
    w.pnorm = tiny ? reinterpret_cast<uint64_t>(sg->d_pnorm.as<uint8_t>() + t.dir_off * kBlock) : 0ull;
    w.tail_norms = tiny ? reinterpret_cast<uint64_t>(sg->d_tail_norms.as<uint8_t>() + t.tail_row) : 0ull;
    w.tail_docs = reinterpret_cast<uint64_t>(ds.tail_docs + t.tail_row);
    w.tail_freqs = reinterpret_cast<uint64_t>(ds.tail_freqs + t.tail_row);
    w.first = wgs[i].first;
    w.nblk = t.nblk;
    w.tail_n = t.docs_count == 1u ? 1u : t.tail_n;
    w.tail_base = t.nblk ? t.tail_base : 0u;
    w.last_doc = t.last_doc;
    w.n_tiles = (ds.num_docs + kJoinTile - 1) / kJoinTile;
    w.n = sr.n;
    w.dead_lo = uint32_t(reinterpret_cast<uint64_t>(ds.dead));
    w.dead_hi = uint32_t(reinterpret_cast<uint64_t>(ds.dead) >> 32);
    w.pad = 0;
    w.pk = reinterpret_cast<uint64_t>(ds.pk);
    w.pad2 = 0;
  }
  lap("  streams: per-term records");
  for (uint32_t u : b->join_units) {
    DevQuery& dq = b->queries[u];
    const uint32_t rows = table_rows(dq.n_caches);
    for (uint32_t j = 0; j < dq.n_terms; ++j) {
      const DevQTerm& qt = b->qterms[dq.first_term + j];
      const size_t sid = stream_of[dq.first_term + j];
      JoinTerm& jt = jterms[dq.first_term + j];
      jt.pad[0] = jt.pad[1] = 0;
      jt.entries = streams[sid].entries;
      jt.bounds = streams[sid].bounds;
      jt.cs = qt.c0 * dq.fx_mul;
      // (the form only matters for a term with frequencies beyond the table's rows: a TF-IDF
      // batch whose terms all fit the tables runs the table-only loop like a BM25 one)
      const bool general = qt.pad1 >= rows;
      jt.mode = (qt.cache_id * rows * 1024u) |
                (general ? kJoinGeneral | (sqrt_kind(qt.kind) ? kJoinSqrt : 0u) : 0u);
    }
  }
  // (the slack behind the last stream is only ever read by masked-off look-ahead: zero it once)
  b->slack_zeroed = false;   // (run_impl zeroes it on the run's stream)
  if (!b->up.copy(b->d_streams.p, streams.data(), streams.size() * sizeof(StreamRec)) ||
      !b->up.copy(b->d_jterms.p, jterms.data(), jterms.size() * sizeof(JoinTerm)) ||
      !b->up.copy(b->d_join_units.p, b->join_units.data(), b->join_units.size() * 4) ||
      !b->up.copy(b->d_join_order.p, order.data(), order.size() * 4))
    return false;
  b->n_streams = uint32_t(streams.size());
  b->n_join_wgs = uint32_t(wgs.size());
  b->join_entries = entries;
  return true;
}

bool launch_join(irs_hip_batch* b, rt::stream_t st) {
  if (!b->n_join_wgs) return true;
  if (b->seg->dev.layout == kSimd4) {
    RT_LAUNCH((k_join<kSimd4>), b->n_join_wgs, kThreads, 0, st, b->d_join_wgs.as<JoinWg>());
  } else {
    RT_LAUNCH((k_join<kScalar>), b->n_join_wgs, kThreads, 0, st, b->d_join_wgs.as<JoinWg>());
  }
  return rt::last_error_ok();
}

// Groups of a batch over several segments (irs_hip_batch_set_shared_threshold): the units of one
// query — where every one of them runs on joined streams and they bin scores alike (the bins
// span [0, upper bound of the query's score]: equal for scorers whose bound does not depend on the
// segment's frequencies).  Anything else keeps a threshold per unit.
bool build_groups(irs_hip_batch* b) {
  b->n_groups = 0;
  const uint32_t n_segs = uint32_t(b->segs.size());
  const bool across = b->comm != nullptr && !b->phrase;
  if (!across && (!b->shared_threshold || n_segs < 2 || n_segs > 64 || b->join_units.empty())) return true;
  const uint32_t nq_user = b->nq_user;
  std::vector<uint8_t> is_join(b->nq, 0);
  for (uint32_t u : b->join_units) is_join[u] = 1;
  std::vector<uint32_t> group_of(b->nq, 0), members(size_t(nq_user) * n_segs, 0xFFFFFFFFu);
  uint32_t grouped = 0;
  for (uint32_t g = 0; g < nq_user && n_segs <= 64; ++g) {
    bool ok = true;
    uint32_t live = 0;
    for (uint32_t sgi = 0; sgi < n_segs && ok; ++sgi) {
      const uint32_t u = sgi * nq_user + g;
      const DevQuery& dq = b->queries[u];
      if (!dq.n_terms) continue;   // (nothing of the query in this segment)
      ok = is_join[u] != 0;
      if (across) {
        // the other ranks' units cannot be asked: only a bound that every segment of the index
        // arrives at by itself qualifies (the boosts of ALL the query's terms, present or not)
        ok = ok && b->group_upper[u] > 0.0;
      } else {
        for (uint32_t s2 = 0; s2 < sgi && ok; ++s2) {
          const DevQuery& other = b->queries[s2 * nq_user + g];
          if (other.n_terms) ok = other.bin_scale == dq.bin_scale && other.k == dq.k;
        }
      }
      ++live;
    }
    if (!ok || live < (across ? 1u : 2u)) continue;
    for (uint32_t sgi = 0; sgi < n_segs; ++sgi) {
      const uint32_t u = sgi * nq_user + g;
      if (!b->queries[u].n_terms) continue;
      if (across) b->queries[u].bin_scale = float(double(kBins) / b->group_upper[u]);
      group_of[u] = g + 1;
      members[size_t(g) * n_segs + sgi] = u;
    }
    ++grouped;
  }
  // (across ranks the collectives run whatever this rank's own units look like)
  if (!grouped && !across) return true;
  if (!b->d_group_of.alloc(group_of.size() * 4) || !b->d_group_members.alloc(members.size() * 4) ||
      !b->d_group_hist.alloc(uint64_t(nq_user) * (kBins + 2) * 4) ||
      !b->d_group_sums.alloc((uint64_t(nq_user) * kGroupSumWords + 2) * 4) ||
      !b->up.copy(b->d_group_of.p, group_of.data(), group_of.size() * 4) ||
      !b->up.copy(b->d_group_members.p, members.data(), members.size() * 4))
    return false;
  b->n_groups = nq_user;
  return true;
}

bool launch_join_pilot(irs_hip_batch* b, rt::stream_t st) {
  const size_t smem = JoinOff::end + kBins * sizeof(uint32_t);
  if (!big_smem(k_join_pilot, smem)) return false;
  RT_LAUNCH(k_join_pilot, uint32_t(b->join_units.size()), b->join_threads, smem, st,
            b->d_join_units.as<uint32_t>(), b->d_queries.as<DevQuery>(),
            b->d_qterms.as<DevQTerm>(), b->d_jterms.as<JoinTerm>(), b->stride_eff,
            b->join_nw_log2, b->d_bstar.as<uint32_t>(), b->estimate ? kPilotMargin : 0u,
            min_bins(b), b->n_groups ? b->d_group_of.as<uint32_t>() : nullptr,
            b->d_group_hist.as<uint32_t>());
  return rt::last_error_ok();
}

// One threshold per group from the summed pilot histograms — summed over the ranks first when the
// batch has a communicator: the units of a query on ALL segments of the index then admit together
// what one heap over all segments would (index-search.cpp:719-779).
bool launch_group_threshold(irs_hip_batch* b, rt::stream_t st) {
  if (!b->n_groups) return true;
  if (b->comm && !b->phrase &&
      !rt::comm::all_reduce_u32(b->comm->h, b->d_group_hist.p, size_t(b->n_groups) * (kBins + 2), st))
    return false;
  RT_LAUNCH(k_group_threshold, b->n_groups, 64, 0, st, b->d_queries.as<DevQuery>(),
            b->d_group_members.as<uint32_t>(), uint32_t(b->segs.size()),
            b->d_group_hist.as<uint32_t>(), b->estimate ? kPilotMargin : 0u, min_bins(b),
            b->d_bstar.as<uint32_t>());
  return rt::last_error_ok();
}

// Paired tiles (join.h join_pairs) for the launch of the plain disjunctions: no segment of theirs
// has deleted docs (their entries leave the doc order k_join_rescore searches in).
// IRS_HIP_JOIN_HALF=0 / irs_hip_batch_set_paired_tiles(0) keeps the 32-bit tiles (A/B runs, tests:
// the two must agree bit for bit).
bool join_half_ok(const irs_hip_batch* b) {
  if (const char* e = std::getenv("IRS_HIP_JOIN_HALF")) {
    if (std::atoi(e) == 0) return false;
  }
  if (!b->pairs_allowed || !b->acc32 || !b->n_join_plain) return false;
  // Where it pays (measured on one MI355X, GPU time summed over the chip): a (unit, doc tile)
  // visited in a pair saves ~1.4 ns — half of that when the batch is too small to keep the chip
  // busy through the tail of the work queue (fewer than ~400 k visits) —, a look-up of
  // k_join_rescore costs ~40 ps, and a unit looks up about min(3 k / G, k) docs in each of its
  // terms (G: the units that share its threshold — the segments of a batch with a shared
  // threshold times the ranks of its communicator; 3 = kPilotMargin).  10 M docs in one segment,
  // k = 1000: 1.14 us saved against 0.35 per unit (5.65 -> 4.74 ms per 1000 units); a 1.25 M-doc
  // share of it alone: 0.07 against 0.2 (stays on 32-bit tiles: 0.97 against 1.10 ms); the
  // same as 8 segments of one batch: 0.14 against 0.08 (5.79 -> 5.27 ms).
  // IRS_HIP_JOIN_HALF=1 / set_paired_tiles(2) pair whatever the size (tests on small segments).
  bool forced = b->pairs_forced;
  if (const char* e = std::getenv("IRS_HIP_JOIN_HALF")) forced = forced || std::atoi(e) == 1;
  uint64_t visits = 0, lookups = 0;
  for (uint32_t u : b->join_units) {
    const DevQuery& dq = b->queries[u];
    if (query_need(dq.op) > 1u) continue;
    if (b->segs[dq.seg]->dev.dead) return false;
    const uint64_t group = uint64_t((b->shared_threshold || b->comm) ? b->segs.size() : 1) *
                           uint64_t(b->comm ? std::max(1, b->comm->n_ranks) : 1);
    visits += b->segs[dq.seg]->dev.num_docs / kJoinTile + 1;
    lookups += std::min<uint64_t>((uint64_t(kPilotMargin) * dq.k + group - 1) / group, uint64_t(dq.k) + 64) *
               dq.n_terms;
  }
  if (!forced && (visits >= 400000 ? 1400ull : 700ull) * visits <= 40ull * lookups) return false;
  return true;
}

bool launch_join_score(irs_hip_batch* b, rt::stream_t st) {
  const size_t smem = JoinOff::end;
  if (!big_smem(k_join_score<kJKPlain>, smem) || !big_smem(k_join_score<kJKCount>, smem) ||
      !big_smem(k_join_score<kJKHalf>, smem))
    return false;
  const bool half = join_half_ok(b);
  b->pairs_used = half;
  const uint32_t waves = b->join_threads / 64;
  uint32_t per_cu = uint32_t((160u * 1024u) / smem);
  per_cu = std::max<uint32_t>(1, std::min<uint32_t>(per_cu, 32u / waves));
  // chunks of up to kJoinChunkTiles tiles, the unit's tiles cut evenly (102 tiles: 4 x 26, not
  // 3 x 32 + 6 — a short last chunk pays the whole per-chunk prologue for a few tiles).  A small
  // batch takes shorter chunks: with fewer than ~20 chunks per resident workgroup the last round
  // of the work queue leaves CUs idle (1000 units x 102 tiles: 8 chunks per workgroup at 26 tiles,
  // 1.12 ms; 21 at 10 tiles, 0.95 ms — profiles/r05_chunks.txt), while a large batch loses to the
  // per-chunk prologue below 26 (10 M docs: 5.61 ms at 32, 5.88 at 16, 6.50 at 8).
  // (paired tiles: up to kJoinChunkTiles = 64 tiles = 32 visits per chunk, 4.98 -> 4.91 ms)
  auto chunking = [&](uint32_t cap, uint32_t& cpq, uint32_t& chunk_tiles) {
    const uint64_t tiles = uint64_t(b->join_units.size()) * b->join_max_tiles;
    const uint64_t wgs = uint64_t(b->seg->cus) * per_cu;
    uint32_t max_chunk = uint32_t(std::min<uint64_t>(cap, std::max<uint64_t>(8, tiles / (20 * wgs))));
    if (const char* e = std::getenv("IRS_HIP_JOIN_CHUNK")) {   // tuning knob: tiles per chunk at most
      const uint32_t v = uint32_t(std::atoi(e));
      if (v >= 1 && v <= cap) max_chunk = v;
    }
    cpq = std::max<uint32_t>(1, (b->join_max_tiles + max_chunk - 1) / max_chunk);
    chunk_tiles = std::max<uint32_t>(1, (b->join_max_tiles + cpq - 1) / cpq);
  };
  const uint32_t n_all = uint32_t(b->join_units.size());
  // [0]: the live counters, [1]: their start values (copied over [0] on the device every run)
  if (!b->d_join_ctr.p && !b->d_join_ctr.alloc(2 * sizeof b->join_ctr_init)) return false;
  // two launches: the plain disjunctions, then the units whose accumulators count matches
  for (uint32_t part = 0; part < 2; ++part) {
    const uint32_t n_units = part ? n_all - b->n_join_plain : b->n_join_plain;
    if (!n_units) continue;
    uint32_t cpq = 1, chunk_tiles = 1;
    chunking((!part && half) ? kJoinChunkTiles : kJoinChunkPlain, cpq, chunk_tiles);
    const uint64_t chunks = uint64_t(n_units) * cpq;
    if (chunks > 0xFFFF0000ull) return false;
    const uint32_t grid = uint32_t(std::min<uint64_t>(chunks, uint64_t(b->seg->cus) * per_cu));
    JoinArgs& a = b->join_args[part];   // read by the kernel from device memory
    a.queries = b->d_queries.as<DevQuery>();
    a.qterms = b->d_qterms.as<DevQTerm>();
    a.jterms = b->d_jterms.as<JoinTerm>();
    a.bstar = b->d_bstar.as<uint32_t>();
    a.cands = b->d_cands.as<uint64_t>();
    a.cand_count = b->d_cand_count.as<uint32_t>();
    a.hits = b->d_hits.as<unsigned long long>();
    a.order = b->d_join_order.as<uint32_t>();
    a.work_counter = b->d_join_ctr.as<uint32_t>() + part * kJoinQueues;
    uint32_t base = 0;
    for (uint32_t g = 0; g <= kJoinQueues; ++g) {
      a.first[g] = b->join_first[part][g];
      a.base[g] = base;
      if (g < kJoinQueues) {
        b->join_ctr_init[part][g] = base;
        base += (b->join_first[part][g + 1] - b->join_first[part][g]) * cpq;
      }
    }
    a.cpq = cpq;
    a.n_units = n_units;
    a.nw_log2 = b->join_nw_log2;
    if (const char* e = std::getenv("IRS_HIP_JOIN_SPLIT_LOG2")) {   // tuning knob: a tile's entries
      const uint32_t v = uint32_t(std::atoi(e));                   // among the first 2^v wavefronts only
      if (v < a.nw_log2) a.nw_log2 = v;
    }
    a.cand_cap = b->cand_cap;
    a.chunk_tiles = chunk_tiles;
    JoinArgs* d_args = b->d_join_args.as<JoinArgs>() + part;
    uint32_t* d_init = a.work_counter + 2 * kJoinQueues;
    if (!b->join_args_valid[part] || std::memcmp(&a, &b->join_args_sent[part], sizeof a) != 0) {
      if (!b->up.copy(d_args, &a, sizeof a) ||
          !b->up.copy(d_init, b->join_ctr_init[part], sizeof b->join_ctr_init[part]) ||
          !b->up.flush(st))
        return false;
      std::memcpy(&b->join_args_sent[part], &a, sizeof a);
      b->join_args_valid[part] = true;
    }
    if (!rt::d2d(a.work_counter, d_init, sizeof b->join_ctr_init[part], st)) return false;
    if (part) {
      RT_LAUNCH(k_join_score<kJKCount>, grid, b->join_threads, smem, st, d_args);
    } else if (half) {
      // (d_join_order: the plain units first — one k_join_rescore workgroup each)
      RT_LAUNCH(k_join_score<kJKHalf>, grid, b->join_threads, smem, st, d_args);
      RT_LAUNCH(k_join_rescore, n_units, kRescoreThreads, 0, st, b->d_join_order.as<uint32_t>(),
                b->d_queries.as<DevQuery>(), b->d_qterms.as<DevQTerm>(), b->d_jterms.as<JoinTerm>(),
                b->d_bstar.as<uint32_t>(), b->d_cands.as<uint64_t>(), b->d_cand_count.as<uint32_t>(),
                b->cand_cap);
    } else {
      RT_LAUNCH(k_join_score<kJKPlain>, grid, b->join_threads, smem, st, d_args);
    }
  }
  return rt::last_error_ok();
}


// k_conj work of the batch's block-driven conjunctions (conj_units): the lead term of a unit is
// its first one (sorted by cost at create); one wavefront per 128-posting block of it (+ one for
// its vint tail / single doc), its record and the other terms' start blocks written by
// k_conj_seek every run.  Rebuilt whenever ensure_scratch deals the conjunctions anew.
int build_conj_work(irs_hip_batch* b) {
  const uint32_t nq = b->nq;
  int rc = IRS_HIP_OK;
  b->conj_items.clear();
  b->conj_total_items = 0;
  b->n_conj_wgs = 0;
  b->n_conj_pilot = 0;
  b->conj_pilot_stride = 0;
  if (b->conj_units.empty()) return rc;
  try {
    // A lead block whose 128 docs fall into many blocks of the other terms — a rare lead against
    // frequent terms — is one wavefront decoding those blocks one after the other: it is cut into
    // 2^lg pieces, a wavefront each (ConjItem).  A lead doc falls into at most one block per term,
    // and a term has df_j / df_lead blocks per lead doc: W = sum_j min(128, df_j / df_lead) blocks
    // per lead block; pieces of about 16.  (Measured on the reference's AndHighLow class — lead df
    // ~280 against 720 k: 0.98 -> 0.11 ms per 256 queries.)  Only while the batch cannot fill the
    // chip anyway: with more lead blocks than wavefront slots every wavefront's chain hides behind
    // the others', and the pieces' repeated lead decodes and shared border blocks only add work
    // (config 5's AND batch: 13.4 -> 15.7 ms with the cut applied to every unit).
    uint64_t lead_items = 0;
    for (uint32_t u : b->conj_units) {
      const DevQuery& dq = b->queries[u];
      if (dq.n_terms) lead_items += b->segs[dq.seg]->terms[b->qterms[dq.first_term].term].nblk + 1u;
    }
    const bool roomy = lead_items < 2ull * 32ull * b->seg->cus;   // (8 wavefronts per SIMD)
    int forced_lg = -1;
    if (const char* e = std::getenv("IRS_HIP_CONJ_SPLIT_LOG2")) {   // tuning / test knob
      forced_lg = std::atoi(e);
      if (forced_lg < 0 || forced_lg > int(kConjSplitMax)) forced_lg = -1;
    }
    std::vector<uint32_t> split_lg;
    for (uint32_t u : b->conj_units) {
      const DevQuery& dq = b->queries[u];
      uint32_t items = 0, lg = 0;
      if (dq.n_terms) {
        const irs_hip_segment* sg = b->segs[dq.seg];
        const DevTerm& t = sg->terms[b->qterms[dq.first_term].term];
        items = t.nblk + ((t.docs_count == 1 || t.tail_n) ? 1u : 0u);
        uint64_t want = 0;
        for (uint32_t j = 1; j < dq.n_terms; ++j) {
          const uint64_t df = sg->terms[b->qterms[dq.first_term + j].term].docs_count;
          want += std::min<uint64_t>(kBlock, df / std::max<uint32_t>(1u, t.docs_count));
        }
        while (roomy && lg < kConjSplitMax && (want >> lg) > 16u) ++lg;
        if (forced_lg >= 0) lg = uint32_t(forced_lg);
      }
      split_lg.push_back(lg);
      b->conj_items.push_back(items << lg);
    }
    // rows of the seek table: the lead items of the conj units, unit after unit
    std::vector<uint32_t> item_base(b->conj_units.size() + 1, 0), unit_items(nq, 0);
    uint64_t total = 0;
    for (size_t c = 0; c < b->conj_units.size(); ++c) {
      item_base[c] = uint32_t(total);
      unit_items[b->conj_units[c]] = uint32_t(total);
      total += b->conj_items[c];
    }
    if (total > 0x7FFFFFFFull) return IRS_HIP_EUNSUPPORTED;
    item_base[b->conj_units.size()] = uint32_t(total);
    b->conj_total_items = uint32_t(total);
    b->n_conj_wgs = uint32_t((total + kConjWaves - 1) / kConjWaves);
    if (!b->d_conj_item_base.alloc(item_base.size() * 4) ||
        !b->d_conj_unit_items.alloc(unit_items.size() * 4) ||
        !b->d_conj_seek.alloc((total + 2) * uint64_t(kMaxTerms) * 4) ||
        !b->d_conj_recs.alloc((total + 1) * sizeof(ConjItem)) ||
        !b->d_conj_item_hits.alloc((total + 1) * 4) ||
        !b->d_conj_units.alloc(b->conj_units.size() * 4) ||
        !b->d_conj_items.alloc(b->conj_items.size() * 4) ||
        !b->d_conj_lg.alloc(split_lg.size() * 4) ||
        !b->d_conj_hist.alloc(uint64_t(nq) * kBins * 4))
      return IRS_HIP_ENOMEM;
    if (!b->up.copy(b->d_conj_item_base.p, item_base.data(), item_base.size() * 4) ||
        !b->up.copy(b->d_conj_unit_items.p, unit_items.data(), unit_items.size() * 4) ||
        !b->up.copy(b->d_conj_units.p, b->conj_units.data(), b->conj_units.size() * 4) ||
        !b->up.copy(b->d_conj_items.p, b->conj_items.data(), b->conj_items.size() * 4) ||
        !b->up.copy(b->d_conj_lg.p, split_lg.data(), split_lg.size() * 4))
      return IRS_HIP_ENOMEM;
  } catch (...) {
    rc = IRS_HIP_ENOMEM;
  }
  return rc;
}

bool ensure_scratch(irs_hip_batch* b) {
  if (b->scratch_ready) return true;
  HostTrace trace("ensure_scratch (units dealt, streams, work lists)");
  b->join_args_valid[0] = b->join_args_valid[1] = b->score_args_valid = false;
  b->min_dirty = b->has_min;   // (a unit's bin_scale may change below: build_groups)
  if (b->phrase) b->tile = 0x40000000u;  // k_phrase is block driven: one "tile" = the segment
  // 32-bit accumulators halve the LDS per doc: twice the tile at the same residency
  // the largest tile that still lets two workgroups share a CU's 160 KB of LDS
  // (AND / min-match batches also keep a match counter byte per doc)
  // doc-tile units: plain disjunctions run as joined posting streams (join.h), the others as
  // work items (score.h)
  {
    const bool allow = join_allowed(b);
    b->tile_units.clear();
    b->join_units.clear();
    b->any_and = false;
    {
      std::vector<uint32_t> can;
      for (uint32_t u : b->all_tile_units)
        if (allow && unit_joinable(b, u)) can.push_back(u);
      const int forced = join_or_forced(b);
      if (forced == 0 || (forced < 0 && !join_or_pays(b, can))) can.clear();
      size_t at = 0;
      for (uint32_t u : b->all_tile_units) {
        if (at < can.size() && can[at] == u) {
          b->join_units.push_back(u);
          ++at;
        } else {
          b->tile_units.push_back(u);
          b->any_and = b->any_and || unit_counts_matches(b->queries[u]);
        }
      }
    }
    if (!b->phrase) {   // (a phrase batch's conj_units are its phrases, fixed at create)
      // a conjunction whose rarest term is far rarer than the rest is cheaper block driven
      // (conj.h decodes only the blocks the lead term's docs fall into): join_and_saving
      b->conj_units.clear();
      const int forced = join_and_forced(b);
      std::vector<uint32_t> joining;
      int64_t saved = 0;
      for (uint32_t u : b->all_conj_units) {
        const int64_t s = (allow && forced != 0 && unit_joinable(b, u))
                              ? (forced == 1 ? 1 : join_and_saving(b, b->queries[u])) : 0;
        if (s > 0) {
          joining.push_back(u);
          saved += s;
        }
      }
      // (a batch that joins plain disjunctions anyway has paid for k_join and the pilot: one more
      // k_join_score launch is all a joined conjunction adds)
      if (forced != 1 && saved < (b->join_units.empty() ? kJoinAndLaunchCost : kJoinAndLaunchCost / 10))
        joining.clear();
      size_t at = 0;
      for (uint32_t u : b->all_conj_units) {
        if (at < joining.size() && joining[at] == u) {
          b->join_units.push_back(u);
          ++at;
        } else {
          b->conj_units.push_back(u);
        }
      }
      if (build_conj_work(b) != IRS_HIP_OK) return false;
    }
    b->joined = !b->join_units.empty();
  }
  // 32-bit accumulators halve the LDS per doc: twice the tile at the same residency
  // the largest tile that still lets two workgroups share a CU's 160 KB of LDS
  // (AND / min-match batches also keep a match counter byte per doc)
  // (chosen anew on every deal: set_path / set_wand re-deal the units, and `any_and` with them)
  if (b->tile_asked) b->tile = b->tile_asked;
  else if (!b->phrase) b->tile = b->acc32 ? (b->any_and ? 8192 : 12288) : 6144;
  // per unit: tiles of its segment, its slice of the plan table and of the per-tile tables;
  // upper bound of its work items: every block once + one more per tile border it may
  // straddle + the decoded tail
  uint64_t first_words = 0, tiles = 0, item_bound = kItemSlack;
  b->n_tiles = 0xFFFFFFFFu;
  b->max_tiles = 0;
  b->join_max_tiles = 0;
  std::vector<uint8_t> is_join(b->nq, 0);
  for (uint32_t u : b->join_units) is_join[u] = 1;
  for (uint32_t u = 0; u < b->nq; ++u) {
    DevQuery& dq = b->queries[u];
    // (conjunctions are block driven unless they run as joined streams)
    const bool tiled = !b->phrase && ((dq.op & 0xFF) != 2 || is_join[u]);
    const uint32_t tile_docs = is_join[u] ? kJoinTile : b->tile;   // (streams are cut at kJoinTile)
    dq.n_tiles = tiled ? (b->segs[dq.seg]->dev.num_docs + tile_docs - 1) / tile_docs : 0u;
    if (b->phrase) dq.n_tiles = 1;
    if (tiled) b->n_tiles = std::min(b->n_tiles, dq.n_tiles);
    if (is_join[u]) {   // no plan table, no work items (k_plan / k_items_* skip the unit)
      dq.first_off = kNoPlan;
      dq.tile_base = 0;
      b->join_max_tiles = std::max(b->join_max_tiles, dq.n_tiles);
      continue;
    }
    dq.first_off = first_words;
    first_words += uint64_t(dq.n_tiles + 1) * b->jt;
    if (tiles + dq.n_tiles > 0xFFFFFF00ull) return false;
    dq.tile_base = uint32_t(tiles);
    tiles += dq.n_tiles;
    b->max_tiles = std::max(b->max_tiles, dq.n_tiles);
    if (tiled) {
      for (uint32_t j = 0; j < dq.n_terms; ++j)
        item_bound += uint64_t(b->segs[dq.seg]->terms[b->qterms[dq.first_term + j].term].nblk) +
                      dq.n_tiles + 1;
    }
  }
  if (item_bound > 0xFFFFFF00ull) return false;
  b->total_tiles = uint32_t(tiles);
  if (b->n_tiles == 0xFFFFFFFFu) b->n_tiles = 0;
  {  // k_score's queue order over the tiled units: heaviest first within every chunk round
    std::vector<std::pair<uint64_t, uint32_t>> work;
    for (uint32_t u : b->tile_units) {
      const DevQuery& dq = b->queries[u];
      uint64_t w = 0;
      for (uint32_t j = 0; j < dq.n_terms; ++j)
        w += b->segs[dq.seg]->terms[b->qterms[dq.first_term + j].term].docs_count;
      work.push_back({w, u});
    }
    // ... segment by segment: the workgroups resident at one time should keep reading the same
    // doc range (shared in L2), so units of one segment stay together
    std::stable_sort(work.begin(), work.end(), [&](const auto& x, const auto& y) {
      const uint32_t sx = b->queries[x.second].seg, sy = b->queries[y.second].seg;
      return sx != sy ? sx < sy : x.first > y.first;
    });
    for (uint32_t i = 0; i < work.size(); ++i) b->queries[i].run_unit = work[i].second;
  }
  b->stride_eff = (b->tile_units.empty() && b->join_units.empty())
                      ? b->stride
                      : std::max<uint32_t>(1, std::min<uint32_t>(b->stride, b->n_tiles / 2));
  if (const char* e = std::getenv("IRS_HIP_WG_THREADS")) {  // tuning knob
    const uint32_t t = uint32_t(std::atoi(e));
    if (t == 256 || t == 512 || t == 1024) b->wg_threads = t;
  }
  // k_pilot / k_score stage TILE norm bytes per tile, NormStage<TILE>::kPieces x 8 per thread
  {
    const uint32_t pieces = (b->tile + 4095u) / 4096u;
    uint32_t need = b->phrase ? 0u : (b->tile + 8u * pieces - 1u) / (8u * pieces);
    b->score_threads = b->wg_threads;
    while (b->score_threads < need) b->score_threads *= 2;
    b->nw_log2 = 0;
    while ((64u << b->nw_log2) < b->score_threads) ++b->nw_log2;
  }
  if (const char* e = std::getenv("IRS_HIP_JOIN_THREADS")) {  // tuning knob
    const uint32_t t = uint32_t(std::atoi(e));
    if (t == 256 || t == 512 || t == 1024) b->join_threads = t;
  }
  b->join_nw_log2 = 0;
  while ((64u << b->join_nw_log2) < b->join_threads) ++b->join_nw_log2;
  if (b->cand_cap == 0) b->cand_cap = default_cand_cap(b);
  const uint64_t rows = uint64_t(b->nq) * b->jt;
  // what every run starts from zero lives in ONE block (one fill per run instead of six: each is
  // a dispatch of its own in front of the first kernel): status, thresholds, candidate and hit
  // counters, the touched / pruned tallies
  {
    const uint64_t nq = b->nq;
    auto up = [](uint64_t v) { return (v + 255u) & ~uint64_t(255); };
    const uint64_t o_status = 0, o_bstar = 256, o_count = o_bstar + up(nq * 4),
                   o_hits = o_count + up(nq * 4), o_touched = o_hits + up(nq * 8),
                   o_pruned = o_touched + up(nq * 16), total = o_pruned + up(nq * 4);
    if (!b->d_zeroed.alloc(total)) return false;
    uint8_t* z = b->d_zeroed.as<uint8_t>();
    b->d_status.view(z + o_status, 4);
    b->d_bstar.view(z + o_bstar, nq * 4);
    b->d_cand_count.view(z + o_count, nq * 4);
    b->d_hits.view(z + o_hits, nq * 8);
    b->d_touched.view(z + o_touched, nq * 16);
    b->d_pruned.view(z + o_pruned, nq * 4);
  }
  if (!b->d_first.alloc(std::max<uint64_t>(first_words, 1) * sizeof(uint32_t)) ||
      !b->d_tails.alloc(rows * sizeof(DevTail)) ||
      !b->d_cands.alloc(uint64_t(b->nq) * b->cand_cap * sizeof(uint64_t)) ||
      !b->d_out.alloc(uint64_t(b->nq) * b->k_max * sizeof(Hit)) ||
      !b->d_out_count.alloc(b->nq * sizeof(uint32_t)) || !b->d_work.alloc(16))
    return false;
  if (b->joined && !build_streams(b)) return false;
  if (!build_groups(b)) return false;
  if (!b->tile_units.empty()) {
    if (!b->d_tile_units.alloc(b->tile_units.size() * 4) ||
        !b->up.copy(b->d_tile_units.p, b->tile_units.data(), b->tile_units.size() * 4))
      return false;
  }
  if (!b->phrase && !b->tile_units.empty()) {
    const uint64_t parts = (tiles + 1 + kScanChunk - 1) / kScanChunk;
    if (!b->d_tile_off.alloc((tiles + 1) * sizeof(uint32_t)) ||
        !b->d_scan_parts.alloc((parts + 1) * sizeof(uint64_t)) ||
        !b->d_items.alloc(item_bound * sizeof(ItemG)) ||
        !b->d_score_args.alloc(sizeof(ScoreArgs)) ||
        !b->d_tile_ub.alloc((tiles + 1) * sizeof(float)))
      return false;
  }
  // (the unit records last: dealing the units and building the streams filled fields in)
  if (!b->up.copy(b->d_queries.p, b->queries.data(), b->queries.size() * sizeof(DevQuery)))
    return false;
  b->scratch_ready = true;
  return true;
}

}  // namespace

// Runs one entry point's body; exceptions (std::bad_alloc out of a std::vector, anything
// else) become status codes: the ABI promises that nothing is thrown across it.
template<typename F>
int guarded(F&& f) noexcept {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return IRS_HIP_ENOMEM;
  } catch (...) {
    return IRS_HIP_EHIP;
  }
}

extern "C" {
static int run_impl(irs_hip_batch* b, rt::stream_t st);
}

// ---- the host half of a run, off the caller's thread -------------------------------------------
// One worker thread per device (started with the first run).  A caller that serves batches in a
// loop prepares and creates batch i + 1 while the worker deals the units of batch i, builds its
// streams and work lists and queues its uploads and kernels; the GPU meanwhile executes batch
// i - 1.  At one GPU the kernels hide all of it; with 8 ranks a rank's kernels take as long as this
// host work (index-search runs its queries on --threads workers for the same reason:
// index-search.cpp:673-722).  IRS_HIP_ASYNC_RUN=0 runs everything on the caller's thread.
namespace worker {
// A job = the host half of a run, or the recovery of a batch whose threshold spans ranks: both
// issue collectives on the batch's communicator, and RCCL pairs collectives by issue ORDER — so
// everything of a device that may issue one goes through this one FIFO (a re-run of
// batch i on the caller's thread raced the worker's run of batch i + 1 on the same communicator).
struct Job {
  irs_hip_batch* b;
  std::function<int()> fn;
};
struct Queue {
  std::mutex m;
  std::condition_variable cv;
  std::deque<Job> jobs;
  std::thread th;
  bool started = false, stop = false;
  int device = 0;
  ~Queue() {
    {
      std::lock_guard<std::mutex> lock(m);
      stop = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};
static Queue& of(int device) {
  static Queue queues[pool::kMaxDevices];
  return queues[device >= 0 && device < pool::kMaxDevices ? device : 0];
}
static void loop(Queue* q) {
  rt::set_device(q->device);
  for (;;) {
    Job job;
    {
      std::unique_lock<std::mutex> lock(q->m);
      q->cv.wait(lock, [&] { return q->stop || !q->jobs.empty(); });
      if (q->jobs.empty()) return;   // (stop)
      job = q->jobs.front();
      q->jobs.pop_front();
    }
    const int rc = guarded([&] {
      if (!rt::set_device(job.b->seg->device)) return int(IRS_HIP_EHIP);
      return job.fn();
    });
    {
      std::lock_guard<std::mutex> lock(job.b->am);
      job.b->async_rc = rc;
      job.b->async_pending = false;
    }
    job.b->acv.notify_all();
  }
}
static bool enabled() {
  static const bool on = [] {
    const char* e = std::getenv("IRS_HIP_ASYNC_RUN");
    return !e || std::atoi(e) != 0;
  }();
  return on;
}
// Queues `fn` for the batch on its device's worker; false: the worker cannot take it (no thread,
// no memory) — nothing is pending, the caller runs `fn` itself.
static bool submit(irs_hip_batch* b, std::function<int()> fn) {
  Queue& q = of(b->seg->device);
  try {
    std::lock_guard<std::mutex> lock(q.m);
    if (!q.started) {
      q.device = b->seg->device;
      q.th = std::thread(loop, &q);
      q.started = true;
    }
    {
      std::lock_guard<std::mutex> block(b->am);
      b->async_pending = true;
      b->async_rc = IRS_HIP_OK;
    }
    try {
      q.jobs.push_back(Job{b, std::move(fn)});
    } catch (...) {
      std::lock_guard<std::mutex> block(b->am);
      b->async_pending = false;   // (nothing was queued: nobody would ever clear it)
      throw;
    }
    q.cv.notify_one();
    return true;
  } catch (...) {
    return false;
  }
}
static bool wanted(const irs_hip_batch* b) { return b->async_pref < 0 ? enabled() : b->async_pref != 0; }
}  // namespace worker

// Before anything else touches a batch: its run, if one was handed to the worker, is queued.
// Returns what that run returned (once).
static int settle(irs_hip_batch* b) {
  if (!b) return IRS_HIP_OK;
  std::unique_lock<std::mutex> lock(b->am);
  b->acv.wait(lock, [&] { return !b->async_pending; });
  const int rc = b->async_rc;
  b->async_rc = IRS_HIP_OK;
  return rc;
}
// an entry point's body behind the batch's pending run
template<typename F>
static int settled(irs_hip_batch* b, F&& f) noexcept {
  return guarded([&] {
    if (const int rc = settle(b)) return rc;
    return f();
  });
}

extern "C" {

uint32_t irs_hip_abi_version(void) { return IRS_HIP_ABI_VERSION; }

const char* irs_hip_strerror(int status) {
  switch (status) {
    case IRS_HIP_OK: return "ok";
    case IRS_HIP_EINVAL: return "invalid argument";
    case IRS_HIP_ECORRUPT: return "corrupt postings data";
    case IRS_HIP_ENOMEM: return "out of memory";
    case IRS_HIP_EHIP: return "HIP runtime error or no gfx950 device";
    case IRS_HIP_EOVERFLOW: return "candidate buffer overflow";
    case IRS_HIP_EUNSUPPORTED: return "unsupported";
    case IRS_HIP_EPEER: return "another rank could not re-execute its batch";
    default: return "unknown status";
  }
}

static int device_arch_impl(int32_t device, char* buf, size_t cap) {
  if (!buf || !cap) return IRS_HIP_EINVAL;
  if (device < 0 || device >= rt::device_count() || !rt::device_arch(device, buf, cap))
    return IRS_HIP_EHIP;
  return IRS_HIP_OK;
}

static int segment_open_impl(const irs_hip_segment_desc* d, irs_hip_segment** out) {
  if (!d || !out) return IRS_HIP_EINVAL;
  *out = nullptr;
  if (!d->doc_file || !d->num_docs || d->num_docs > 0x7FFF0000u ||
      (d->layout != IRS_HIP_LAYOUT_SCALAR && d->layout != IRS_HIP_LAYOUT_SIMD4) ||
      (d->num_terms && !d->terms) || d->wand_count > 16 || d->wand_type > IRS_HIP_WAND_MIN_NORM ||
      (d->doc_mask_count && !d->doc_mask))
    return IRS_HIP_EINVAL;
  if (d->norm_kind != IRS_HIP_NORM2 && d->norm_kind != IRS_HIP_NORM_LEGACY) return IRS_HIP_EINVAL;
  if (d->norms) {
    if (d->norm_width != 1 && d->norm_width != 2 && d->norm_width != 4) return IRS_HIP_EINVAL;
    if (d->norm_kind == IRS_HIP_NORM_LEGACY && d->norm_width != 4) return IRS_HIP_EINVAL;
    // dense column covering every doc (columnstore2.cpp:650-789); sparse columns
    // are not on the benchmark path
    if (d->norm_min_doc != kDocMin || d->norm_count < d->num_docs) return IRS_HIP_EUNSUPPORTED;
  }
  int32_t version = -1;
  const size_t hdr = check_doc_header(d->doc_file, d->doc_file_len, &version);
  if (!hdr) return IRS_HIP_ECORRUPT;
  if (d->doc_file_len >= 0xFFFFFF00ull) return IRS_HIP_EUNSUPPORTED;  // block offsets are u32
  // PostingsFormat: odd versions are the SSE (simd4) layouts (formats_10.cpp:283-313)
  if (version < 0 || version > 5) return IRS_HIP_ECORRUPT;
  if ((version & 1) != (d->layout == IRS_HIP_LAYOUT_SIMD4 ? 1 : 0)) return IRS_HIP_EINVAL;
  size_t pos_hdr = 0;
  if (d->pos_file) {
    // positions need frequencies (IndexFeatures::POS implies FREQ)
    int32_t pos_version = -1;
    pos_hdr = check_pos_header(d->pos_file, d->pos_file_len, &pos_version);
    if (!pos_hdr) return IRS_HIP_ECORRUPT;
    if (pos_version != version) return IRS_HIP_ECORRUPT;
    if (!d->has_freq) return IRS_HIP_EINVAL;
    if (d->pos_features & ~(IRS_HIP_POS_OFFSETS | IRS_HIP_POS_PAYLOADS)) return IRS_HIP_EINVAL;
    if (d->pos_features) return IRS_HIP_EUNSUPPORTED;  // the `.pos` tail interleaves them
  }
  if (!device_usable(d->device)) return IRS_HIP_EHIP;

  irs_hip_segment* s = new (std::nothrow) irs_hip_segment;
  if (!s) return IRS_HIP_ENOMEM;
  s->device = d->device;
  s->cus = std::max(1, rt::device_cus(d->device));
  int rc = IRS_HIP_OK;
  do {
    try {
      s->terms.resize(d->num_terms);
    } catch (...) {
      rc = IRS_HIP_ENOMEM;
      break;
    }
    uint64_t blocks = 0, tail_rows = 0;
    for (uint32_t i = 0; i < d->num_terms && rc == IRS_HIP_OK; ++i) {
      const irs_hip_term_meta& m = d->terms[i];
      DevTerm t{};
      t.docs_count = m.docs_count;
      t.tail_row = uint32_t(tail_rows);
      tail_rows += m.docs_count == 1 ? 1u : m.docs_count % kBlock;
      if (tail_rows > 0xFFFFFF00ull) rc = IRS_HIP_EUNSUPPORTED;
      if (m.docs_count == 1) {
        t.single_doc = kDocMin + uint32_t(m.e_skip_start);  // formats_10.cpp:1887
        t.single_freq = m.freq;
        t.doc_start = 0;
      } else if (m.docs_count > 1) {
        if (m.doc_start < hdr || m.doc_start >= d->doc_file_len) rc = IRS_HIP_ECORRUPT;
        t.doc_start = m.doc_start;
        t.nblk = m.docs_count / kBlock;
        t.tail_n = m.docs_count % kBlock;
        t.dir_off = blocks;
        blocks += t.nblk;
        // block offsets are kept as u32 relative to doc_start
        if (m.docs_count > kBlock && m.e_skip_start > 0xFFFFFFFFull) rc = IRS_HIP_EUNSUPPORTED;
        if (m.docs_count > kBlock && d->wand_count) {
          if (s->skip_at.empty()) s->skip_at.assign(d->num_terms, 0);
          s->skip_at[i] = m.doc_start + m.e_skip_start;
          if (s->skip_at[i] >= d->doc_file_len) rc = IRS_HIP_ECORRUPT;
        }
      }
      s->terms[i] = t;
    }
    if (rc != IRS_HIP_OK) break;
    s->total_blocks = blocks;
    s->has_pos = d->pos_file != nullptr;
    s->wand_type = d->wand_type;
    const uint64_t norm_bytes = d->norms ? uint64_t(d->norm_width) * d->norm_count : 0;
    if (!s->d_doc.alloc(d->doc_file_len + kPadBytes) ||
        (d->norms && !s->d_norms.alloc(norm_bytes + kPadBytes)) ||
        !s->d_terms.alloc(std::max<size_t>(1, s->terms.size()) * sizeof(DevTerm)) ||
        !s->d_blk_off.alloc((blocks + 1) * 4) || !s->d_blk_last.alloc((blocks + 1) * 4) ||
        !s->d_blk_bits.alloc((blocks + 1) * 2) || !s->d_blk_aoff.alloc((blocks + 1) * 4) ||
        !s->d_blk_dir.alloc((blocks + 1) * sizeof(BlkDir)) ||
        !s->d_blk_term.alloc((blocks + 1) * 4) ||
        !s->d_tail_docs.alloc((tail_rows + 1) * 4) || !s->d_tail_freqs.alloc((tail_rows + 1) * 4) ||
        !s->d_status.alloc(4)) {
      rc = IRS_HIP_ENOMEM;
      break;
    }
    bool okc = rt::h2d(s->d_doc.p, d->doc_file, d->doc_file_len, nullptr) &&
               rt::dmemset(s->d_doc.as<uint8_t>() + d->doc_file_len, 0, kPadBytes, nullptr) &&
               rt::h2d(s->d_terms.p, s->terms.data(), s->terms.size() * sizeof(DevTerm), nullptr);
    if (d->norms) {
      okc = okc && rt::h2d(s->d_norms.p, d->norms, norm_bytes, nullptr) &&
            rt::dmemset(s->d_norms.as<uint8_t>() + norm_bytes, 0, kPadBytes, nullptr);
    }
    if (!okc || !rt::sync(nullptr)) {
      rc = IRS_HIP_EHIP;
      break;
    }
    DevSegment& v = s->dev;
    v.doc = s->d_doc.as<uint8_t>();
    v.doc_len = d->doc_file_len;
    v.norms = d->norms ? s->d_norms.as<uint8_t>() : nullptr;
    v.norm_width = d->norms ? d->norm_width : 0;
    v.norm_min_doc = d->norms ? d->norm_min_doc : kDocMin;
    v.norm_count = d->norms ? d->norm_count : 0;
    v.norm_legacy = (d->norms && d->norm_kind == IRS_HIP_NORM_LEGACY) ? 1u : 0u;
    v.terms = s->d_terms.as<DevTerm>();
    v.num_terms = d->num_terms;
    v.num_docs = d->num_docs;
    v.blk_off = s->d_blk_off.as<uint32_t>();
    v.blk_last = s->d_blk_last.as<uint32_t>();
    v.blk_bits = s->d_blk_bits.as<uint16_t>();
    v.blk_aoff = s->d_blk_aoff.as<uint32_t>();
    v.blk_dir = s->d_blk_dir.as<BlkDir>();
    v.blk_term = s->d_blk_term.as<uint32_t>();
    v.tail_docs = s->d_tail_docs.as<uint32_t>();
    v.tail_freqs = s->d_tail_freqs.as<uint32_t>();
    v.pk = nullptr;  // set by build_packed_image
    v.has_freq = d->has_freq ? 1 : 0;
    v.layout = d->layout;
    v.wand_count = d->wand_count;
    s->live_docs = d->num_docs;
    if (d->doc_mask_count) {
      // DocumentMask -> bitmap, bit (doc - kDocMin); a whole doc tile behind the last doc stays
      // readable (the tile kernels test their accumulators' docs group by group)
      const uint64_t words = (uint64_t(d->num_docs) + 12288u + 31u) / 32u + 16u;
      std::vector<uint32_t> bits;
      try {
        bits.assign(words, 0u);
      } catch (...) {
        rc = IRS_HIP_ENOMEM;
        break;
      }
      uint64_t gone = 0;
      for (uint64_t i = 0; i < d->doc_mask_count; ++i) {
        const uint32_t doc = d->doc_mask[i];
        if (doc < kDocMin || doc > d->num_docs) continue;
        const uint32_t j = doc - kDocMin;
        gone += (bits[j >> 5] >> (j & 31u)) & 1u ? 0u : 1u;
        bits[j >> 5] |= 1u << (j & 31u);
      }
      if (gone) {
        if (!s->d_dead.alloc(words * 4)) {
          rc = IRS_HIP_ENOMEM;
          break;
        }
        if (!rt::h2d(s->d_dead.p, bits.data(), words * 4, nullptr) || !rt::sync(nullptr)) {
          rc = IRS_HIP_EHIP;
          break;
        }
        v.dead = s->d_dead.as<uint32_t>();
        s->live_docs = d->num_docs - gone;
      }
    }
    rc = d->layout == IRS_HIP_LAYOUT_SIMD4 ? build_directory<kSimd4>(s)
                                           : build_directory<kScalar>(s);
    if (rc == IRS_HIP_OK && d->pos_file) {
      std::vector<uint64_t> pos_end;
      uint64_t rows = 0, ptail_rows = 0;
      try {
        s->pterms.resize(d->num_terms);
        pos_end.resize(d->num_terms);
      } catch (...) {
        rc = IRS_HIP_ENOMEM;
        break;
      }
      for (uint32_t i = 0; i < d->num_terms && rc == IRS_HIP_OK; ++i) {
        const irs_hip_term_meta& m = d->terms[i];
        DevPosTerm pt{};
        if (m.docs_count) {
          if (m.freq < m.docs_count || m.pos_start < pos_hdr || m.pos_start > d->pos_file_len)
            rc = IRS_HIP_ECORRUPT;
          pt.pos_start = m.pos_start;
          pt.total = m.freq;
          pt.nfull = m.freq / kBlock;
          pt.tail_n = m.freq % kBlock;
          pt.row = rows;
          rows += pt.nfull;
          pt.tail_row = uint32_t(ptail_rows);
          ptail_rows += pt.tail_n;
          if (ptail_rows > 0xFFFFFF00ull) rc = IRS_HIP_EUNSUPPORTED;
        }
        s->pterms[i] = pt;
        pos_end[i] = m.pos_end;
      }
      if (rc != IRS_HIP_OK) break;
      if (!s->d_pos.alloc(d->pos_file_len + kPadBytes) ||
          !s->d_pterms.alloc(std::max<size_t>(1, s->pterms.size()) * sizeof(DevPosTerm)) ||
          !s->d_pblk_off.alloc((rows + 1) * 4) || !s->d_pblk_bits.alloc(rows + 1) ||
          !s->d_blk_pos.alloc((blocks + 1) * 4) ||
          !s->d_ptail.alloc((ptail_rows + 1) * 4)) {
        rc = IRS_HIP_ENOMEM;
        break;
      }
      if (!rt::h2d(s->d_pos.p, d->pos_file, d->pos_file_len, nullptr) ||
          !rt::dmemset(s->d_pos.as<uint8_t>() + d->pos_file_len, 0, kPadBytes, nullptr) ||
          !rt::h2d(s->d_pterms.p, s->pterms.data(), s->pterms.size() * sizeof(DevPosTerm),
                   nullptr) ||
          !rt::sync(nullptr)) {
        rc = IRS_HIP_EHIP;
        break;
      }
      v.pos = s->d_pos.as<uint8_t>();
      v.pos_len = d->pos_file_len;
      v.pterms = s->d_pterms.as<DevPosTerm>();
      v.pblk_off = s->d_pblk_off.as<uint32_t>();
      v.pblk_bits = s->d_pblk_bits.as<uint8_t>();
      v.blk_pos = s->d_blk_pos.as<uint32_t>();
      v.ptail = s->d_ptail.as<uint32_t>();
      // PostingsFormat < POSITIONS_ZEROBASED (formats_10.cpp:283-304): one-based storage
      v.pos_base = version < 2 ? 1u : 0u;
      rc = d->layout == IRS_HIP_LAYOUT_SIMD4 ? build_positions<kSimd4>(s, pos_end)
                                             : build_positions<kScalar>(s, pos_end);
    }
    s->device_bytes = s->d_doc.n + s->d_norms.n + s->d_terms.n + s->d_blk_off.n +
                      s->d_blk_last.n + s->d_blk_bits.n + s->d_blk_aoff.n + s->d_blk_dir.n + s->d_blk_term.n + s->d_pk.n +
                      s->d_tail_docs.n + s->d_tail_freqs.n + s->d_pos.n + s->d_pterms.n +
                      s->d_pblk_off.n + s->d_pblk_bits.n + s->d_blk_pos.n + s->d_ptail.n + s->d_dead.n;
  } while (false);
  if (rc != IRS_HIP_OK) {
    delete s;
    return rc;
  }
  *out = s;
  return IRS_HIP_OK;
}

void irs_hip_segment_close(irs_hip_segment* seg) {
  if (!seg) return;
  rt::set_device(seg->device);
  pool::tl_free_now = true;   // (its buffers are freed, not pooled)
  delete seg;
  pool::tl_free_now = false;
}

uint64_t irs_hip_segment_device_bytes(const irs_hip_segment* seg) {
  return seg ? seg->device_bytes : 0;
}
uint64_t irs_hip_segment_live_docs(const irs_hip_segment* seg) { return seg ? seg->live_docs : 0; }

static int decode_term_impl(irs_hip_segment* seg, uint32_t term, uint32_t* docs, uint32_t* freqs,
                        uint32_t cap, uint32_t* count) {
  if (!seg || !docs || !count || term >= seg->dev.num_terms) return IRS_HIP_EINVAL;
  if (freqs && !seg->dev.has_freq) return IRS_HIP_EINVAL;
  if (!rt::set_device(seg->device)) return IRS_HIP_EHIP;
  const DevTerm& t = seg->terms[term];
  *count = t.docs_count;
  if (t.docs_count == 0) return IRS_HIP_OK;
  if (cap < t.docs_count) return IRS_HIP_EINVAL;
  DevBuf dd, df;
  const size_t bytes = size_t(t.docs_count) * 4;
  if (!dd.alloc(bytes) || (freqs && !df.alloc(bytes))) return IRS_HIP_ENOMEM;
  const uint32_t items = t.nblk + 1;
  const uint32_t grid = (items + kWaves - 1) / kWaves;
  if (seg->dev.layout == kSimd4) {
    RT_LAUNCH((k_decode_term<kSimd4>), grid, kThreads, 0, nullptr, seg->dev, term,
              dd.as<uint32_t>(), freqs ? df.as<uint32_t>() : nullptr);
  } else {
    RT_LAUNCH((k_decode_term<kScalar>), grid, kThreads, 0, nullptr, seg->dev, term,
              dd.as<uint32_t>(), freqs ? df.as<uint32_t>() : nullptr);
  }
  if (!rt::last_error_ok() || !rt::d2h(docs, dd.p, bytes, nullptr) ||
      (freqs && !rt::d2h(freqs, df.p, bytes, nullptr)) || !rt::sync(nullptr))
    return IRS_HIP_EHIP;
  return IRS_HIP_OK;
}

static int decode_positions_impl(irs_hip_segment* seg, uint32_t term, uint32_t* positions,
                             uint64_t cap, uint64_t* count) {
  if (!seg || !positions || !count || term >= seg->dev.num_terms) return IRS_HIP_EINVAL;
  if (!seg->dev.pos) return IRS_HIP_EINVAL;  // the segment was opened without `.pos`
  if (!rt::set_device(seg->device)) return IRS_HIP_EHIP;
  const DevTerm& t = seg->terms[term];
  const uint64_t total = seg->pterms[term].total;
  *count = total;
  if (t.docs_count == 0 || total == 0) return IRS_HIP_OK;
  if (cap < total) return IRS_HIP_EINVAL;
  DevBuf dp;
  if (!dp.alloc(size_t(total) * 4)) return IRS_HIP_ENOMEM;
  const uint32_t items = t.nblk + 1;
  const uint32_t grid = (items + kWaves - 1) / kWaves;
  if (seg->dev.layout == kSimd4) {
    RT_LAUNCH((k_decode_positions<kSimd4>), grid, kThreads, 0, nullptr, seg->dev, term,
              dp.as<uint32_t>());
  } else {
    RT_LAUNCH((k_decode_positions<kScalar>), grid, kThreads, 0, nullptr, seg->dev, term,
              dp.as<uint32_t>());
  }
  if (!rt::last_error_ok() || !rt::d2h(positions, dp.p, size_t(total) * 4, nullptr) ||
      !rt::sync(nullptr))
    return IRS_HIP_EHIP;
  return IRS_HIP_OK;
}

static int bit_union_impl(irs_hip_segment* seg, const uint32_t* terms, uint32_t n_terms,
                      uint64_t* set, uint64_t n_words, uint64_t* count) {
  if (!seg || (!terms && n_terms) || !set || !n_words) return IRS_HIP_EINVAL;
  if (!rt::set_device(seg->device)) return IRS_HIP_EHIP;
  uint64_t total = 0;
  for (uint32_t i = 0; i < n_terms; ++i) {
    if (terms[i] == IRS_HIP_NO_TERM) continue;
    if (terms[i] >= seg->dev.num_terms) return IRS_HIP_EINVAL;
    total += seg->terms[terms[i]].docs_count;  // formats_10.cpp:3796, 3802
  }
  if (count) *count = total;
  if (!n_terms) return IRS_HIP_OK;
  // work list: up to kUnionBlocks blocks of one term per workgroup (+ its tail)
  std::vector<UnionWg> wgs;
  try {
    for (uint32_t i = 0; i < n_terms; ++i) {
      if (terms[i] == IRS_HIP_NO_TERM) continue;
      const DevTerm& t = seg->terms[terms[i]];
      if (t.docs_count == 0) continue;
      uint32_t b = 0;
      do {
        wgs.push_back(UnionWg{terms[i], b, 0u, 0u});
        b += kUnionBlocks;
      } while (b < t.nblk);
    }
  } catch (...) {
    return IRS_HIP_ENOMEM;
  }
  if (wgs.empty()) return IRS_HIP_OK;
  if (wgs.size() > 0x7FFFFFFFull) return IRS_HIP_EUNSUPPORTED;
  DevBuf d_wgs, d_set;
  const size_t set_bytes = size_t(n_words) * 8;
  if (!d_wgs.alloc(wgs.size() * sizeof(UnionWg)) || !d_set.alloc(set_bytes)) return IRS_HIP_ENOMEM;
  // (bits already set by the caller are kept: the set goes up first.  Round 6 tried to leave the
  // upload out when the caller's set is empty — a scan of it + a device memset — and to stage the
  // result through page-locked memory of the pool: 0.39 and 0.54 ms per call against 0.245; the
  // runtime's own staging of pageable copies is the fastest of the three at 1.25 MB.)
  if (!rt::h2d(d_wgs.p, wgs.data(), wgs.size() * sizeof(UnionWg), nullptr) ||
      !rt::h2d(d_set.p, set, set_bytes, nullptr))
    return IRS_HIP_EHIP;
  const uint64_t n_bits = n_words * 64;
  if (seg->dev.layout == kSimd4) {
    RT_LAUNCH((k_bit_union<kSimd4>), uint32_t(wgs.size()), kThreads, 0, nullptr, seg->dev,
              d_wgs.as<UnionWg>(), d_set.as<uint32_t>(), n_bits);
  } else {
    RT_LAUNCH((k_bit_union<kScalar>), uint32_t(wgs.size()), kThreads, 0, nullptr, seg->dev,
              d_wgs.as<UnionWg>(), d_set.as<uint32_t>(), n_bits);
  }
  if (!rt::last_error_ok() || !rt::d2h(set, d_set.p, set_bytes, nullptr) || !rt::sync(nullptr))
    return IRS_HIP_EHIP;
  return IRS_HIP_OK;
}

// Several unions at once, only their populations coming back: the bitsets stay on the device
// (one per set of a pass; passes of at most ~1 GB of them).
static int bit_union_counts_impl(irs_hip_segment* seg, const uint32_t* terms, const uint32_t* offsets,
                                 uint32_t n_sets, uint64_t* counts) {
  if (!seg || !offsets || !counts || (!terms && n_sets && offsets[n_sets] != offsets[0])) return IRS_HIP_EINVAL;
  if (!rt::set_device(seg->device)) return IRS_HIP_EHIP;
  for (uint32_t i = 0; i < n_sets; ++i) {
    if (offsets[i + 1] < offsets[i]) return IRS_HIP_EINVAL;
    counts[i] = 0;
  }
  if (!n_sets) return IRS_HIP_OK;
  for (uint32_t i = offsets[0]; i < offsets[n_sets]; ++i)
    if (terms[i] != IRS_HIP_NO_TERM && terms[i] >= seg->dev.num_terms) return IRS_HIP_EINVAL;
  const uint64_t n_words = (uint64_t(seg->dev.num_docs) + 64) / 64;   // bit index = doc id
  const uint64_t words32 = n_words * 2, n_bits = n_words * 64;
  const uint32_t per_pass = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(n_sets, (1ull << 30) / (n_words * 8))));
  DevBuf d_sets, d_wgs, d_counts;
  if (!d_sets.alloc(uint64_t(per_pass) * n_words * 8) || !d_counts.alloc(uint64_t(per_pass) * 8)) return IRS_HIP_ENOMEM;
  std::vector<UnionWg> wgs;
  std::vector<unsigned long long> got(per_pass);
  for (uint32_t s0 = 0; s0 < n_sets; s0 += per_pass) {
    const uint32_t ns = std::min(per_pass, n_sets - s0);
    wgs.clear();
    for (uint32_t s = 0; s < ns; ++s) {
      for (uint32_t i = offsets[s0 + s]; i < offsets[s0 + s + 1]; ++i) {
        if (terms[i] == IRS_HIP_NO_TERM) continue;
        const DevTerm& t = seg->terms[terms[i]];
        if (t.docs_count == 0) continue;
        uint32_t b = 0;
        do {
          wgs.push_back(UnionWg{terms[i], b, s, 0u});
          b += kUnionBlocks;
        } while (b < t.nblk);
      }
    }
    if (wgs.size() > 0x7FFFFFFFull) return IRS_HIP_EUNSUPPORTED;
    if (!rt::dmemset(d_sets.p, 0, uint64_t(ns) * n_words * 8, nullptr)) return IRS_HIP_EHIP;
    if (!wgs.empty()) {
      if (!d_wgs.alloc(wgs.size() * sizeof(UnionWg))) return IRS_HIP_ENOMEM;
      if (!rt::h2d(d_wgs.p, wgs.data(), wgs.size() * sizeof(UnionWg), nullptr)) return IRS_HIP_EHIP;
      if (seg->dev.layout == kSimd4) {
        RT_LAUNCH((k_bit_union<kSimd4>), uint32_t(wgs.size()), kThreads, 0, nullptr, seg->dev,
                  d_wgs.as<UnionWg>(), d_sets.as<uint32_t>(), n_bits);
      } else {
        RT_LAUNCH((k_bit_union<kScalar>), uint32_t(wgs.size()), kThreads, 0, nullptr, seg->dev,
                  d_wgs.as<UnionWg>(), d_sets.as<uint32_t>(), n_bits);
      }
    }
    RT_LAUNCH(k_union_counts, ns, kThreads, 0, nullptr, d_sets.as<uint32_t>(), words32,
              d_counts.as<unsigned long long>());
    if (!rt::last_error_ok() || !rt::d2h(got.data(), d_counts.p, uint64_t(ns) * 8, nullptr) || !rt::sync(nullptr))
      return IRS_HIP_EHIP;
    for (uint32_t s = 0; s < ns; ++s) counts[s0 + s] = got[s];
  }
  return IRS_HIP_OK;
}

static int term_directory_impl(irs_hip_segment* seg, uint32_t term, uint32_t* last_docs,
                           uint64_t* offsets, uint32_t cap, uint32_t* count) {
  if (!seg || !count || term >= seg->dev.num_terms) return IRS_HIP_EINVAL;
  if (!rt::set_device(seg->device)) return IRS_HIP_EHIP;
  const DevTerm& t = seg->terms[term];
  *count = t.nblk;
  if (!t.nblk) return IRS_HIP_OK;
  if (cap < t.nblk || !last_docs || !offsets) return IRS_HIP_EINVAL;
  std::vector<uint32_t> rel(t.nblk);
  if (!rt::d2h(last_docs, seg->d_blk_last.as<uint32_t>() + t.dir_off, size_t(t.nblk) * 4,
               nullptr) ||
      !rt::d2h(rel.data(), seg->d_blk_off.as<uint32_t>() + t.dir_off, size_t(t.nblk) * 4,
               nullptr) ||
      !rt::sync(nullptr))
    return IRS_HIP_EHIP;
  for (uint32_t i = 0; i < t.nblk; ++i) offsets[i] = t.doc_start + rel[i];
  return IRS_HIP_OK;
}

// ----------------------------------------------------------------- batch --

static int batch_create_impl(irs_hip_segment* seg, const irs_hip_query* queries, uint32_t nq,
                         const irs_hip_term_scorer* terms, uint32_t n_entries,
                         irs_hip_batch** out) {
  return irs_hip_batch_create_multi(&seg, 1, queries, nq, terms, n_entries, out);
}

static int batch_create_multi_impl(irs_hip_segment* const* segs, uint32_t n_segs,
                               const irs_hip_query* queries, uint32_t nq_user,
                               const irs_hip_term_scorer* all_terms, uint32_t n_entries,
                               irs_hip_batch** out) {
  if (!segs || !n_segs || !queries || !all_terms || !out || !nq_user) return IRS_HIP_EINVAL;
  HostTrace trace("batch_create (query records)");
  *out = nullptr;
  if (uint64_t(n_segs) * nq_user > 0x7FFFFFFFull) return IRS_HIP_EINVAL;
  for (uint32_t s = 0; s < n_segs; ++s) {
    if (!segs[s]) return IRS_HIP_EINVAL;
    if (!segs[s]->dev.has_freq) return IRS_HIP_EUNSUPPORTED;  // scorers need IndexFeatures::FREQ
    // one launch per kernel covers every segment: same device, same block layout
    if (segs[s]->device != segs[0]->device) return IRS_HIP_EINVAL;
    if (segs[s]->dev.layout != segs[0]->dev.layout) return IRS_HIP_EUNSUPPORTED;
  }
  if (!rt::set_device(segs[0]->device)) return IRS_HIP_EHIP;
  irs_hip_batch* b = new (std::nothrow) irs_hip_batch;
  if (!b) return IRS_HIP_ENOMEM;
  b->seg = segs[0];
  const uint32_t nq = n_segs * nq_user;
  b->nq = nq;
  b->nq_user = nq_user;
  int rc = IRS_HIP_OK;
  try {
    b->segs.assign(segs, segs + n_segs);
    b->queries.resize(nq);
    b->count_precise.assign(nq, 0);
    b->group_upper.assign(nq, 0.0);
    b->qterms.reserve(size_t(n_entries) * n_segs);
    std::vector<int> exps;
    exps.reserve(nq);
    std::vector<DevQTerm> row;
    std::vector<double> smins;
    for (uint32_t q = 0; q < nq && rc == IRS_HIP_OK; ++q) {
      // unit q = (segment q / nq_user, query q % nq_user); the segment's own term entries
      irs_hip_segment* seg = segs[q / nq_user];
      const irs_hip_term_scorer* terms = all_terms + size_t(q / nq_user) * n_entries;
      const irs_hip_query& in = queries[q % nq_user];
      if ((in.op != IRS_HIP_OP_OR && in.op != IRS_HIP_OP_AND && in.op != IRS_HIP_OP_MINMATCH &&
           in.op != IRS_HIP_OP_PHRASE) ||
          in.n_terms == 0 || in.merge > IRS_HIP_MERGE_MIN ||
          (in.op == IRS_HIP_OP_PHRASE && in.merge != IRS_HIP_MERGE_SUM) ||
          in.n_terms > IRS_HIP_MAX_TERMS || in.k == 0 || in.k > IRS_HIP_MAX_K ||
          uint64_t(in.first_term) + in.n_terms > n_entries) {
        rc = IRS_HIP_EINVAL;
        break;
      }
      const bool is_phrase = in.op == IRS_HIP_OP_PHRASE;
      if (q == 0) b->phrase = is_phrase;
      if (is_phrase != b->phrase) {  // a batch holds phrase queries only, or none
        rc = IRS_HIP_EUNSUPPORTED;
        break;
      }
      if (is_phrase) {
        if (in.n_terms > IRS_HIP_MAX_PHRASE_TERMS || terms[in.first_term].phrase_offset != 0) {
          rc = IRS_HIP_EINVAL;
          break;
        }
        if (!seg->dev.pos) {  // FixedPhraseQuery needs FREQ | POS (phrase_query.cpp:63-66)
          rc = IRS_HIP_EUNSUPPORTED;
          break;
        }
      }
      row.clear();     // (one allocation for the whole batch: 8000 units otherwise pay 16000)
      smins.clear();   // per present term: the smallest score of one posting
      bool absent = false, same_bound = true;
      double upper = 0.0, min_score = 1e300, upper_all = 0.0;
      for (uint32_t j = 0; j < in.n_terms; ++j) {
        const irs_hip_term_scorer& ts = terms[in.first_term + j];
        DevQTerm qt{};
        qt.term = ts.term;
        qt.c0 = ts.c0;
        qt.norm_const = ts.norm_const;
        qt.norm_length = ts.norm_length;
        qt.cache_id = kMaxCaches;
        qt.pad0 = is_phrase ? ts.phrase_offset : 0u;
        if (ts.term != IRS_HIP_NO_TERM && ts.term >= seg->dev.num_terms) rc = IRS_HIP_EINVAL;
        if (!(ts.c0 >= 0.f) || !std::isfinite(ts.c0)) rc = IRS_HIP_EINVAL;
        if (rc != IRS_HIP_OK) break;
        // (a zero boost is legal: every posting then scores 0 — the fixed-point accumulators
        // still mark the doc as matched, and sums below kMaxTerms units come back as 0)
        const bool norms = seg->dev.norms != nullptr;
        const bool legacy = norms && seg->dev.norm_legacy;
        switch (ts.kind) {
          case IRS_HIP_SCORE_BM25:
            qt.kind = !norms ? kBM25One
                      : legacy ? kBM25Legacy
                               : (seg->dev.norm_width == 1 ? kBM25Tiny : kBM25Wide);
            if (!(ts.norm_const + ts.norm_length > 0.f)) rc = IRS_HIP_EINVAL;
            break;
          case IRS_HIP_SCORE_BM15:
            qt.kind = kBM15;
            if (!(ts.norm_const > 0.f)) rc = IRS_HIP_EINVAL;
            break;
          case IRS_HIP_SCORE_BM1: qt.kind = kBM1; break;
          case IRS_HIP_SCORE_TFIDF: qt.kind = kTfidf; break;
          case IRS_HIP_SCORE_TFIDF_NORM:
            qt.kind = !norms ? kTfidf
                      : legacy ? kTfidfLegacy
                               : (seg->dev.norm_width == 1 ? kTfidfTiny : kTfidfWide);
            break;
          default: rc = IRS_HIP_EINVAL;
        }
        if (rc != IRS_HIP_OK) break;
        // (BM25 family: a posting scores below its boost c0 whatever the segment holds; the
        // TF-IDF bound grows with the segment's largest frequency)
        same_bound = same_bound && (ts.kind == IRS_HIP_SCORE_BM25 || ts.kind == IRS_HIP_SCORE_BM15 ||
                                    ts.kind == IRS_HIP_SCORE_BM1);
        upper_all += double(ts.c0);
        // TermQuery::execute: no term state in this segment -> empty iterator
        // (term_query.cpp:41-43)
        if (qt.term == IRS_HIP_NO_TERM || seg->terms[qt.term].docs_count == 0) {
          absent = true;
          continue;
        }
        const DevTerm& t = seg->terms[qt.term];
        qt.pad1 = t.tf_bound;
        {
          // smallest score one posting of this term can have (tf = 1, longest doc)
          const double c0 = qt.c0, nc = qt.norm_const, nl = qt.norm_length;
          double smin = 0.0;
          switch (qt.kind) {
            case kBM1: smin = c0; break;
            case kBM15: smin = c0 - c0 / (1.0 + 1.0 / nc); break;
            case kBM25Tiny: smin = c0 - c0 / (1.0 + 1.0 / (nc + nl * 255.0)); break;
            case kBM25One: smin = c0 - c0 / (1.0 + 1.0 / (nc + nl)); break;
            case kTfidf: smin = c0; break;
            case kTfidfTiny: smin = c0 / std::sqrt(255.0); break;
            default: smin = 0.0;  // wide norms: unbounded below
          }
          min_score = std::min(min_score, smin);
          smins.push_back(smin);
        }
        const bool tfidf = qt.kind == kTfidf || qt.kind == kTfidfTiny || qt.kind == kTfidfWide ||
                           qt.kind == kTfidfLegacy;
        upper += tfidf ? double(qt.c0) * std::sqrt(double(t.tf_bound)) : double(qt.c0);
        b->postings += t.docs_count;
        b->alg_bytes += uint64_t(t.blocks_bytes) + t.tail_bytes;
        if (qt.kind == kBM25Tiny || qt.kind == kBM25Wide || qt.kind == kTfidfTiny ||
            qt.kind == kTfidfWide || qt.kind == kBM25Legacy || qt.kind == kTfidfLegacy)
          b->alg_bytes += uint64_t(t.docs_count) * seg->dev.norm_width;
        row.push_back(qt);
      }
      if (rc != IRS_HIP_OK) break;
      b->alg_bytes += 8ull * in.k;
      DevQuery& dq = b->queries[q];
      dq.k = in.k;
      dq.seg = q / nq_user;
      // How many of the (present) terms a doc must match.  Or: 1.  And: all, and one absent
      // term empties it (MakeScoreAdapters<true>, boolean_query.cpp:50-53).  MinMatch(m)
      // (MinMatchQuery::execute, boolean_query.cpp:212-247): m > #sub-queries or m > #present
      // -> empty; m == #present -> conjunction; m <= 1 -> disjunction; otherwise the
      // min-match block disjunction: every matching term scores, docs with < m matches drop.
      uint32_t need = 1;
      if (in.op == IRS_HIP_OP_AND) {
        need = absent ? 0xFFu : uint32_t(row.size());
      } else if (in.op == IRS_HIP_OP_MINMATCH) {
        // Or::prepare turns min_match_count == 0 into the all-docs filter
        // (boolean_filter.cpp:213): not a posting-list query, not on this path
        if (in.min_match == 0) {
          rc = IRS_HIP_EUNSUPPORTED;
          break;
        }
        const uint32_t m = in.min_match;
        need = (m > in.n_terms || m > row.size()) ? 0xFFu : m;
      }
      if (is_phrase) {
        // no phrase state for a segment lacking one of the terms (phrase_filter.cpp:254-258)
        need = absent ? 0xFFu : 1u;
        // the phrase's scorer is one stats blob: every entry must carry the same values
        for (const DevQTerm& qt : row)
          if (qt.kind != row[0].kind || qt.c0 != row[0].c0 ||
              qt.norm_const != row[0].norm_const || qt.norm_length != row[0].norm_length)
            rc = IRS_HIP_EINVAL;
        if (rc != IRS_HIP_OK) break;
      }
      if (need == 0xFFu) row.clear();
      // A doc that exists matches at least `need` terms: c matched postings score at least
      // c times the mean of the `need` smallest per-term minima — the score below which no
      // posting of a matching doc falls ON AVERAGE, which is what bounds the relative error of
      // a fixed-point sum that loses a constant per posting
      if (need > 1 && need != 0xFFu && !smins.empty() && !is_phrase) {
        std::sort(smins.begin(), smins.end());
        double sm = 0.0;
        for (uint32_t i = 0; i < need && i < smins.size(); ++i) sm += smins[i];
        min_score = sm / double(need);
      }
      // low byte of op: 0 = disjunction in doc tiles, 1 = doc tiles with per-doc match
      // counters (min-match), 2 = conjunction, block by block of its rarest term (conj.h)
      dq.op = 0;
      if (need > 1 && !row.empty() && !is_phrase) {
        if (need == row.size()) {
          // MakeConjunction sorts by cost (conjunction.hpp:450-453): the cheapest leads, and
          // the scores are summed in that order
          std::stable_sort(row.begin(), row.end(), [&](const DevQTerm& x, const DevQTerm& y) {
            return seg->terms[x.term].docs_count < seg->terms[y.term].docs_count;
          });
          dq.op = int32_t(2u | (need << 8));
        } else {
          dq.op = int32_t(1u | (need << 8));
          b->any_and = true;
        }
      }
      // The filter's ScoreMergeType (boolean_filter.hpp:39-43).  One sub-iterator: its score as
      // it is (MakeDisjunction :1422-1426, MakeConjunction :444).  kMin in a disjunction merges
      // with the 0 of every sub-iterator that is not on the doc (basic_disjunction,
      // disjunction.hpp:338-351) resp. with the zeroed score buffer (block_disjunction
      // :1308-1351): two sub-iterators -> min where both match, else 0; more (or the
      // min-match block disjunction) -> 0 for every doc.
      uint32_t merge = row.size() > 1 ? in.merge : uint32_t(IRS_HIP_MERGE_SUM);
      if (merge == IRS_HIP_MERGE_MIN && (dq.op & 0xFF) != 2) {
        if ((dq.op & 0xFF) == 0 && row.size() == 2) {
          dq.op |= int32_t(1u << 18);
          b->any_and = true;   // (the per-doc match counters tell "both")
        } else {
          for (DevQTerm& qt : row) qt.c0 = 0.f;
          upper = 0.0;
          min_score = 0.0;
          merge = IRS_HIP_MERGE_SUM;
        }
      }
      dq.op |= int32_t(merge << 16);
      if (!is_phrase) ((dq.op & 0xFF) == 2 ? b->all_conj_units : b->all_tile_units).push_back(q);
      // match counts in the low bits of a 32-bit accumulator (join.h COUNT) round every posting
      // to 16 fixed-point units (+-8): relative to any doc's score that is at most
      // 8 * upper / (2^29 * min_score) — allowed while it stays below 2e-6
      b->count_precise[q] = row.size() <= kJoinCountTerms && min_score > 0.0 && upper > 0.0 &&
                            upper / min_score <= 125.0;
      // table slots (kernels.h "table_kind"): one per distinct (kind, norm_const, norm_length)
      uint32_t n_caches = 0;
      float cnc[kMaxCaches], cnl[kMaxCaches];
      int32_t ckind[kMaxCaches];
      for (DevQTerm& qt : row) {
        if (qt.kind != kBM25Tiny && qt.kind != kBM25One && qt.kind != kBM15 &&
            qt.kind != kTfidf && qt.kind != kTfidfTiny)
          continue;
        uint32_t c = 0;
        for (; c < n_caches; ++c)
          if (ckind[c] == qt.kind && cnc[c] == qt.norm_const && cnl[c] == qt.norm_length) break;
        if (c == n_caches && n_caches < kMaxCaches) {
          ckind[c] = qt.kind;
          cnc[c] = qt.norm_const;
          cnl[c] = qt.norm_length;
          ++n_caches;
        }
        qt.cache_id = c < kMaxCaches ? c : kMaxCaches;
      }
      dq.n_caches = n_caches;
      dq.n_terms = uint32_t(row.size());
      dq.first_term = uint32_t(b->qterms.size());
      upper *= 1.0 + 1e-6;
      if (!row.empty() && upper == 0.0) upper = 1.0;   // every boost is 0: all scores are 0
      if (!row.empty() && !(upper > 0.0 && std::isfinite(upper))) {
        rc = IRS_HIP_EUNSUPPORTED;
        break;
      }
      dq.bin_scale = row.empty() ? 0.f : float(double(kBins) / upper);
      // (irs_hip_batch_set_comm) the bound every segment of the index computes alike
      b->group_upper[q] = (same_bound && !is_phrase && !row.empty() && upper_all > 0.0 &&
                           upper_all * (1.0 + 1e-6) >= upper && std::isfinite(upper_all))
                              ? upper_all * (1.0 + 1e-6)
                              : 0.0;
      // fixed-point accumulation: upper < 2^e.  32-bit accumulators (2^(30-e) units) lose at
      // most one unit per posting, i.e. <= upper / (2^29 * min_score) relative to any doc's
      // score: used only while that stays below 2e-6 for every query of the batch.
      int e = 0;
      if (!row.empty()) {
        (void)std::frexp(upper, &e);
        if (e < -60 || e > 60) {
          rc = IRS_HIP_EUNSUPPORTED;
          break;
        }
        if (!(min_score > 0.0) || upper / min_score > 1000.0) b->acc32 = false;
      }
      exps.push_back(e);
      b->qterms.insert(b->qterms.end(), row.begin(), row.end());
      b->jt = std::max(b->jt, dq.n_terms);
      b->k_max = std::max(b->k_max, in.k);
    }
    if (const char* env = std::getenv("IRS_HIP_ACC")) {  // tuning / test knob
      if (std::atoi(env) == 64) b->acc32 = false;
    }
    for (uint32_t q = 0; q < nq && rc == IRS_HIP_OK && q < exps.size(); ++q) {
      const int e = exps[q];
      b->queries[q].fx_mul = std::ldexp(1.f, (b->acc32 ? 30 : 29) - e);
      b->queries[q].fx_inv = std::ldexp(1.f, e - (b->acc32 ? 30 : 61));
    }
  } catch (...) {
    rc = IRS_HIP_ENOMEM;
  }
  if (rc == IRS_HIP_OK && b->phrase) {
    // k_phrase work: the lead term of a unit is its rarest one; one wavefront per
    // 128-posting block of it (+ one for its vint tail / single doc); records and start
    // blocks of the other terms written by k_conj_seek every run
    try {
      std::vector<uint32_t> lead_of(nq, 0);
      for (uint32_t u = 0; u < nq; ++u) {
        const DevQuery& dq = b->queries[u];
        if (!dq.n_terms) continue;
        const irs_hip_segment* sg = b->segs[dq.seg];
        uint32_t best = 0xFFFFFFFFu, items = 0;
        for (uint32_t j = 0; j < dq.n_terms; ++j) {
          const DevTerm& t = sg->terms[b->qterms[dq.first_term + j].term];
          if (t.docs_count < best) {
            best = t.docs_count;
            items = t.nblk + ((t.docs_count == 1 || t.tail_n) ? 1u : 0u);
            lead_of[u] = j;
          }
        }
        // (the lists of the pilot pass: same bookkeeping as for conjunctions)
        b->conj_units.push_back(u);
        b->conj_items.push_back(items);
      }
      std::vector<uint32_t> item_base(b->conj_units.size() + 1, 0), unit_items(nq, 0);
      uint64_t total = 0;
      for (size_t c = 0; c < b->conj_units.size(); ++c) {
        item_base[c] = uint32_t(total);
        unit_items[b->conj_units[c]] = uint32_t(total);
        total += b->conj_items[c];
      }
      item_base[b->conj_units.size()] = uint32_t(total);
      if (total > 0x7FFFFFFFull) {
        rc = IRS_HIP_EUNSUPPORTED;
      } else if (total) {
        b->conj_total_items = uint32_t(total);
        b->n_phrase_wgs = uint32_t((total + kPhraseWaves - 1) / kPhraseWaves);
        if (!b->d_conj_units.alloc(b->conj_units.size() * 4) ||
            !b->d_conj_items.alloc(b->conj_items.size() * 4) ||
            !b->d_conj_item_base.alloc(item_base.size() * 4) ||
            !b->d_conj_unit_items.alloc(unit_items.size() * 4) ||
            !b->d_lead_of.alloc(lead_of.size() * 4) ||
            !b->d_conj_seek.alloc((total + 2) * uint64_t(kMaxTerms) * 4) ||
            !b->d_conj_recs.alloc((total + 1) * sizeof(ConjItem)) ||
            !b->d_conj_item_hits.alloc((total + 1) * 4) ||
            !b->d_conj_hist.alloc(uint64_t(nq) * kBins * 4))
          rc = IRS_HIP_ENOMEM;
        else if (!b->up.copy(b->d_conj_units.p, b->conj_units.data(), b->conj_units.size() * 4) ||
                 !b->up.copy(b->d_conj_items.p, b->conj_items.data(), b->conj_items.size() * 4) ||
                 !b->up.copy(b->d_conj_item_base.p, item_base.data(), item_base.size() * 4) ||
                 !b->up.copy(b->d_conj_unit_items.p, unit_items.data(), unit_items.size() * 4) ||
                 !b->up.copy(b->d_lead_of.p, lead_of.data(), lead_of.size() * 4))
          rc = IRS_HIP_ENOMEM;
      }
    } catch (...) {
      rc = IRS_HIP_ENOMEM;
    }
  }
  // (the block-driven kernels read the norms of a lead block's docs from the posting-order copy)
  // — a permanent copy per segment, one byte per posting (irs_hip_segment_device_bytes counts
  // it): built the first time a conjunction / phrase whose scorer reads norms arrives
  if (rc == IRS_HIP_OK && (b->phrase || !b->all_conj_units.empty())) {
    bool wanted = false;
    for (const DevQTerm& qt : b->qterms) wanted = wanted || needs_norm(qt.kind);
    for (irs_hip_segment* sg : b->segs)
      if (wanted && rc == IRS_HIP_OK) rc = prepare_posting_norms(sg);
  }
  if (rc == IRS_HIP_OK) {
    if (b->jt == 0) b->jt = 1;
    if (b->qterms.empty()) b->qterms.push_back(DevQTerm{});
    std::vector<DevSegment> dsegs;
    for (irs_hip_segment* sg : b->segs) {
      // (another thread's batch may be adding the lazily built tables to `dev` right now)
      std::lock_guard<std::mutex> lock(sg->wand_mutex);
      dsegs.push_back(sg->dev);
    }
    if (!b->d_queries.alloc(b->queries.size() * sizeof(DevQuery)) ||
        !b->d_qterms.alloc(b->qterms.size() * sizeof(DevQTerm)) ||
        !b->d_segs.alloc(dsegs.size() * sizeof(DevSegment))) {
      rc = IRS_HIP_ENOMEM;
    } else if (!b->up.copy(b->d_segs.p, dsegs.data(), dsegs.size() * sizeof(DevSegment)) ||
               !b->up.copy(b->d_qterms.p, b->qterms.data(), b->qterms.size() * sizeof(DevQTerm))) {
      // (d_queries goes out from ensure_scratch, once the units' tile geometry is in)
      rc = IRS_HIP_ENOMEM;
    }
  }
  if (rc != IRS_HIP_OK) {
    delete b;
    return rc;
  }
  *out = b;
  return IRS_HIP_OK;
}

// Before a setter re-deals the units (buffers are reallocated, tables rewritten): whatever of the
// batch is still queued must be through — its last run AND a plan stage queued ahead.
static bool quiesce(irs_hip_batch* b) {
  bool ok = true;
  if (b->plan_pending) {
    ok = b->ev_planned_ready && rt::event_sync(b->ev_planned);
    b->plan_pending = false;
  }
  if (b->ran) ok = rt::sync(b->stream) && ok;
  return ok;
}

static int batch_configure_impl(irs_hip_batch* b, uint32_t tile_docs, uint32_t pilot_stride,
                            uint32_t cand_cap) {
  if (!b) return IRS_HIP_EINVAL;
  if (tile_docs && tile_docs != 4096 && tile_docs != 6144 && tile_docs != 8192 &&
      tile_docs != 12288)
    return IRS_HIP_EINVAL;
  if (cand_cap && cand_cap < b->k_max) return IRS_HIP_EINVAL;
  if (!rt::set_device(b->seg->device)) return IRS_HIP_EHIP;
  if (!quiesce(b)) return IRS_HIP_EHIP;
  if (!b->phrase) b->tile_asked = b->tile = tile_docs;  // phrase tiles are fixed; 0: by the units' needs
  if (pilot_stride) b->stride = pilot_stride;
  b->cand_cap = cand_cap;
  b->scratch_ready = false;
  b->planned = false;   // a plan queued ahead used the old geometry: run() plans inline
  return IRS_HIP_OK;
}

static int batch_set_path_impl(irs_hip_batch* b, int path) {
  if (!b || path < IRS_HIP_PATH_AUTO || path > IRS_HIP_PATH_JOINED) return IRS_HIP_EINVAL;
  if (!rt::set_device(b->seg->device)) return IRS_HIP_EHIP;
  if (!quiesce(b)) return IRS_HIP_EHIP;
  b->path_pref = path;
  b->scratch_ready = false;
  b->planned = false;   // (a plan queued ahead was made for the other path)
  return IRS_HIP_OK;
}

static int batch_set_shared_threshold_impl(irs_hip_batch* b, int enable) {
  if (!b) return IRS_HIP_EINVAL;
  if (!rt::set_device(b->seg->device)) return IRS_HIP_EHIP;
  if (!quiesce(b)) return IRS_HIP_EHIP;
  b->shared_threshold = enable != 0;
  b->scratch_ready = false;
  b->planned = false;
  return IRS_HIP_OK;
}

static int batch_set_comm_impl(irs_hip_batch* b, irs_hip_comm* comm) {
  if (!b) return IRS_HIP_EINVAL;
  if (!rt::set_device(b->seg->device)) return IRS_HIP_EHIP;
  if (!quiesce(b)) return IRS_HIP_EHIP;
  if (comm && !b->d_agree.p && !b->d_agree.alloc(64)) return IRS_HIP_ENOMEM;
  b->comm = comm;
  b->scratch_ready = false;
  b->planned = false;
  return IRS_HIP_OK;
}

static int batch_set_async_impl(irs_hip_batch* b, int enable) {
  if (!b) return IRS_HIP_EINVAL;
  b->async_pref = enable < 0 ? -1 : (enable ? 1 : 0);
  return IRS_HIP_OK;
}

static int batch_set_paired_tiles_impl(irs_hip_batch* b, int enable) {
  if (!b) return IRS_HIP_EINVAL;
  if (!rt::set_device(b->seg->device)) return IRS_HIP_EHIP;
  if (!quiesce(b)) return IRS_HIP_EHIP;
  b->pairs_allowed = enable != 0;
  b->pairs_forced = enable == 2;
  return IRS_HIP_OK;
}
static int batch_paired_tiles_impl(irs_hip_batch* b, int* used) {
  if (!b || !used) return IRS_HIP_EINVAL;
  *used = (b->joined && b->pairs_used) ? 1 : 0;
  return IRS_HIP_OK;
}

static int batch_path_impl(irs_hip_batch* b, int* path) {
  if (!b || !path) return IRS_HIP_EINVAL;
  *path = b->joined ? IRS_HIP_PATH_JOINED : IRS_HIP_PATH_ITEMS;
  return IRS_HIP_OK;
}

static int batch_set_wand_impl(irs_hip_batch* b, int enable) {
  if (!b) return IRS_HIP_EINVAL;
  if (b->ran) return IRS_HIP_EINVAL;   // before the first run: the segment records are uploaded once
  if (!rt::set_device(b->seg->device)) return IRS_HIP_EHIP;
  if (!quiesce(b)) return IRS_HIP_EHIP;
  b->wand = enable != 0;
  // a plan queued ahead (irs_hip_batch_plan) was made without the tile bounds: run() re-plans
  b->planned = false;
  b->scratch_ready = false;   // (which path the batch takes depends on it)
  if (!b->wand) return IRS_HIP_OK;
  std::vector<DevSegment> dsegs;
  for (irs_hip_segment* sg : b->segs) {
    if (const int rc = prepare_blockmax(sg)) return rc;
    dsegs.push_back(sg->dev);
  }
  if (!b->up.copy(b->d_segs.p, dsegs.data(), dsegs.size() * sizeof(DevSegment)))
    return IRS_HIP_ENOMEM;
  return IRS_HIP_OK;
}

static int batch_set_min_scores_impl(irs_hip_batch* b, const float* min_scores) {
  if (!b) return IRS_HIP_EINVAL;
  if (!rt::set_device(b->seg->device)) return IRS_HIP_EHIP;
  if (!min_scores) {
    b->has_min = false;
    return IRS_HIP_OK;
  }
  const uint32_t nq_user = b->nq / uint32_t(b->segs.size());
  std::vector<float> mins(b->nq, 0.f);
  for (uint32_t u = 0; u < b->nq; ++u) {
    const float m = min_scores[u % nq_user];
    if (!(m >= 0.f)) return IRS_HIP_EINVAL;   // (also NaN)
    mins[u] = m;
  }
  if (!quiesce(b)) return IRS_HIP_EHIP;
  if (!b->d_min_bin.alloc(mins.size() * 4) || !b->d_min_score.alloc(mins.size() * 4))
    return IRS_HIP_ENOMEM;
  b->min_scores.swap(mins);
  b->has_min = true;
  b->min_dirty = true;   // (the bins go out with the next run: stage_min_bins)
  return IRS_HIP_OK;
}

// The caller's min scores as score bins, in the scale the units bin with NOW: ensure_scratch may
// have re-scaled a unit (build_groups: one bound for a query on every rank), so this runs behind it.
static bool stage_min_bins(irs_hip_batch* b) {
  if (!b->has_min || !b->min_dirty) return true;
  std::vector<uint32_t> bins(b->nq, 0u);
  for (uint32_t u = 0; u < b->nq; ++u) {
    // the bin score_bin() puts a score of m into: docs at or above m land in it or higher
    const float x = std::fmin(b->min_scores[u] * b->queries[u].bin_scale, float(kBins - 1));
    bins[u] = uint32_t(x);
  }
  if (!b->up.copy(b->d_min_bin.p, bins.data(), bins.size() * 4) ||
      !b->up.copy(b->d_min_score.p, b->min_scores.data(), b->min_scores.size() * 4))
    return false;
  b->min_dirty = false;
  return true;
}

static int term_blockmax_impl(irs_hip_segment* seg, uint32_t term, uint32_t* max_freqs,
                              uint32_t* min_norms, uint32_t cap, uint32_t* count) {
  if (!seg || !count || term >= seg->dev.num_terms) return IRS_HIP_EINVAL;
  if (!rt::set_device(seg->device)) return IRS_HIP_EHIP;
  const DevTerm& t = seg->terms[term];
  *count = t.nblk;
  if (!t.nblk) return IRS_HIP_OK;
  if (cap < t.nblk || !max_freqs || !min_norms) return IRS_HIP_EINVAL;
  if (const int rc = prepare_blockmax(seg)) return rc;
  if (!rt::d2h(max_freqs, seg->d_blk_maxf.as<uint32_t>() + t.dir_off, size_t(t.nblk) * 4, nullptr) ||
      !rt::d2h(min_norms, seg->d_blk_minn.as<uint32_t>() + t.dir_off, size_t(t.nblk) * 4, nullptr) ||
      !rt::sync(nullptr))
    return IRS_HIP_EHIP;
  return IRS_HIP_OK;
}

static int segment_wand_source_impl(irs_hip_segment* seg, uint64_t* from_index, uint64_t* total) {
  if (!seg) return IRS_HIP_EINVAL;
  if (!rt::set_device(seg->device)) return IRS_HIP_EHIP;
  if (const int rc = prepare_blockmax(seg)) return rc;
  if (from_index) *from_index = seg->wand_from_index;
  if (total) *total = seg->total_blocks;
  return IRS_HIP_OK;
}

static int comm_unique_id_impl(uint8_t* id) {
  if (!id) return IRS_HIP_EINVAL;
  return rt::comm::unique_id(id) ? IRS_HIP_OK : IRS_HIP_EHIP;
}

static int comm_init_rank_impl(int32_t device, const uint8_t* id, int32_t n_ranks, int32_t rank,
                               irs_hip_comm** out) {
  if (!id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return IRS_HIP_EINVAL;
  *out = nullptr;
  if (!device_usable(device)) return IRS_HIP_EHIP;
  irs_hip_comm* c = new (std::nothrow) irs_hip_comm;
  if (!c) return IRS_HIP_ENOMEM;
  c->device = device;
  c->n_ranks = n_ranks;
  c->rank = rank;
  if (!rt::comm::init_rank(&c->h, n_ranks, id, rank)) {
    delete c;
    return IRS_HIP_EHIP;
  }
  *out = c;
  return IRS_HIP_OK;
}

static int topk_allgather_impl(irs_hip_comm* c, const void* d_send, void* d_recv,
                               uint64_t bytes_per_rank, void* stream) {
  if (!c || !d_send || !d_recv || !bytes_per_rank) return IRS_HIP_EINVAL;
  if (!rt::set_device(c->device)) return IRS_HIP_EHIP;
  return rt::comm::all_gather(c->h, d_send, d_recv, bytes_per_rank,
                              static_cast<rt::stream_t>(stream))
             ? IRS_HIP_OK : IRS_HIP_EHIP;
}

static int batch_touched_impl(irs_hip_batch* b, uint64_t* doc_bytes, uint64_t* positions) {
  if (!b || !b->ran || !b->count_touched) return IRS_HIP_EINVAL;
  if (!rt::set_device(b->seg->device)) return IRS_HIP_EHIP;
  std::vector<uint64_t> v(size_t(b->nq) * 2);
  if (!rt::d2h(v.data(), b->d_touched.p, v.size() * 8, b->stream) || !rt::sync(b->stream))
    return IRS_HIP_EHIP;
  uint64_t bytes = 0, pos = 0;
  for (uint32_t u = 0; u < b->nq; ++u) {
    bytes += v[2 * u];
    pos += v[2 * u + 1];
  }
  if (doc_bytes) *doc_bytes = bytes;
  if (positions) *positions = pos;
  return IRS_HIP_OK;
}

static int batch_profile_impl(irs_hip_batch* b, int enable) {
  if (!b) return IRS_HIP_EINVAL;
  if (!rt::set_device(b->seg->device)) return IRS_HIP_EHIP;
  if ((enable & 1) && !b->events_ready) {
    for (auto& e : b->ev)
      if (!rt::event_create(&e)) return IRS_HIP_EHIP;
    b->events_ready = true;
  }
  b->profile = (enable & 1) != 0;
  b->count_touched = (enable & 2) != 0;
  return IRS_HIP_OK;
}

// The planning stage of a run: tile -> first block tables, per-term records, the work items of
// the tile kernels.  A pure function of (batch, segments): it touches nothing a run of ANOTHER
// batch reads, so a caller may queue it ahead on a second stream (irs_hip_batch_plan).
static bool plan_stage(irs_hip_batch* b, rt::stream_t st) {
  auto mark = [&](int i) { return !b->profile || rt::event_record(b->ev[i], st); };
  bool ok = mark(2 * IRS_HIP_K_PLAN);
  // (a joined batch without conjunctions needs none of k_plan's tables)
  if (ok && (b->phrase || !b->tile_units.empty() || !b->conj_units.empty())) {
    RT_LAUNCH(k_plan, b->nq * b->jt, kThreads, 0, st, b->d_segs.as<DevSegment>(),
              b->d_queries.as<DevQuery>(), b->d_qterms.as<DevQTerm>(), b->jt, b->tile,
              b->d_first.as<uint32_t>(), b->d_tails.as<DevTail>());
    ok = rt::last_error_ok();
  }
  if (ok && b->joined) ok = launch_join(b, st);
  if (ok && !b->phrase && !b->tile_units.empty()) ok = launch_items(b, st);
  return ok && mark(2 * IRS_HIP_K_PLAN + 1);
}

static int batch_plan_impl(irs_hip_batch* b, void* stream) {
  if (!b) return IRS_HIP_EINVAL;
  if (!rt::set_device(b->seg->device)) return IRS_HIP_EHIP;
  if (!ensure_scratch(b) || !stage_min_bins(b)) return IRS_HIP_ENOMEM;
  rt::stream_t st = static_cast<rt::stream_t>(stream);
  bool ok = true;
  if (!b->ev_planned_ready) ok = b->ev_planned_ready = rt::event_create(&b->ev_planned);
  // (the tables are rewritten: the batch's own previous run must be through with them)
  if (ok && b->ev_done_ready && b->ran) ok = rt::stream_wait(st, b->ev_done);
  // (a plan queued earlier and never consumed may still run on ANOTHER stream)
  if (ok && b->plan_pending) ok = rt::stream_wait(st, b->ev_planned);
  ok = ok && b->up.flush(st) && plan_stage(b, st) && rt::event_record(b->ev_planned, st);
  b->planned = ok;
  b->plan_pending = b->plan_pending || b->ev_planned_ready;   // (whatever got queued)
  return ok ? IRS_HIP_OK : IRS_HIP_EHIP;
}

static int run_impl(irs_hip_batch* b, rt::stream_t st) {
  HostTrace trace("batch_run (scratch + uploads + launches queued)");
  if (!ensure_scratch(b) || !stage_min_bins(b)) return IRS_HIP_ENOMEM;
  b->stream = st;
  // a copy of the PREVIOUS run's results to host memory may still be reading d_out / d_hits on the
  // download stream: this run rewrites them only behind it (and those host results are stale then:
  // irs_hip_batch_host_results refuses them until the next irs_hip_batch_results_to_host)
  if (b->host_pending) {
    if (!rt::stream_wait(st, b->ev_host)) return IRS_HIP_EHIP;
    b->host_pending = false;
  }
  const bool simd = b->seg->dev.layout == kSimd4;
  auto mark = [&](int i) { return !b->profile || rt::event_record(b->ev[i], st); };
  // the batch's tables (built in page-locked memory since create) go out: before the batch's
  // first run on the device's copy stream (nothing on the GPU reads or writes these buffers yet),
  // afterwards in the run's own stream order (a running kernel may still read what they replace)
  bool ok = true;
  // a plan stage that is still queued (used below, or made stale by a setter) reads and writes
  // the tables this run uploads and rewrites: both streams get behind it
  const bool after_plan = b->plan_pending && b->ev_planned_ready;
  if (after_plan) ok = rt::stream_wait(st, b->ev_planned);
  rt::stream_t up_st = (!b->ran && !b->up.pending.empty()) ? upload_stream(b->seg->device) : nullptr;
  if (up_st && after_plan) ok = ok && rt::stream_wait(up_st, b->ev_planned);
  b->plan_pending = false;
  if (up_st) {
    if (!b->ev_up_ready) ok = b->ev_up_ready = rt::event_create(&b->ev_up);
    ok = ok && b->up.flush(up_st) && rt::event_record(b->ev_up, up_st) && rt::stream_wait(st, b->ev_up);
  } else {
    ok = b->up.flush(st);
  }
  if (ok && b->joined && !b->slack_zeroed) {
    // (the slack behind the last stream is only ever read by masked-off look-ahead: zero it once)
    ok = rt::dmemset(b->d_entries.as<uint32_t>() + b->join_entries, 0, kJoinSlack * 4, st);
    b->slack_zeroed = ok;
  }
  ok = ok && rt::dmemset(b->d_zeroed.p, 0, b->d_zeroed.n, st);   // (ensure_scratch: six tables)
  // 1. plan (already queued by irs_hip_batch_plan: wait for it instead)
  const bool tiles = !b->phrase && !b->tile_units.empty();
  if (b->planned) {
    ok = ok && rt::stream_wait(st, b->ev_planned);
    b->planned = false;
  } else {
    ok = ok && plan_stage(b, st);
  }
  // 2. pilot: per-query score-bin threshold (phrase batches have none: few docs match)
  ok = ok && mark(2 * IRS_HIP_K_PILOT);
  if (b->n_groups) ok = ok && rt::dmemset(b->d_group_hist.p, 0, b->d_group_hist.n, st);
  if (b->joined) ok = ok && launch_join_pilot(b, st);
  ok = ok && launch_group_threshold(b, st);
  if (tiles)
    ok = ok && (simd ? launch_pilot_acc<kSimd4>(b, st) : launch_pilot_acc<kScalar>(b, st));
  ok = ok && mark(2 * IRS_HIP_K_PILOT + 1);
  // 3. score every tile / every lead block
  ok = ok && mark(2 * IRS_HIP_K_SCORE);
  if (b->phrase)
    ok = ok && (simd ? launch_phrase_terms<kSimd4>(b, st) : launch_phrase_terms<kScalar>(b, st));
  else if (tiles)
    ok = ok && (simd ? launch_score_acc<kSimd4>(b, st) : launch_score_acc<kScalar>(b, st));
  if (b->joined) ok = ok && launch_join_score(b, st);
  if (!b->phrase) ok = ok && (simd ? launch_conj<kSimd4>(b, st) : launch_conj<kScalar>(b, st));
  ok = ok && mark(2 * IRS_HIP_K_SCORE + 1);
  // 4. exact top-k
  ok = ok && mark(2 * IRS_HIP_K_SELECT);
  if (ok) {
    uint32_t sort_cap = 64;
    while (sort_cap < b->k_max) sort_cap <<= 1;
    const uint32_t stage_cap = std::min<uint32_t>(b->cand_cap, kSelectStage);
    const size_t smem = size_t(sort_cap + stage_cap) * sizeof(uint64_t);
    ok = big_smem(k_select, smem);
    if (ok) {
      RT_LAUNCH(k_select, b->nq, kThreads, smem, st, b->d_queries.as<DevQuery>(),
                b->d_cands.as<uint64_t>(), b->cand_cap, b->d_cand_count.as<uint32_t>(),
                b->d_hits.as<unsigned long long>(), b->d_out.as<Hit>(), b->k_max,
                b->d_out_count.as<uint32_t>(), b->d_status.as<uint32_t>(), stage_cap, sort_cap,
                b->d_bstar.as<uint32_t>(), min_bins(b), b->d_pruned.as<uint32_t>(),
                b->has_min ? b->d_min_score.as<float>() : static_cast<const float*>(nullptr),
                b->n_groups ? b->d_group_of.as<uint32_t>() : static_cast<const uint32_t*>(nullptr));
      if (b->n_groups) {
        RT_LAUNCH(k_group_sums, (b->n_groups + 63u) / 64u, 64, 0, st, b->d_queries.as<DevQuery>(),
                  b->d_group_members.as<uint32_t>(), uint32_t(b->segs.size()), b->n_groups,
                  b->d_out_count.as<uint32_t>(), b->d_hits.as<unsigned long long>(),
                  b->d_bstar.as<uint32_t>(), min_bins(b), b->d_status.as<uint32_t>(),
                  b->d_group_sums.as<uint32_t>());
        ok = rt::last_error_ok();
        if (ok && b->comm && !b->phrase)
          ok = rt::comm::all_reduce_u32(b->comm->h, b->d_group_sums.p,
                                        size_t(b->n_groups) * kGroupSumWords + 2, st);
        if (ok)
          RT_LAUNCH(k_group_verdict, (b->n_groups + 63u) / 64u, 64, 0, st, b->d_queries.as<DevQuery>(),
                    b->n_groups, b->d_group_sums.as<uint32_t>(), b->d_status.as<uint32_t>());
      }
      ok = ok && rt::last_error_ok();
    }
  }
  ok = ok && mark(2 * IRS_HIP_K_SELECT + 1);
  // the status word follows the kernels into page-locked memory; the event marks this run
  if (ok && !b->h_status) {
    ok = b->h_status_buf.alloc(64);
    b->h_status = b->h_status_buf.as<uint32_t>();
  }
  if (ok && !b->ev_done_ready) ok = b->ev_done_ready = rt::event_create(&b->ev_done);
  ok = ok && rt::d2h(b->h_status, b->d_status.p, 4, st) && rt::event_record(b->ev_done, st);
  b->ran = true;
  return ok ? IRS_HIP_OK : IRS_HIP_EHIP;
}

static int batch_run_impl(irs_hip_batch* b, void* stream) {
  if (!b) return IRS_HIP_EINVAL;
  if (!rt::set_device(b->seg->device)) return IRS_HIP_EHIP;
  rt::stream_t st = static_cast<rt::stream_t>(stream);
  // (what a queued run returns is reported by the next call on the batch)
  if (worker::wanted(b) && worker::submit(b, [b, st] { return run_impl(b, st); })) return IRS_HIP_OK;
  return run_impl(b, st);
}

// k_select flagged a problem with the candidates of some query.
//   underflow: the estimated threshold was too high.  Re-run with the sound one
//     (the batch stays in sound mode from then on).
//   overflow: the candidate buffer was too small — the pilot sample was not
//     representative, or many docs tie at the k-th score bin.  Re-run exactly:
//     first with a full histogram pass (stride 1), then with the buffer grown to
//     the largest candidate count seen.
// Results are never silently truncated.
// Recovery of a batch whose threshold spans ranks (irs_hip_batch_set_comm) is decided by ALL ranks
// or by none.  Every rank enters recover() together — the status bits are all-reduced inside the
// run — but whether a rank CAN re-run is its own business (memory for a larger candidate buffer,
// an overflow it cannot afford): one that returned by itself would leave the others waiting in the
// re-run's collectives for ever.  So each such exit is a vote: the ranks all-reduce "I cannot", and
// a re-run happens only when the sum is zero.  local_rc: what this rank would return by itself
// (IRS_HIP_OK: it can go on).  Returns IRS_HIP_OK when every rank can, this rank's own error when
// it cannot, IRS_HIP_EPEER when only others cannot.  Without a communicator: local_rc.
static int all_ranks_can(irs_hip_batch* b, int local_rc) {
  if (!b->comm || b->phrase) return local_rc;
  uint32_t vote = local_rc == IRS_HIP_OK ? 0u : 1u;
  if (!b->d_agree.p || !rt::h2d(b->d_agree.p, &vote, 4, b->stream) ||
      !rt::comm::all_reduce_u32(b->comm->h, b->d_agree.p, 1, b->stream) ||
      !rt::d2h(&vote, b->d_agree.p, 4, b->stream) || !rt::sync(b->stream))
    return local_rc != IRS_HIP_OK ? local_rc : IRS_HIP_EHIP;
  if (local_rc != IRS_HIP_OK) return local_rc;
  return vote ? IRS_HIP_EPEER : IRS_HIP_OK;
}

static int recover_overflow(irs_hip_batch* b);
static int recover_now(irs_hip_batch* b, uint32_t status);
// The re-run of a batch whose threshold spans ranks issues collectives (the vote, the run's two
// all-reduces): it takes its turn in the device's worker queue BEHIND the runs the caller has
// submitted since — on every rank alike, because every rank makes the same calls in the same order
// and reaches the same verdict — instead of racing them from the caller's thread.
static int recover(irs_hip_batch* b, uint32_t status) {
  if (b->comm && !b->phrase && worker::wanted(b) &&
      worker::submit(b, [b, status] { return recover_now(b, status); })) {
    std::unique_lock<std::mutex> lock(b->am);
    b->acv.wait(lock, [&] { return !b->async_pending; });
    const int rc = b->async_rc;
    b->async_rc = IRS_HIP_OK;
    return rc;
  }
  return recover_now(b, status);
}
static int recover_now(irs_hip_batch* b, uint32_t status) {
  ++b->reruns;
  if (std::getenv("IRS_HIP_TRACE")) {   // which units made the batch run again
    std::vector<uint32_t> cc(b->nq), oc(b->nq);
    std::vector<unsigned long long> hh(b->nq);
    if (rt::d2h(cc.data(), b->d_cand_count.p, size_t(b->nq) * 4, b->stream) &&
        rt::d2h(oc.data(), b->d_out_count.p, size_t(b->nq) * 4, b->stream) &&
        rt::d2h(hh.data(), b->d_hits.p, size_t(b->nq) * 8, b->stream) && rt::sync(b->stream)) {
      uint32_t under = 0, over = 0, first_u = ~0u, first_o = ~0u;
      for (uint32_t u = 0; u < b->nq; ++u) {
        if (cc[u] > b->cand_cap) { ++over; if (first_o == ~0u) first_o = u; }
        if (oc[u] < b->queries[u].k && hh[u] > oc[u]) { ++under; if (first_u == ~0u) first_u = u; }
      }
      std::fprintf(stderr, "[irs_hip] re-run: status %u, %u units short of k (first %u: listed %u of %llu matches, "
                   "%u candidates), %u over the candidate cap %u (first %u: %u)\n", status, under, first_u,
                   first_u != ~0u ? oc[first_u] : 0u, first_u != ~0u ? hh[first_u] : 0ull,
                   first_u != ~0u ? cc[first_u] : 0u, over, b->cand_cap, first_o, first_o != ~0u ? cc[first_o] : 0u);
    }
  }
  if (status & kStatusUnderflow) {
    b->estimate = false;
    // the sound threshold admits about k * (pilot stride) candidates per unit: a denser pilot
    // keeps the candidate buffer of a large batch (units x cap x 8 bytes) in bounds
    b->stride_eff = std::min<uint32_t>(b->stride_eff, 16);
    const uint32_t cap = default_cand_cap(b);  // the sound threshold admits more candidates
    int can = IRS_HIP_OK;
    if (cap > b->cand_cap) {
      if (b->d_cands.alloc(uint64_t(b->nq) * cap * sizeof(uint64_t))) b->cand_cap = cap;
      else can = IRS_HIP_ENOMEM;
    }
    if (const int all = all_ranks_can(b, can)) return all;
    const int rc = run_impl(b, b->stream);
    if (rc != IRS_HIP_OK) return rc;
    if (!rt::d2h(&status, b->d_status.p, 4, b->stream) || !rt::sync(b->stream))
      return IRS_HIP_EHIP;
    if (status & kStatusUnderflow) return IRS_HIP_EHIP;  // cannot happen with a sound threshold
  }
  return (status & kStatusOverflow) ? recover_overflow(b) : IRS_HIP_OK;
}

static int recover_overflow(irs_hip_batch* b) {
  constexpr uint64_t kMaxCandBytes = 16ull << 30;
  for (int attempt = 0; attempt < 3; ++attempt) {
    // the failed run counted every candidate it met, also those it could not store: with the
    // same threshold a buffer of that size holds them all.  Growing it is the cheap way out;
    // only when that is not affordable, a full histogram pass (stride 1: the tightest sound
    // threshold) comes first — it costs as much as the scoring pass itself.
    std::vector<uint32_t> cc(b->nq);
    if (!rt::d2h(cc.data(), b->d_cand_count.p, size_t(b->nq) * 4, b->stream) ||
        !rt::sync(b->stream))
      return IRS_HIP_EHIP;
    const uint64_t need = uint64_t(*std::max_element(cc.begin(), cc.end())) + 1024;
    const bool affordable = need * b->nq * sizeof(uint64_t) <= kMaxCandBytes;
    if (need - 1024 > b->cand_cap && need <= 262144) {
      for (irs_hip_segment* sg : b->segs) {   // (what later batches on these segments start with)
        uint32_t seen = sg->cand_cap_hint.load();
        while (seen < need && !sg->cand_cap_hint.compare_exchange_weak(seen, uint32_t(need))) {}
      }
    }
    int can = IRS_HIP_OK;
    if (b->comm && need - 1024 <= b->cand_cap) {
      // (the overflow is another rank's: this one only takes part in the re-run's collectives)
    } else if (need > b->cand_cap && affordable) {
      if (b->d_cands.alloc(need * b->nq * sizeof(uint64_t))) b->cand_cap = uint32_t(need);
      else can = IRS_HIP_ENOMEM;
    } else if (b->stride_eff != 1) {
      b->stride_eff = 1;
    } else {
      can = IRS_HIP_EOVERFLOW;
    }
    if (const int all = all_ranks_can(b, can)) return all;
    int rc = run_impl(b, b->stream);
    if (rc != IRS_HIP_OK) return rc;
    uint32_t status = 0;
    if (!rt::d2h(&status, b->d_status.p, 4, b->stream) || !rt::sync(b->stream))
      return IRS_HIP_EHIP;
    if (!(status & kStatusOverflow)) return IRS_HIP_OK;
  }
  return IRS_HIP_EOVERFLOW;
}

static int batch_timings_impl(irs_hip_batch* b, float ms[IRS_HIP_K_COUNT]) {
  if (!b || !ms || !b->profile || !b->ran) return IRS_HIP_EINVAL;
  // (the batch's own last run: later work on the stream is not waited for)
  if (!rt::set_device(b->seg->device) || !b->ev_done_ready || !rt::event_sync(b->ev_done))
    return IRS_HIP_EHIP;
  for (int i = 0; i < IRS_HIP_K_COUNT; ++i)
    if (!rt::event_elapsed(&ms[i], b->ev[2 * i], b->ev[2 * i + 1])) return IRS_HIP_EHIP;
  return IRS_HIP_OK;
}

static int batch_reruns_impl(irs_hip_batch* b, uint32_t* count) {
  if (!b || !count) return IRS_HIP_EINVAL;
  *count = b->reruns;
  return IRS_HIP_OK;
}

static int batch_work_impl(irs_hip_batch* b, uint64_t* algorithmic_bytes, uint64_t* postings) {
  if (!b) return IRS_HIP_EINVAL;
  if (algorithmic_bytes) *algorithmic_bytes = b->alg_bytes;
  if (postings) *postings = b->postings;
  return IRS_HIP_OK;
}

static int batch_results_impl(irs_hip_batch* b, irs_hip_hit* hits, uint32_t k_stride,
                          uint32_t* counts, uint64_t* total_hits) {
  if (!b || !hits || !counts || !b->ran || k_stride < b->k_max) return IRS_HIP_EINVAL;
  if (!rt::set_device(b->seg->device)) return IRS_HIP_EHIP;
  uint32_t status = 0;
  // hits land in a page-locked buffer owned by the batch (a pageable destination would
  // be staged by the runtime at a fraction of the PCIe rate), then go to the caller's layout
  const size_t hit_bytes = size_t(b->nq) * b->k_max * sizeof(Hit);
  if (b->h_pin.n < hit_bytes && !b->h_pin.alloc(hit_bytes)) return IRS_HIP_ENOMEM;
  const Hit* tmp = b->h_pin.as<Hit>();
  if (!rt::d2h(&status, b->d_status.p, 4, b->stream) || !rt::sync(b->stream))
    return IRS_HIP_EHIP;
  if (status & (kStatusOverflow | kStatusUnderflow)) {
    const int rc = recover(b, status);
    if (rc != IRS_HIP_OK) return rc;
    status = 0;
  }
  if (!rt::d2h(b->h_pin.p, b->d_out.p, hit_bytes, b->stream) ||
      !rt::d2h(counts, b->d_out_count.p, size_t(b->nq) * 4, b->stream) ||
      (total_hits && !rt::d2h(total_hits, b->d_hits.p, size_t(b->nq) * 8, b->stream)) ||
      !rt::sync(b->stream))
    return IRS_HIP_EHIP;
  if (status & kStatusOverflow) return IRS_HIP_EOVERFLOW;
  for (uint32_t q = 0; q < b->nq; ++q) {
    for (uint32_t i = 0; i < counts[q]; ++i) {
      const Hit& h = tmp[size_t(q) * b->k_max + i];
      hits[size_t(q) * k_stride + i].score = h.score;
      hits[size_t(q) * k_stride + i].doc = h.doc;
    }
  }
  return IRS_HIP_OK;
}

// Waits for the batch, reads its status word and re-executes it when the candidate
// buffer or the threshold estimate fell short.  Afterwards d_out / d_out_count hold the
// exact top-k.
// Waits for the batch's own last run (its event: work the caller queued behind it on the same
// stream — the next batch's kernels, say — keeps running) and reads the status it left.
static int verify_run(irs_hip_batch* b) {
  if (!b->ev_done_ready || !b->h_status || !rt::event_sync(b->ev_done)) return IRS_HIP_EHIP;
  const uint32_t status = *b->h_status;
  if (status & (kStatusOverflow | kStatusUnderflow)) {
    const int rc = recover(b, status);
    if (rc != IRS_HIP_OK) return rc;
    if (!rt::sync(b->stream)) return IRS_HIP_EHIP;
  }
  return IRS_HIP_OK;
}

static int batch_device_results_impl(irs_hip_batch* b, void** d_hits, void** d_counts,
                                 uint32_t* k_max) {
  if (!b || !b->ran) return IRS_HIP_EINVAL;
  if (!rt::set_device(b->seg->device)) return IRS_HIP_EHIP;
  const int rc = verify_run(b);
  if (rc != IRS_HIP_OK) return rc;
  if (d_hits) *d_hits = b->d_out.p;
  if (d_counts) *d_counts = b->d_out_count.p;
  if (k_max) *k_max = b->k_max;
  return IRS_HIP_OK;
}

static int batch_results_to_device_impl(irs_hip_batch* b, void* d_hits, void* d_counts,
                                    void* stream) {
  if (!b || !b->ran || !d_hits || !d_counts) return IRS_HIP_EINVAL;
  if (!rt::set_device(b->seg->device)) return IRS_HIP_EHIP;
  rt::stream_t st = static_cast<rt::stream_t>(stream);
  const int rc = verify_run(b);
  if (rc != IRS_HIP_OK) return rc;
  if (!rt::d2d(d_hits, b->d_out.p, size_t(b->nq) * b->k_max * sizeof(Hit), st) ||
      !rt::d2d(d_counts, b->d_out_count.p, size_t(b->nq) * 4, st))
    return IRS_HIP_EHIP;
  // (destroy must not hand d_out back to the pool while these copies are queued)
  if (!b->ev_used_ready) b->ev_used_ready = rt::event_create(&b->ev_used);
  if (!b->ev_used_ready || !rt::event_record(b->ev_used, st)) return IRS_HIP_EHIP;
  b->ev_used_pending = true;
  return IRS_HIP_OK;
}

// The checked results on their way to page-locked host memory, asynchronously: where the
// reference's harness ends (index-search.cpp:782-807: the sorted (score, doc) pairs of every task
// in host memory).  The copy is queued on `stream` (null: the device's download stream) behind the
// batch's own last run — NOT behind whatever the caller has queued since, so the results of batch i
// cross PCIe while the kernels of batch i + 1 run.
static int batch_results_to_host_impl(irs_hip_batch* b, void* stream) {
  if (!b || !b->ran) return IRS_HIP_EINVAL;
  if (!rt::set_device(b->seg->device)) return IRS_HIP_EHIP;
  const int rc = verify_run(b);
  if (rc != IRS_HIP_OK) return rc;
  const size_t hit_bytes = size_t(b->nq) * b->k_max * sizeof(Hit);
  const size_t cnt_off = (hit_bytes + 63) & ~size_t(63);
  const size_t tot_off = cnt_off + ((size_t(b->nq) * 4 + 63) & ~size_t(63));
  const size_t total = tot_off + size_t(b->nq) * 8;
  if (b->host_pending && !rt::event_sync(b->ev_host)) return IRS_HIP_EHIP;   // (the previous copy)
  b->host_pending = false;
  if (b->h_res.n < total && !b->h_res.alloc(total)) return IRS_HIP_ENOMEM;
  rt::stream_t st = stream ? static_cast<rt::stream_t>(stream) : download_stream(b->seg->device);
  if (!st) st = b->stream;
  if (!b->ev_host_ready) b->ev_host_ready = rt::event_create(&b->ev_host);
  uint8_t* h = b->h_res.as<uint8_t>();
  if (!b->ev_host_ready || !rt::stream_wait(st, b->ev_done) ||
      !rt::d2h(h, b->d_out.p, hit_bytes, st) ||
      !rt::d2h(h + cnt_off, b->d_out_count.p, size_t(b->nq) * 4, st) ||
      !rt::d2h(h + tot_off, b->d_hits.p, size_t(b->nq) * 8, st) ||
      !rt::event_record(b->ev_host, st))
    return IRS_HIP_EHIP;
  b->host_pending = true;
  return IRS_HIP_OK;
}

static int batch_host_results_impl(irs_hip_batch* b, const irs_hip_hit** hits, uint32_t* k_stride,
                                   const uint32_t** counts, const uint64_t** total_hits) {
  if (!b || !b->host_pending) return IRS_HIP_EINVAL;
  if (!rt::set_device(b->seg->device)) return IRS_HIP_EHIP;
  if (!rt::event_sync(b->ev_host)) return IRS_HIP_EHIP;
  const size_t hit_bytes = size_t(b->nq) * b->k_max * sizeof(Hit);
  const size_t cnt_off = (hit_bytes + 63) & ~size_t(63);
  const size_t tot_off = cnt_off + ((size_t(b->nq) * 4 + 63) & ~size_t(63));
  const uint8_t* h = b->h_res.as<uint8_t>();
  static_assert(sizeof(irs_hip_hit) == sizeof(Hit), "irs_hip_hit is the device's Hit");
  if (hits) *hits = reinterpret_cast<const irs_hip_hit*>(h);
  if (k_stride) *k_stride = b->k_max;
  if (counts) *counts = reinterpret_cast<const uint32_t*>(h + cnt_off);
  if (total_hits) *total_hits = reinterpret_cast<const uint64_t*>(h + tot_off);
  return IRS_HIP_OK;
}

void irs_hip_batch_destroy(irs_hip_batch* b) {
  if (!b) return;
  settle(b);
  rt::set_device(b->seg->device);
  // Its buffers go back to the pool: every piece of queued work that touches them must be
  // through — the batch's own last run (ev_done), a plan queued ahead, copies out of d_out.
  // Not the whole stream: the caller may have queued the NEXT batch behind this one.
  bool waited = true;
  if (b->ran) waited = b->ev_done_ready && rt::event_sync(b->ev_done);
  if (b->planned || b->plan_pending)
    waited = waited && b->ev_planned_ready && rt::event_sync(b->ev_planned);
  if (b->ev_used_pending) waited = waited && rt::event_sync(b->ev_used);
  if (b->host_pending) waited = waited && rt::event_sync(b->ev_host);
  if (!waited && b->ran) rt::sync(b->stream);
  if (b->events_ready)
    for (auto& e : b->ev) rt::event_destroy(e);
  if (b->ev_done_ready) rt::event_destroy(b->ev_done);
  if (b->ev_planned_ready) rt::event_destroy(b->ev_planned);
  if (b->ev_used_ready) rt::event_destroy(b->ev_used);
  if (b->ev_up_ready) rt::event_destroy(b->ev_up);
  if (b->ev_host_ready) rt::event_destroy(b->ev_host);
  delete b;
}

static int query_batch_impl(irs_hip_segment* seg, const irs_hip_query* queries, uint32_t nq,
                        const irs_hip_term_scorer* terms, uint32_t n_entries,
                        irs_hip_hit* hits, uint32_t k_stride, uint32_t* counts,
                        uint64_t* total_hits) {
  irs_hip_batch* b = nullptr;
  int rc = irs_hip_batch_create(seg, queries, nq, terms, n_entries, &b);
  if (rc != IRS_HIP_OK) return rc;
  rc = irs_hip_batch_run(b, nullptr);
  if (rc == IRS_HIP_OK) rc = irs_hip_batch_results(b, hits, k_stride, counts, total_hits);
  irs_hip_batch_destroy(b);
  return rc;
}

static int merge_topk_impl(int32_t device, const void* const* d_lists, const void* const* d_counts,
                       const uint32_t* seg_ids, uint32_t n_lists, uint32_t n_queries,
                       uint32_t k, void* d_out, void* d_out_seg, void* d_out_counts,
                       void* stream) {
  if (!d_lists || !d_counts || !seg_ids || !n_lists || n_lists > 16 || !n_queries || !k ||
      !d_out || !d_out_seg || !d_out_counts)
    return IRS_HIP_EINVAL;
  if (uint64_t(n_lists) * k > kMergeMax) return IRS_HIP_EUNSUPPORTED;
  if (!device_usable(device)) return IRS_HIP_EHIP;
  MergeLists ml{};
  for (uint32_t i = 0; i < n_lists; ++i) {
    ml.hits[i] = static_cast<const Hit*>(d_lists[i]);
    ml.counts[i] = static_cast<const uint32_t*>(d_counts[i]);
    ml.seg_ids[i] = seg_ids[i];
  }
  const size_t smem = merge_smem_bytes(n_lists, k);
  if (!big_smem(k_merge_topk, smem)) return IRS_HIP_EHIP;
  RT_LAUNCH(k_merge_topk, n_queries, kThreads, smem, static_cast<rt::stream_t>(stream), ml,
            n_lists, k, static_cast<Hit*>(d_out), static_cast<uint32_t*>(d_out_seg),
            static_cast<uint32_t*>(d_out_counts));
  return rt::last_error_ok() ? IRS_HIP_OK : IRS_HIP_EHIP;
}


// ---- the exported entry points: nothing C++ leaves this library (status codes only) ----
int irs_hip_device_arch(int32_t device, char* buf, size_t cap) {
  return guarded([&] { return device_arch_impl(device, buf, cap); });
}
int irs_hip_segment_open(const irs_hip_segment_desc* d, irs_hip_segment** out) {
  return guarded([&] { return segment_open_impl(d, out); });
}
int irs_hip_decode_term(irs_hip_segment* seg, uint32_t term, uint32_t* docs, uint32_t* freqs, uint32_t cap, uint32_t* count) {
  return guarded([&] { return decode_term_impl(seg, term, docs, freqs, cap, count); });
}
int irs_hip_decode_positions(irs_hip_segment* seg, uint32_t term, uint32_t* positions, uint64_t cap, uint64_t* count) {
  return guarded([&] { return decode_positions_impl(seg, term, positions, cap, count); });
}
int irs_hip_bit_union(irs_hip_segment* seg, const uint32_t* terms, uint32_t n_terms, uint64_t* set, uint64_t n_words, uint64_t* count) {
  return guarded([&] { return bit_union_impl(seg, terms, n_terms, set, n_words, count); });
}
int irs_hip_bit_union_counts(irs_hip_segment* seg, const uint32_t* terms, const uint32_t* offsets, uint32_t n_sets, uint64_t* counts) {
  return guarded([&] { return bit_union_counts_impl(seg, terms, offsets, n_sets, counts); });
}
int irs_hip_term_directory(irs_hip_segment* seg, uint32_t term, uint32_t* last_docs, uint64_t* offsets, uint32_t cap, uint32_t* count) {
  return guarded([&] { return term_directory_impl(seg, term, last_docs, offsets, cap, count); });
}
int irs_hip_batch_create(irs_hip_segment* seg, const irs_hip_query* queries, uint32_t nq, const irs_hip_term_scorer* terms, uint32_t n_entries, irs_hip_batch** out) {
  return guarded([&] { return batch_create_impl(seg, queries, nq, terms, n_entries, out); });
}
int irs_hip_batch_create_multi(irs_hip_segment* const* segs, uint32_t n_segs, const irs_hip_query* queries, uint32_t nq_user, const irs_hip_term_scorer* all_terms, uint32_t n_entries, irs_hip_batch** out) {
  return guarded([&] { return batch_create_multi_impl(segs, n_segs, queries, nq_user, all_terms, n_entries, out); });
}
int irs_hip_batch_configure(irs_hip_batch* b, uint32_t tile_docs, uint32_t pilot_stride, uint32_t cand_cap) {
  return settled(b, [&] { return batch_configure_impl(b, tile_docs, pilot_stride, cand_cap); });
}
int irs_hip_batch_profile(irs_hip_batch* b, int enable) {
  return settled(b, [&] { return batch_profile_impl(b, enable); });
}
int irs_hip_batch_set_path(irs_hip_batch* b, int path) {
  return settled(b, [&] { return batch_set_path_impl(b, path); });
}
int irs_hip_batch_set_shared_threshold(irs_hip_batch* b, int enable) {
  return settled(b, [&] { return batch_set_shared_threshold_impl(b, enable); });
}
int irs_hip_batch_set_comm(irs_hip_batch* b, irs_hip_comm* comm) {
  return settled(b, [&] { return batch_set_comm_impl(b, comm); });
}
int irs_hip_batch_set_async(irs_hip_batch* b, int enable) {
  return settled(b, [&] { return batch_set_async_impl(b, enable); });
}
int irs_hip_batch_path(irs_hip_batch* b, int* path) {
  return settled(b, [&] { return batch_path_impl(b, path); });
}
int irs_hip_batch_set_paired_tiles(irs_hip_batch* b, int enable) {
  return settled(b, [&] { return batch_set_paired_tiles_impl(b, enable); });
}
int irs_hip_batch_paired_tiles(irs_hip_batch* b, int* used) {
  return settled(b, [&] { return batch_paired_tiles_impl(b, used); });
}
int irs_hip_batch_set_wand(irs_hip_batch* b, int enable) {
  return settled(b, [&] { return batch_set_wand_impl(b, enable); });
}
int irs_hip_batch_set_min_scores(irs_hip_batch* b, const float* min_scores) {
  return settled(b, [&] { return batch_set_min_scores_impl(b, min_scores); });
}
int irs_hip_term_blockmax(irs_hip_segment* seg, uint32_t term, uint32_t* max_freqs,
                          uint32_t* min_norms, uint32_t cap, uint32_t* count) {
  return guarded([&] { return term_blockmax_impl(seg, term, max_freqs, min_norms, cap, count); });
}
int irs_hip_segment_wand_source(irs_hip_segment* seg, uint64_t* from_index, uint64_t* total) {
  return guarded([&] { return segment_wand_source_impl(seg, from_index, total); });
}
int irs_hip_device_alloc(int32_t device, uint64_t bytes, void** d_out) {
  return guarded([&] {
    if (!d_out) return int(IRS_HIP_EINVAL);
    *d_out = nullptr;
    if (!device_usable(device)) return int(IRS_HIP_EHIP);
    *d_out = rt::dmalloc(bytes);
    return int(*d_out ? IRS_HIP_OK : IRS_HIP_ENOMEM);
  });
}
void irs_hip_device_free(int32_t device, void* d_ptr) {
  if (rt::set_device(device)) rt::dfree(d_ptr);
}
int irs_hip_device_upload(int32_t device, void* d_dst, const void* h_src, uint64_t bytes) {
  return guarded([&] {
    if ((!d_dst || !h_src) && bytes) return int(IRS_HIP_EINVAL);
    if (!rt::set_device(device)) return int(IRS_HIP_EHIP);
    return int(rt::h2d(d_dst, h_src, bytes, nullptr) && rt::sync(nullptr) ? IRS_HIP_OK : IRS_HIP_EHIP);
  });
}
int irs_hip_device_download(int32_t device, void* h_dst, const void* d_src, uint64_t bytes) {
  return guarded([&] {
    if ((!h_dst || !d_src) && bytes) return int(IRS_HIP_EINVAL);
    if (!rt::set_device(device)) return int(IRS_HIP_EHIP);
    return int(rt::d2h(h_dst, d_src, bytes, nullptr) && rt::sync(nullptr) ? IRS_HIP_OK : IRS_HIP_EHIP);
  });
}
int irs_hip_device_trim(int32_t device) {
  return guarded([&] {
    if (device < 0 || device >= rt::device_count() || !rt::set_device(device)) return int(IRS_HIP_EHIP);
    pool::release_all(device, false);
    pool::release_all(device, true);
    return int(IRS_HIP_OK);
  });
}
int irs_hip_device_sync(int32_t device, void* stream) {
  return guarded([&] {
    if (!rt::set_device(device)) return int(IRS_HIP_EHIP);
    return int(rt::sync(static_cast<rt::stream_t>(stream)) ? IRS_HIP_OK : IRS_HIP_EHIP);
  });
}
int irs_hip_comm_unique_id(uint8_t id[IRS_HIP_COMM_ID_BYTES]) {
  return guarded([&] { return comm_unique_id_impl(id); });
}
int irs_hip_comm_init_rank(int32_t device, const uint8_t id[IRS_HIP_COMM_ID_BYTES], int32_t n_ranks,
                           int32_t rank, irs_hip_comm** out) {
  return guarded([&] { return comm_init_rank_impl(device, id, n_ranks, rank, out); });
}
int irs_hip_comm_library(char* buf, size_t cap) {
  return guarded([&] {
    if (!buf || !cap) return int(IRS_HIP_EINVAL);
    return rt::comm::library(buf, cap) ? int(IRS_HIP_OK) : int(IRS_HIP_EHIP);
  });
}
void irs_hip_comm_destroy(irs_hip_comm* c) {
  if (!c) return;
  rt::set_device(c->device);
  rt::comm::destroy(c->h);
  delete c;
}
int irs_hip_topk_allgather(irs_hip_comm* c, const void* d_send, void* d_recv,
                           uint64_t bytes_per_rank, void* stream) {
  return guarded([&] { return topk_allgather_impl(c, d_send, d_recv, bytes_per_rank, stream); });
}
int irs_hip_batch_touched(irs_hip_batch* b, uint64_t* doc_bytes, uint64_t* positions) {
  return settled(b, [&] { return batch_touched_impl(b, doc_bytes, positions); });
}
int irs_hip_batch_plan(irs_hip_batch* b, void* stream) {
  return settled(b, [&] { return batch_plan_impl(b, stream); });
}
int irs_hip_batch_run(irs_hip_batch* b, void* stream) {
  return settled(b, [&] { return batch_run_impl(b, stream); });
}
int irs_hip_batch_results_to_host(irs_hip_batch* b, void* stream) {
  return settled(b, [&] { return batch_results_to_host_impl(b, stream); });
}
int irs_hip_batch_host_results(irs_hip_batch* b, const irs_hip_hit** hits, uint32_t* k_stride,
                               const uint32_t** counts, const uint64_t** total_hits) {
  return settled(b, [&] { return batch_host_results_impl(b, hits, k_stride, counts, total_hits); });
}
int irs_hip_batch_timings(irs_hip_batch* b, float ms[IRS_HIP_K_COUNT]) {
  return settled(b, [&] { return batch_timings_impl(b, ms); });
}
int irs_hip_batch_reruns(irs_hip_batch* b, uint32_t* count) {
  return settled(b, [&] { return batch_reruns_impl(b, count); });
}
int irs_hip_batch_work(irs_hip_batch* b, uint64_t* algorithmic_bytes, uint64_t* postings) {
  return settled(b, [&] { return batch_work_impl(b, algorithmic_bytes, postings); });
}
int irs_hip_batch_results(irs_hip_batch* b, irs_hip_hit* hits, uint32_t k_stride, uint32_t* counts, uint64_t* total_hits) {
  return settled(b, [&] { return batch_results_impl(b, hits, k_stride, counts, total_hits); });
}
int irs_hip_batch_device_results(irs_hip_batch* b, void** d_hits, void** d_counts, uint32_t* k_max) {
  return settled(b, [&] { return batch_device_results_impl(b, d_hits, d_counts, k_max); });
}
int irs_hip_batch_results_to_device(irs_hip_batch* b, void* d_hits, void* d_counts, void* stream) {
  return settled(b, [&] { return batch_results_to_device_impl(b, d_hits, d_counts, stream); });
}
int irs_hip_query_batch(irs_hip_segment* seg, const irs_hip_query* queries, uint32_t nq, const irs_hip_term_scorer* terms, uint32_t n_entries, irs_hip_hit* hits, uint32_t k_stride, uint32_t* counts, uint64_t* total_hits) {
  return guarded([&] { return query_batch_impl(seg, queries, nq, terms, n_entries, hits, k_stride, counts, total_hits); });
}
int irs_hip_merge_topk(int32_t device, const void* const* d_lists, const void* const* d_counts, const uint32_t* seg_ids, uint32_t n_lists, uint32_t n_queries, uint32_t k, void* d_out, void* d_out_seg, void* d_out_counts, void* stream) {
  return guarded([&] { return merge_topk_impl(device, d_lists, d_counts, seg_ids, n_lists, n_queries, k, d_out, d_out_seg, d_out_counts, stream); });
}

}  // extern "C"
