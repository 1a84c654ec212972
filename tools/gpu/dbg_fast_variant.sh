#!/bin/sh
# tools/gpu/dbg_fast_variant.sh — builds gpurun_variants/libirs_hip_fdbg.so: k_join_fast with cycle
# counters around the phases of its tile loop (accumulate / request / barrier / epilogue / barrier),
# printed per wavefront of workgroup 7 at kernel end.  Dev tool: run with tools/join_tune.py --runs fdbg:1024.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
TMP="$(mktemp -d /tmp/irsdbg.XXXXXX)"
cp -r "$ROOT/iresearch_amd/csrc" "$TMP/csrc"; rm -f "$TMP"/csrc/*.so
python3 - "$TMP/csrc/fast.h" <<'PY'
import sys
p=sys.argv[1]
s=open(p).read()
def rep(old,new):
    global s
    assert s.count(old)==1, old[:60]
    s=s.replace(old,new)
rep('''  request(0);
  for (uint32_t u = 0; u < ntile; ++u) {
    // ---- accumulate this wavefront's entries of tile u
''','''  request(0);
  uint32_t dd0 = 0, dd1 = 0, dd2 = 0, dd3 = 0, dd4 = 0;
  for (uint32_t u = 0; u < ntile; ++u) {
    // ---- accumulate this wavefront's entries of tile u
    const uint32_t t0 = uint32_t(__builtin_readcyclecounter());
''')
rep('''    request(u + 1u);
    __syncthreads();   // B1: every accumulation of tile u has landed''','''    const uint32_t t1 = uint32_t(__builtin_readcyclecounter());
    request(u + 1u);
    const uint32_t t2 = uint32_t(__builtin_readcyclecounter());
    __syncthreads();   // B1: every accumulation of tile u has landed
    const uint32_t t3 = uint32_t(__builtin_readcyclecounter());''')
rep('''    __syncthreads();   // B2: accumulators are clear again
  }
  return hit_pk;''','''    const uint32_t t4 = uint32_t(__builtin_readcyclecounter());
    __syncthreads();   // B2: accumulators are clear again
    const uint32_t t5 = uint32_t(__builtin_readcyclecounter());
    dd0 += t1 - t0; dd1 += t2 - t1; dd2 += t3 - t2; dd3 += t4 - t3; dd4 += t5 - t4;
  }
  if (lane == 0 && blockIdx.x == 7u) {
    unsigned long long* d = g_dbg + 8u * wv;
    d[0] += dd0; d[1] += dd1; d[2] += dd2; d[3] += dd3; d[4] += dd4; d[5] += ntile;
  }
  return hit_pk;''')
rep('''// The tiles of one chunk.  Everything it needs lives in LDS''','''__device__ unsigned long long g_dbg[8 * 17];
// The tiles of one chunk.  Everything it needs lives in LDS''')
rep('''  if (tid == 0) {
    vars[kJPendQ] = pend_q;
    vars[kJPendBase] = pend_base;
    vars[kJPendN] = pend_n;
  }
  __syncthreads();
  {
    const uint32_t cap = args->cand_cap;''','''  if (blockIdx.x == 7u && lane == 0) {
    const unsigned long long* d = g_dbg + 8u * (tid >> 6);
    printf("DBG wave %u tiles %llu acc %llu req %llu b1 %llu epi %llu b2 %llu (ticks per tile)\\n", tid >> 6, d[5],
           d[0] / (d[5] + 1), d[1] / (d[5] + 1), d[2] / (d[5] + 1), d[3] / (d[5] + 1), d[4] / (d[5] + 1));
  }
  if (tid == 0) {
    vars[kJPendQ] = pend_q;
    vars[kJPendBase] = pend_base;
    vars[kJPendN] = pend_n;
  }
  __syncthreads();
  {
    const uint32_t cap = args->cand_cap;''')
open(p,'w').write(s)
PY
mkdir -p "$ROOT/gpurun_variants"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-gpu-rdc -Wno-unused-function \
  -I "$ROOT/include" -I "$TMP/csrc" -I "$TMP/csrc/hip" -o "$ROOT/gpurun_variants/libirs_hip_fdbg.so" "$TMP"/csrc/*.hip
rm -rf "$TMP"; echo built fdbg
