#!/usr/bin/env python3
"""Dev utility (host only): how the restated CPU loop (oracle, `cpu_baseline.kind = "port"`)
scales with threads on this box — and what the container's CPU quota is."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import oracle
    import parity
    from iresearch_amd import synth
    print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        if os.path.exists(f):
            print(f, open(f).read().strip())
    docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    seg = synth.build_segment(docs, 4096)
    view = parity.oracle_view(seg)
    ranks = synth.make_queries(256, 8, 16, 4096, synth.SEED + 2)
    metas = np.stack([np.stack([parity.metas_for(seg, [int(r) - 1 for r in row])]) for row in ranks])
    sc = oracle.Scorer(oracle.SCORER_BM25, 1.2, 0.75, 0)
    for th in (1, 8, 32, 64, 128, 256):
        nq = min(256, max(8, th))
        t0 = time.perf_counter()
        oracle.search_batch([view], metas[:nq], oracle.OP_OR, sc, 1000, threads=th)
        dt = time.perf_counter() - t0
        print("threads %3d: %3d queries in %.2f s = %.1f queries/s" % (th, nq, dt, nq / dt), flush=True)


if __name__ == "__main__":
    main()
