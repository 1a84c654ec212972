#!/usr/bin/env python3
"""Dev utility: the bulk APIs next to the query path on one GPU — irs_hip_decode_term,
irs_hip_bit_union, irs_hip_decode_positions — timed end to end (device allocation, kernel,
copy back); run it under `rocprofv3 --kernel-trace --stats` for the kernel times alone."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch  # noqa: F401  (one HIP runtime per process)

    from iresearch_amd import search, synth
    seg = synth.build_segment(args.docs, 4096, with_positions=True)
    sr = search.SegmentReader.from_synth(seg)
    dc = seg.metas["docs_count"].astype(np.int64)
    tf = seg.metas["freq"].astype(np.int64)

    def timed(fn):
        fn()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        return (time.perf_counter() - t0) / args.reps

    for term in (0, 15, 255):
        dt = timed(lambda: sr.decode_term(term))
        print("decode_term rank %4d: %9d postings  %.2f ms  %.2f G postings/s (incl. copy back)"
              % (term + 1, dc[term], dt * 1e3, dc[term] / dt / 1e9), flush=True)
    for term in (0, 15, 255):
        dt = timed(lambda: sr.decode_positions(term))
        print("decode_positions rank %4d: %9d positions %.2f ms  %.2f G positions/s" %
              (term + 1, tf[term], dt * 1e3, tf[term] / dt / 1e9), flush=True)
    words = (args.docs + 1 + 63) // 64
    for lo, hi in ((15, 79), (0, 1024), (1024, 4096)):
        terms = np.arange(lo, hi, dtype=np.uint32)
        n = int(dc[lo:hi].sum())
        dt = timed(lambda: sr.bit_union(terms, words))
        print("bit_union terms [%d, %d): %10d postings  %.2f ms  %.2f G postings/s" %
              (lo, hi, n, dt * 1e3, n / dt / 1e9), flush=True)


if __name__ == "__main__":
    main()
