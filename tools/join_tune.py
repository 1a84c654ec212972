#!/usr/bin/env python3
"""Dev utility: the bench index built ONCE, then the headline batch (OR-of-8 BM25 top-1000)
timed under several builds of libirs_hip.so and execution settings, back to back on one GPU.

  python tools/join_tune.py --runs base:items,base:512,base:1024,w8p4:1024

A run is LIB:SETTING — LIB = `base` (iresearch_amd/csrc/libirs_hip.so) or NAME for
gpurun_variants/libirs_hip_NAME.so; SETTING = `items` (work-item path), `exact[THREADS]` (joined
posting streams, one-pass exact kernel) or the threads per workgroup of the two-pass joined path.  The first run's hits are the reference every
other run must reproduce bit for bit.
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--terms", type=int, default=8)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--runs", default="base:items,base:512,base:1024")
    ap.add_argument("--scorer", default="bm25", choices=["bm25", "tfidf", "bm15"])
    args = ap.parse_args()
    import torch  # noqa: F401  (one HIP runtime per process, see _lib.lib)

    from iresearch_amd import _lib, search, synth
    from iresearch_amd.search import BM25, TFIDF, Or, by_term
    t0 = time.perf_counter()
    seg = synth.build_segment(args.docs, 4096)
    print("index built in %.1f s" % (time.perf_counter() - t0), flush=True)
    ranks = synth.make_queries(args.queries, args.terms, 16, 4096, synth.SEED + 2)
    filters = [Or([by_term(int(r) - 1) for r in row]) for row in ranks]
    scorer = {"bm25": BM25(), "tfidf": TFIDF(True), "bm15": BM25(1.2, 0.0)}[args.scorer]
    st = search.SegmentStats(seg.docs_with_field, seg.total_term_freq,
                             np.asarray(seg.metas["docs_count"]))
    prep = search.prepare(filters, scorer, [st])
    ref = None
    libs = {}
    for run in args.runs.split(","):
        name, setting = run.split(":")
        if name not in libs:
            path = (os.path.join(ROOT, "iresearch_amd", "csrc", "libirs_hip.so") if name == "base"
                    else os.path.join(ROOT, "gpurun_variants", "libirs_hip_%s.so" % name))
            L = _lib.bind(ctypes.CDLL(path))
            libs[name] = (L, search.SegmentReader.from_synth(seg, L=L))
        L, sr = libs[name]
        b = sr.batch(prep, args.k).profile(True)
        # settings: items | exact[threads] | [threads]  (the joined path at 256 / 512 / 1024 threads)
        # (a trailing "-np": the joined plain disjunctions on 32-bit tiles, not on paired tiles)
        if setting.endswith("-np"):
            setting = setting[:-3]
            b.set_paired_tiles(False)
        if setting == "items":
            b.set_path(_lib.PATH_ITEMS)
        else:
            os.environ["IRS_HIP_JOIN_THREADS"] = (setting[5:] if setting.startswith("exact") else setting) or "1024"
            b.set_path(_lib.PATH_JOINED)
        b.run()
        try:
            hits, counts, totals = b.results()
            if ref is None:
                ref = (hits.copy(), counts.copy(), totals.copy())
            same = (np.array_equal(ref[0], hits) and np.array_equal(ref[1], counts)
                    and np.array_equal(ref[2], totals))
        except Exception as e:   # an ablation build whose results make no sense: timings only
            same = "ERROR(%s)" % type(e).__name__
        ms = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            b.run()
            ms.append(b.timings())
        ms[-1] = b.timings()
        dt = (time.perf_counter() - t0) / args.steps
        avg = np.mean(ms, axis=0)
        alg, post = b.work()
        print("%-14s path %d  step %.2f ms  qps %.0f  plan/join %.2f pilot %.2f score %.2f select %.2f"
              "  A/(join+score) %.0f GB/s  same_as_first=%s reruns=%d" % (
                  run, b.path(), dt * 1e3, args.queries / dt, *avg,
                  alg / (avg[0] + avg[2]) / 1e6, same, b.reruns()), flush=True)
        b.close()


if __name__ == "__main__":
    main()
