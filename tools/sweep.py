#!/usr/bin/env python3
"""Dev utility: build the bench index once, then time several batch
configurations (tile size, pilot stride) back to back on one GPU.
  python tools/sweep.py --docs 10000000 --configs 8192:16,4096:16,16384:16,8192:64
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
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--configs", default="4096:16")
    ap.add_argument("--nocheck", action="store_true")
    ap.add_argument("--op", default="or", choices=["or", "and", "mm", "phrase"],
                    help="Or / And / Or(min_match=terms-1) / by_phrase of --terms consecutive words")
    ap.add_argument("--lo-rank", type=int, default=16)
    ap.add_argument("--hi-rank", type=int, default=4096)
    ap.add_argument("--terms", type=int, default=8)
    ap.add_argument("--layout", type=int, default=1, help="0 = scalar (1_5), 1 = simd4 (1_5simd)")
    ap.add_argument("--scorer", default="bm25", choices=["bm25", "tfidf", "bm15"])
    ap.add_argument("--lib", default=None, help="alternative build of libirs_hip.so (A/B runs)")
    ap.add_argument("--wand", action="store_true", help="also time the batch with block-max pruning")
    ap.add_argument("--clustered", action="store_true", help="bursty posting lists (topic docs)")
    ap.add_argument("--path", default="auto", choices=["auto", "items", "joined"],
                    help="irs_hip_batch_set_path: work items / block-driven kernels, or joined streams "
                         "wherever a unit can take them")
    ap.add_argument("--touched", action="store_true",
                    help="And / by_phrase: one more run that counts the bytes actually decoded")
    args = ap.parse_args()
    import torch

    from iresearch_amd import _lib, search, synth
    from iresearch_amd.search import BM25, TFIDF, And, Or, by_phrase, by_term
    L = _lib.bind(ctypes.CDLL(args.lib)) if args.lib else _lib.lib()
    t0 = time.perf_counter()
    kw = dict(topic_docs=4096, topic_percent=85, topic_terms=12) if args.clustered else {}
    seg = synth.build_segment(args.docs, 4096, layout=args.layout,
                              with_positions=args.op == "phrase", **kw)
    t1 = time.perf_counter()
    sr = search.SegmentReader.from_synth(seg, L=L)
    print("index built in %.1f s, staged in %.2f s (%.0f MB .doc, %.0f MB .pos, %.0f MB in HBM)" % (
        t1 - t0, time.perf_counter() - t1, seg.doc_file.size / 1e6,
        0 if seg.pos_file is None else seg.pos_file.size / 1e6, sr.device_bytes() / 1e6),
        flush=True)
    ranks = synth.make_queries(args.queries, args.terms, args.lo_rank, args.hi_rank,
                               synth.SEED + 2)
    if args.op == "phrase":
        filters = [by_phrase([int(r) - 1 for r in row]) for row in ranks]
    elif args.op == "and":
        filters = [And([by_term(int(r) - 1) for r in row]) for row in ranks]
    elif args.op == "mm":
        filters = [Or([by_term(int(r) - 1) for r in row], min_match=args.terms - 1) for row in ranks]
    else:
        filters = [Or([by_term(int(r) - 1) for r in row]) for row in ranks]
    scorer = {"bm25": BM25(), "tfidf": TFIDF(True), "bm15": BM25(1.2, 0.0)}[args.scorer]
    st = search.SegmentStats(seg.docs_with_field, seg.total_term_freq,
                             np.asarray(seg.metas["docs_count"]))
    prep = search.prepare(filters, scorer, [st])
    ref = None
    for cfg in args.configs.split(","):
        tile, stride = (int(x) for x in cfg.split(":"))
        path = {"auto": _lib.PATH_AUTO, "items": _lib.PATH_ITEMS, "joined": _lib.PATH_JOINED}[args.path]
        b = sr.batch(prep, args.k).configure(tile, stride, 0).set_path(path).profile(True)
        b.run()
        hits, counts, totals = b.results()
        print("   path: %s" % ("joined" if b.path() == _lib.PATH_JOINED else "items"), flush=True)
        if ref is None:
            ref = hits.copy()
        same = bool(np.array_equal(ref, hits))
        ms = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            b.run()
            ms.append(b.timings())
        dt = (time.perf_counter() - t0) / args.steps
        avg = np.mean(ms, axis=0)
        alg, post = b.work()
        print("tile=%d stride=%d  step %.2f ms  qps %.0f  plan %.2f pilot %.2f score %.2f select %.2f"
              "  score GB/s %.1f  same_hits=%s reruns=%d" % (tile, stride, dt * 1e3, args.queries / dt,
                                                              *avg, alg / avg[2] / 1e6, same,
                                                              b.reruns()), flush=True)
        print("   hits/query: mean %.0f max %d" % (float(np.mean(totals)), int(np.max(totals))),
              flush=True)
        if args.touched:
            b.profile(3).run()
            td, tp = b.touched()
            print("   touched: %.1f MB .doc + norms (%.1f%% of the terms' %.1f MB), %.1f MB positions; "
                  "%.1f GB/s of touched bytes" % (td / 1e6, 100.0 * td / max(1, alg), alg / 1e6,
                                                  tp / 1e6, (td + tp) / avg[2] / 1e6), flush=True)
        b.close()
        if args.wand:
            b = sr.batch(prep, args.k).configure(tile, stride, 0).set_wand(True).profile(True)
            b.run()
            whits, wcounts, wtotals = b.results()
            ms = []
            t0 = time.perf_counter()
            for _ in range(args.steps):
                b.run()
                ms.append(b.timings())
            dt = (time.perf_counter() - t0) / args.steps
            avg = np.mean(ms, axis=0)
            print("   WAND: step %.2f ms  plan %.2f pilot %.2f score %.2f select %.2f  same top-k=%s  "
                  "docs evaluated %.1f%% of the exhaustive run's hits" % (
                      dt * 1e3, *avg, bool(np.array_equal(whits, hits)),
                      100.0 * float(np.sum(wtotals)) / max(1.0, float(np.sum(totals)))), flush=True)
            b.close()


if __name__ == "__main__":
    main()
