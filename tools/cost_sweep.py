#!/usr/bin/env python3
"""Dev utility: plain disjunctions under both execution paths — joined posting streams (join.h)
and work items (score.h) — for batches of different sizes and different amounts of term sharing,
back to back on one index.  Prints, per batch shape, what a cost rule can be fitted on: the
postings of the DISTINCT streams (what k_join decodes and writes once), the postings the queries
reference (what either scoring kernel reads), the (unit, doc tile) pairs, and the step time of
each path.  VERDICT r03 item 3: the joined default must not lose to `--path items`.

  python tools/cost_sweep.py --docs 10000000 --shapes 1000x4d,128x8d,16x8d,1000x8s,128x8s,16x8s
  (QxT + d: no term shared by two queries, s: the bench's log-uniform draw with sharing)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def distinct_queries(n_queries, n_terms, lo, hi, seed):
    """log-uniform ranks in [lo, hi], every rank used at most once in the whole batch"""
    rng = np.random.default_rng(seed)
    need = n_queries * n_terms
    assert need <= hi - lo + 1, "not enough ranks for a batch without shared terms"
    w = 1.0 / np.arange(lo, hi + 1, dtype=np.float64)     # log-uniform density
    picked = rng.choice(np.arange(lo, hi + 1), size=need, replace=False, p=w / w.sum())
    return picked.reshape(n_queries, n_terms).astype(np.uint32)


def max_tf(seg, lo_rank):
    """largest frequency among the postings of rank lo_rank (decoded by the test oracle)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import oracle
        _, f = oracle.decode_term(seg.doc_file, seg.metas[lo_rank - 1], seg.layout)
        return int(f.max())
    except Exception:  # noqa: BLE001
        return -1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--shapes", default="1000x4d,128x8d,16x8d,1000x8s,128x8s,16x8s")
    ap.add_argument("--scorer", default="bm25", choices=["bm25", "tfidf"])
    ap.add_argument("--mean-len", type=int, default=0, help="0: the bench corpus (100)")
    ap.add_argument("--lo-rank", type=int, default=16,
                    help="most frequent rank the queries draw from (1: with --mean-len 1000 the top "
                         "terms' frequencies pass 64 — entries with the two low tf bits in use)")
    args = ap.parse_args()
    import torch  # noqa: F401  (one HIP runtime per process)

    from iresearch_amd import _lib, search, synth
    from iresearch_amd.search import BM25, TFIDF
    t0 = time.perf_counter()
    kw = dict(mean_len=args.mean_len, stddev_len=args.mean_len // 3) if args.mean_len else {}
    seg = synth.build_segment(args.docs, 4096, **kw)
    sr = search.SegmentReader.from_synth(seg)
    print("index built in %.1f s; largest frequency of a queried term: %d" % (
        time.perf_counter() - t0, max_tf(seg, args.lo_rank)), flush=True)
    df = np.asarray(seg.metas["docs_count"]).astype(np.int64)
    scorer = BM25() if args.scorer == "bm25" else TFIDF(True)
    st = search.SegmentStats(seg.docs_with_field, seg.total_term_freq, df)
    tiles = (args.docs + 12287) // 12288
    for shape in args.shapes.split(","):
        q, rest = shape.split("x")
        nq, nt, kind = int(q), int(rest[:-1]), rest[-1]
        if kind == "d":
            ranks = distinct_queries(nq, nt, args.lo_rank, 4096, 7)
        else:
            ranks = synth.make_queries(nq, nt, args.lo_rank, 4096, synth.SEED + 2)
        rows = ranks.astype(np.int64) - 1
        refs = int(df[rows].sum())
        distinct = int(df[np.unique(rows)].sum())
        line = "%-9s refs %7.1f M  distinct %7.1f M  unit-tiles %7d " % (
            shape, refs / 1e6, distinct / 1e6, nq * tiles)
        out = {}
        for name, path in (("items", _lib.PATH_ITEMS), ("joined", _lib.PATH_JOINED),
                           ("auto", _lib.PATH_AUTO)):
            arrays = search.prepare_disjunctions(rows, scorer, [st], [sr], args.k)
            b = search.QueryBatch(sr, arrays).set_path(path).profile(True)
            b.run()
            res = b.results()
            if "ref" in out:
                assert all(np.array_equal(x, y) for x, y in zip(out["ref"], res)), (shape, name)
            out["ref"] = res
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                b.run()
            b.results()
            dt = (time.perf_counter() - t0) / args.steps * 1e3
            took = "joined" if b.path() == _lib.PATH_JOINED else "items"
            line += " | %s %.3f ms (%s; stages %s)" % (
                name, dt, took, " ".join("%.2f" % x for x in b.timings()))
            b.close()
        print(line, flush=True)


if __name__ == "__main__":
    main()
