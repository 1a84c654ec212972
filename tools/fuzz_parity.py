#!/usr/bin/env python3
"""Randomised differential run of the C ABI against the oracle (test infrastructure: imports
oracle/ through tests/parity.py).  Every round builds a fresh segment (random size, vocabulary,
layout, clustering, wand data, positions, every second round a random set of DELETED documents —
the segment's DocumentMask, which the oracle applies as SegmentReaderImpl::mask does), a batch of
random Or / And / min-match / by_term
filters with random boosts and merge types (or by_phrase filters), a random scorer and k, and
checks the results as the parity tests do; the same batch is then re-run with block-max pruning
(top-k must not change) and with the k-th score pushed down as irs::score::Min.

  python tools/fuzz_parity.py --seconds 120            # on the GPU (libirs_hip.so)
  python tools/fuzz_parity.py --sim --seconds 60       # on the CPU emulator
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--sim", action="store_true")
    ap.add_argument("--max-docs", type=int, default=400_000)
    args = ap.parse_args()
    import parity
    from iresearch_amd import _lib, search, synth
    from iresearch_amd.search import (BM25, MERGE_MAX, MERGE_MIN, MERGE_SUM, TFIDF, And, Or,
                                      by_phrase, by_term)
    if args.sim:
        L = _lib.bind(ctypes.CDLL(os.path.join(ROOT, "tests", "sim", "libirs_hip_sim.so")))
        args.max_docs = min(args.max_docs, 40_000)
    else:
        L = _lib.lib()
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    rounds = queries = 0
    while time.time() < t_end:
        docs = int(rng.integers(2_000, args.max_docs))
        max_rank = int(rng.choice([32, 128, 512, 2048]))
        layout = int(rng.integers(0, 2))
        clustered = bool(rng.integers(0, 2))
        wand_count = int(rng.integers(0, 2))
        positions = bool(rng.integers(0, 3) == 0)
        kw = dict(topic_docs=int(rng.choice([512, 2048, 8192])), topic_percent=85,
                  topic_terms=12) if clustered else {}
        seg = synth.build_segment(docs, max_rank, layout=layout, wand_count=wand_count,
                                  with_positions=positions, seed=int(rng.integers(1, 1 << 30)), **kw)
        if rounds % 2 == 1:   # deleted docs: a random share, a run of consecutive ones, the ends
            share = float(rng.choice([0.001, 0.05, 0.5]))
            run0 = int(rng.integers(1, docs))
            seg.doc_mask = np.concatenate([
                (rng.choice(docs, max(1, int(docs * share)), replace=False) + 1).astype(np.uint32),
                np.arange(run0, min(docs, run0 + int(rng.integers(1, 700))) + 1, dtype=np.uint32),
                np.array([1, docs], np.uint32)[:int(rng.integers(0, 3))]])
        sr = search.SegmentReader.from_synth(seg, L=L)
        st = [parity.segment_stats(seg)]
        scorer = [BM25(), BM25(1.2, 0.0), BM25(0.0, 0.0), BM25(2.0, 1.0), TFIDF(False),
                  TFIDF(True)][int(rng.integers(0, 6))]
        k = int(rng.choice([1, 10, 100, 1000]))

        def term():
            # mostly frequent ranks (Zipf), sometimes rare or absent ones
            r = int(rng.integers(0, 4))
            hi = [8, 64, max_rank, max_rank + 40][r]
            return by_term(int(rng.integers(0, hi)), float(rng.choice([1.0, 1.0, 0.5, 2.5, 0.0])))

        def merge():
            return int(rng.choice([MERGE_SUM, MERGE_SUM, MERGE_MAX, MERGE_MIN]))

        if positions and rng.integers(0, 2):
            filters = [by_phrase([int(t) for t in rng.integers(0, min(max_rank, 48), int(rng.integers(1, 5)))])
                       for _ in range(12)]
            prep = search.prepare(filters, scorer, st)
            b = sr.batch(prep, k)
            hits, counts, totals = (x.copy() for x in b.run().results())
            parity.check_phrase_segment(seg, filters, scorer, k, hits, counts, totals)
        else:
            filters = []
            for _ in range(16):
                n = int(rng.integers(1, 9))
                subs = [term() for _ in range(n)]
                kind = int(rng.integers(0, 4))
                if kind == 0 or n == 1:
                    filters.append(Or(subs, merge=merge()))
                elif kind == 1:
                    if rng.integers(0, 3) == 0:   # a rare lead against frequent terms (lead blocks in pieces)
                        subs = [by_term(int(rng.integers(max_rank // 2, max_rank)))] + \
                               [by_term(int(rng.integers(0, 6))) for _ in range(int(rng.integers(1, 4)))]
                    filters.append(And(subs[:int(rng.integers(2, 6))] if n > 2 else subs, merge=merge()))
                elif kind == 2:
                    filters.append(Or(subs, min_match=int(rng.integers(2, n + 1)), merge=merge()))
                else:
                    filters.append(subs[0])
            prep = search.prepare(filters, scorer, st)
            b = sr.batch(prep, k)
            hits, counts, totals = (x.copy() for x in b.run().results())
            parity.check_single_segment(seg, filters, scorer, k, hits, counts, totals)
            if rounds % 3 == 0:   # the same results through page-locked host memory, a run later
                hh, hc, ht = b.run().results_to_host().host_results()
                assert np.array_equal(hc, counts) and np.array_equal(ht, totals), "host results: counts"
                for q in range(len(filters)):
                    assert np.array_equal(hh[q, :counts[q]], hits[q, :counts[q]]), "host results: hits"
            # the other execution path (work items / block-driven kernels): checked against the
            # oracle like the first run (which took the joined streams wherever it could)
            ib = sr.batch(prep, k).set_path(_lib.PATH_ITEMS)
            ih, ic, it = (x.copy() for x in ib.run().results())
            parity.check_single_segment(seg, filters, scorer, k, ih, ic, it)
            assert np.array_equal(ic, counts) and np.array_equal(it, totals), "paths: counts"
            ib.close()
            # joined streams wherever a unit is eligible (whatever the cost rules would deal) —
            # against the oracle, counts as before
            # (on paired doc tiles whatever the segment's size: k_join_score<kJKHalf> + k_join_rescore)
            jb = sr.batch(prep, k).set_path(_lib.PATH_JOINED).set_paired_tiles(2)
            jh, jc, jt = (x.copy() for x in jb.run().results())
            parity.check_single_segment(seg, filters, scorer, k, jh, jc, jt)
            assert np.array_equal(jc, counts) and np.array_equal(jt, totals), "joined: counts"
            # ... and on 32-bit tiles: bit for bit the same lists
            ub = sr.batch(prep, k).set_path(_lib.PATH_JOINED).set_paired_tiles(0)
            uh, uc, ut = ub.run().results()
            assert np.array_equal(uc, jc) and np.array_equal(ut, jt) and np.array_equal(uh, jh), "paired tiles"
            ub.close()
            jb.close()
            # block-max pruning (on the work-item / block-driven kernels): the same top-k, bit for bit
            wb = sr.batch(prep, k).set_path(_lib.PATH_ITEMS).set_wand(True)
            wh, wc, wt = wb.run().results()
            assert np.array_equal(ic, wc), "wand: counts"
            for q in range(len(filters)):
                assert np.array_equal(ih[q, :ic[q]], wh[q, :ic[q]]), ("wand: top-k", q)
            assert (wt <= totals).all()
            wb.close()
            # every third round: the same filters over several segments in ONE batch
            # (irs_hip_batch_create_multi; statistics over all of them) — per segment the oracle's
            if rounds % 3 == 0:
                extra = [synth.build_segment(int(rng.integers(1_000, max(2_000, docs // 2))), max_rank,
                                             layout=layout, first_doc=docs * (i + 1),
                                             seed=int(rng.integers(1, 1 << 30)))
                         for i in range(int(rng.integers(1, 3)))]
                msegs = [seg] + extra
                readers = [sr] + [search.SegmentReader.from_synth(x, L=L) for x in extra]
                mprep = search.prepare(filters, scorer, [parity.segment_stats(x) for x in msegs])
                mb = search.QueryBatch(readers, mprep, k)
                mh, mc, mt = (x.copy() for x in mb.run().results())
                for i, x in enumerate(msegs):
                    parity.check_single_segment(x, filters, scorer, k, mh[i], mc[i], mt[i], msegs)
                mb.close()
                # one threshold per query for all segments: the merged top k must not change
                sb = search.QueryBatch(readers, mprep, k).set_shared_threshold(True)
                sh, sc, stot = sb.run().results()
                assert np.array_equal(stot, mt) and (sc <= mc).all(), "shared threshold: counts"
                plain = search.merge_topk_host([(mh[i], mc[i]) for i in range(len(msegs))], k)
                shared = search.merge_topk_host([(sh[i], sc[i]) for i in range(len(msegs))], k)
                assert plain == shared, "shared threshold: merged top-k"
                sb.close()
                for r in readers[1:]:
                    r.close()
        # every fourth round: scored multi-term filters (scored_terms_limit) over one or two
        # segments — the collector's choice, totals over ALL visited terms, the zero-score fill
        if rounds % 4 == 0:
            xsegs, xreaders = [seg], [sr]
            if rng.integers(0, 2):
                x = synth.build_segment(int(rng.integers(1_000, max(2_000, docs // 2))), max_rank,
                                        layout=layout, first_doc=docs, seed=int(rng.integers(1, 1 << 30)))
                xsegs.append(x)
                xreaders.append(search.SegmentReader.from_synth(x, L=L))
            limit = int(rng.choice([0, 1, 3, 16]))
            visits = []
            for _ in range(6):
                per_seg = []
                for x in xsegs:
                    n_terms = len(x.metas)
                    lo = int(rng.integers(0, n_terms))
                    ords = np.arange(lo, min(n_terms, lo + int(rng.integers(0, 60))), dtype=np.uint32)
                    ords = ords[rng.random(len(ords)) < 0.8]          # (a wildcard skips terms)
                    per_seg.append(ords[np.asarray(x.metas["docs_count"])[ords] > 0])
                visits.append(per_seg)
            xk = int(rng.choice([1, 10, 100]))
            xprep = search.prepare_expansions(visits, limit, scorer, [parity.segment_stats(x) for x in xsegs])
            xh, xc, xt = search.execute_expansions(xreaders, xprep, xk)
            parity.check_expansions(xsegs, visits, limit, scorer, xk, xh, xc, xt)
            for r in xreaders[1:]:
                r.close()
            queries += len(visits)
        # irs::score::Min = the k-th score: the same top-k again
        kth = np.array([hits[q, counts[q] - 1]["score"] if counts[q] else 0.0
                        for q in range(len(filters))], np.float32)
        h2, c2, t2 = b.set_min_scores(kth).run().results()
        assert np.array_equal(c2, counts) and np.array_equal(t2, totals), "min score: counts"
        for q in range(len(filters)):
            assert np.array_equal(h2[q, :counts[q]], hits[q, :counts[q]]), ("min score: top-k", q)
        b.close()
        sr.close()
        rounds += 1
        queries += len(filters)
    print("fuzz ok: %d rounds, %d queries, seed %d" % (rounds, queries, args.seed))


if __name__ == "__main__":
    main()
