#!/usr/bin/env python3
"""How much could block-max WAND (wanderator formats_10.cpp:2424-2824,
block_disjunction's min/WAND lambda disjunction.hpp:1133-1167) prune on the benchmark corpus?

Host-only analysis (numpy + the test oracle's skip reader; nothing on the GPU): the segment
is indexed WITH a BM25 scorer, so every level-0 skip entry holds the block's
(max freq, min norm) payload.  For OR-of-8 top-k queries the exact k-th score theta is
taken from the oracle; a doc range can be skipped iff the sum over the query terms of the
block-max score of the block covering it is <= theta.  Reported: the fraction of postings
that live in blocks which could be skipped even with PERFECT knowledge of theta."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=32)
    ap.add_argument("--terms", type=int, default=8)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--topic-docs", type=int, default=0,
                    help="clustered corpus: docs per topic run (0 = the i.i.d. benchmark corpus)")
    ap.add_argument("--topic-percent", type=int, default=30)
    ap.add_argument("--topic-terms", type=int, default=16)
    args = ap.parse_args()
    import oracle
    import parity
    from iresearch_amd import search, synth
    from iresearch_amd.search import BM25, Or, by_term

    seg = synth.build_segment(args.docs, 4096, keep_postings=True, wand_count=1,
                              wand_kind=synth.WAND_MIN_NORM, topic_docs=args.topic_docs,
                              topic_percent=args.topic_percent, topic_terms=args.topic_terms)
    view = parity.oracle_view(seg)
    st = [parity.segment_stats(seg)]
    ranks = synth.make_queries(args.queries, args.terms, 16, 4096, synth.SEED + 2)
    scorer = BM25()
    osc = parity.oracle_scorer(scorer)
    total_post = skippable_post = 0
    present_dead_post = present_skip_post = 0
    for row in ranks:
        terms = [int(r) - 1 for r in row]
        flt = Or([by_term(t) for t in terms])
        prep = search.prepare([flt], scorer, st)[0]
        metas = parity.metas_for(seg, terms)
        hits, _ = oracle.search([view], metas[None, :], oracle.OP_OR, osc, args.k)
        theta = float(hits["score"].min())
        # upper bound of every term over doc ranges: ub[t][doc] = block-max score of the
        # block covering doc (the last partial block / short lists: the term's global bound)
        ub = np.zeros((len(terms), seg.num_docs + 2), np.float32)
        for j, (t, (kind, c0, nc, nl)) in enumerate(zip(terms, prep.scorers)):
            d, f = seg.postings[t + 1]
            def score(tf, norm):
                return c0 - c0 / (1.0 + tf / (nc + nl * norm))
            glob = score(float(f.max()), float(seg.norms[d - 1].min()))
            ub[j, :] = glob
            if len(d) > 128:
                last, _, _, mf, mn = oracle.read_skip0(seg.doc_file, seg.metas[t], 1, True)
                lo = 1
                for b in range(len(last)):
                    ub[j, lo:int(last[b]) + 1] = score(float(mf[b]), float(mn[b]))
                    lo = int(last[b]) + 1
        bound = ub.sum(axis=0)
        dead = bound <= theta          # docs no scorer could lift above the threshold
        # presence-based bound (what WAND's pivoting amounts to): only the terms that CONTAIN
        # the doc contribute their block-max
        pres = np.zeros(seg.num_docs + 2, np.float32)
        for j, t in enumerate(terms):
            d, _ = seg.postings[t + 1]
            pres[d] += ub[j, d]
        pdead = pres <= theta
        for j, t in enumerate(terms):
            d, _ = seg.postings[t + 1]
            total_post += len(d)
            # a block is skippable only if EVERY doc of its range is dead
            dd = dead[d]
            nb = len(d) // 128
            if nb:
                blk_dead = dd[:nb * 128].reshape(nb, 128).all(axis=1)
                skippable_post += int(blk_dead.sum()) * 128
            pd = pdead[d]
            present_dead_post += int(pd.sum())
            if nb:
                present_skip_post += int(pd[:nb * 128].reshape(nb, 128).all(axis=1).sum()) * 128
    corpus = ("clustered (runs of %d docs, %d %% of the tokens from %d topic terms)" %
              (args.topic_docs, args.topic_percent, args.topic_terms)) if args.topic_docs \
        else "i.i.d. benchmark corpus"
    print("%s, docs %d, %d OR-%d queries, k=%d: %.2f %% of the postings sit in blocks that a "
          "perfect block-max test could skip" % (corpus, args.docs, args.queries, args.terms,
                                                  args.k, 100.0 * skippable_post / max(total_post, 1)))
    print("  presence-based bound (sum of the block-maxes of the terms that contain the doc — "
          "what WAND's pivot sees): %.1f %% of the postings belong to docs that cannot reach the "
          "threshold (no full scoring needed), %.2f %% sit in blocks made of such docs only "
          "(no decode needed)" % (100.0 * present_dead_post / max(total_post, 1),
                                  100.0 * present_skip_post / max(total_post, 1)))


if __name__ == "__main__":
    main()
