#!/usr/bin/env python3
"""How many doc TILES of a disjunction could a threshold skip (pruning in the
joined path, "tile bounds from k_join's own per-tile maxima")?

Host-only analysis (numpy + the test oracle; nothing on the GPU).  For OR-of-T top-k queries the
exact k-th score theta is taken from the oracle — the best threshold any rising estimate could
ever reach — and the bound of a 12288-doc tile is the sum over the query's terms of the EXACT
largest posting score of the term inside the tile (tighter than a block-max (max tf, min norm)
pair).  A tile can be skipped iff its bound is below theta.  Reported: the share of tiles, and of
the postings inside them, that even this perfect test could skip.

  python tools/tile_potential.py --docs 1000000 --terms 3 --k 100 --topic-docs 16384 --topic-percent 85
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
TILE = 12288


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=24)
    ap.add_argument("--terms", type=int, default=3)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--topic-docs", type=int, default=4096)
    ap.add_argument("--topic-percent", type=int, default=85)
    ap.add_argument("--topic-terms", type=int, default=12)
    args = ap.parse_args()
    import oracle
    import parity
    from iresearch_amd import search, synth
    from iresearch_amd.search import BM25, Or, by_term
    seg = synth.build_segment(args.docs, 4096, keep_postings=True, topic_docs=args.topic_docs,
                              topic_percent=args.topic_percent, topic_terms=args.topic_terms)
    view = parity.oracle_view(seg)
    st = [parity.segment_stats(seg)]
    ranks = synth.make_queries(args.queries, args.terms, 16, 4096, synth.SEED + 2)
    scorer = BM25()
    osc = parity.oracle_scorer(scorer)
    nt = (args.docs + TILE) // TILE + 1
    tot_tiles = dead_tiles = tot_post = dead_post = 0
    for row in ranks:
        terms = [int(r) - 1 for r in row]
        prep = search.prepare([Or([by_term(t) for t in terms])], scorer, st)[0]
        metas = parity.metas_for(seg, terms)
        hits, _ = oracle.search([view], metas[None, :], oracle.OP_OR, osc, args.k)
        theta = float(hits["score"].min()) if len(hits) >= args.k else 0.0
        bound = np.zeros(nt)
        per = []
        for t, (kind, c0, nc, nl) in zip(terms, prep.scorers):
            d, f = seg.postings[t + 1]
            sc = c0 - c0 / (1.0 + f / (nc + nl * seg.norms[d - 1].astype(np.float64)))
            tile = (d - 1) // TILE
            tm = np.zeros(nt)
            np.maximum.at(tm, tile, sc)
            bound += tm
            per.append((tile, len(d)))
        dead = bound < theta
        tot_tiles += nt
        dead_tiles += int(dead.sum())
        for tile, n in per:
            tot_post += n
            dead_post += int(dead[tile].sum())
    print("docs %d, OR-%d, k=%d, runs of %d docs with %d%% of the tokens from %d topic terms: "
          "%.1f%% of the (query, tile) pairs could be skipped with the exact k-th score as the "
          "threshold, holding %.1f%% of the queries' postings" % (
              args.docs, args.terms, args.k, args.topic_docs, args.topic_percent, args.topic_terms,
              100.0 * dead_tiles / tot_tiles, 100.0 * dead_post / max(tot_post, 1)))


if __name__ == "__main__":
    main()
