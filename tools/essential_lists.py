#!/usr/bin/env python3
"""MaxScore split of the bench batch (VERDICT r04 "next" 1a): how many of a disjunction's postings
sit in NON-ESSENTIAL lists — lists whose summed score upper bounds stay below the threshold, so
that a doc found only there can never reach the top k (the min / WAND lambda of
`block_disjunction`, disjunction.hpp:1133-1167, applied per TERM instead of per doc window).

Host-only analysis (numpy; nothing on the GPU, no kernel).  For every query of the headline batch
(1000 x OR-8, ranks log-uniform in [16, 4096], BM25, top-1000, 10 M docs) it computes
  * every doc's exact score (bm25.cpp:348-353 with the norm cache), the exact k-th score theta_k
    and the score of rank 3k (theta_3k: what the pilot threshold aims at, ~3k candidates);
  * per term the EXACT largest posting score of its list (a per-stream maximum of T[tf][norm]:
    query independent up to c0) and the loose bound c0 (tf -> inf);
  * the split that leaves out the most postings: the subset S of the 8 terms with
    sum(bound(S)) < theta and the largest sum(df(S)) (256 subsets, brute force);
  * the docs the essential lists hold (the reference's wand-mode "hits"), and of those the ones
    whose essential partial score + sum(bound(S)) >= theta: the docs that must be LOOKED UP in
    the non-essential lists;
  * the same for SMALLER non-essential sets — the most frequent lists first while
    sum(bound) <= alpha * theta_3k — which trade postings left out against docs to look up.

  python tools/essential_lists.py --docs 10000000 --queries 1000 > profiles/r05_essential_lists.txt
"""
import argparse
import itertools
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ALPHAS = (0.25, 0.4, 0.5, 0.6, 0.75, 0.9)
G = {}


def one_query(qi):
    seg, ranks, scorer, cache, norms, k, tmax, subsets = (G[x] for x in (
        "seg", "ranks", "scorer", "cache", "norms", "k", "tmax", "subsets"))
    N = seg.num_docs
    row = ranks[qi]
    acc = np.zeros(N + 1, np.float32)
    per = []
    for r in row:
        r = int(r)
        d, f = seg.postings[r]
        st = scorer.collect(seg.docs_with_field, len(d), seg.total_term_freq)
        c0 = float(scorer.term_scorer(st)[1])
        sc = (c0 - c0 / (1.0 + f * cache[norms[d - 1]])).astype(np.float32)
        acc[d] += sc
        per.append((d, sc, c0, c0 * tmax[r], len(d)))
    live = np.flatnonzero(acc)
    union = int(live.size)
    vals = acc[live]
    top = np.partition(vals, vals.size - 3 * k)[vals.size - 3 * k:]
    top.sort()
    th3k, thk = float(top[0]), float(top[-k])
    df = np.array([p[4] for p in per])
    ub = np.array([p[3] for p in per])
    loose = np.array([p[2] for p in per])
    P = int(df.sum())

    def best(theta, bound):
        bs, bd = (), 0
        for s in subsets:
            if s and bound[list(s)].sum() < theta and df[list(s)].sum() > bd:
                bs, bd = s, int(df[list(s)].sum())
        return bs, bd

    def split(s, theta):   # -> (non-essential postings, docs in the essential lists, lookups)
        ess = np.zeros(N + 1, np.float32)
        nes = set(s)
        for i, (d, sc, _, _, _) in enumerate(per):
            if i not in nes:
                ess[d] += sc
        e = ess[ess > 0]
        slack = float(ub[list(s)].sum()) if s else 0.0
        return int(df[list(s)].sum()) if s else 0, int(e.size), int(np.count_nonzero(e + slack >= theta)) if s else 0

    out = {"P": P, "union": union, "thk": thk, "th3k": th3k, "ubsum": float(ub.sum())}
    for tag, theta in (("k", thk), ("3k", th3k)):
        s, _ = best(theta, ub)
        out[tag] = (len(s),) + split(s, theta)
    out["loose_k"] = best(thk, loose)[1]
    # smaller non-essential sets: most frequent lists first while the bounds stay within
    # alpha * theta_3k; the test a doc must pass is against theta_3k (the run-time threshold)
    order = np.argsort(-df)
    for a in ALPHAS:
        s, ssum = [], 0.0
        for i in order:
            if ssum + ub[i] <= a * th3k:
                s.append(int(i))
                ssum += ub[i]
        out[a] = (len(s),) + split(tuple(s), th3k)
    return qi, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--terms", type=int, default=8)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--rows", type=int, default=40, help="per-query rows printed")
    args = ap.parse_args()
    from iresearch_amd import synth
    from iresearch_amd.search import BM25
    t0 = time.time()
    seg = synth.build_segment(args.docs, 4096, keep_postings=True)
    sys.stderr.write("index built in %.1f s\n" % (time.time() - t0))
    ranks = synth.make_queries(args.queries, args.terms, 16, 4096, synth.SEED + 2)
    scorer = BM25()
    norms = seg.norms.astype(np.int64)
    probe = scorer.collect(seg.docs_with_field, 1, seg.total_term_freq)
    nc, nl = float(probe.norm_const), float(probe.norm_length)
    cache = np.zeros(256)
    cache[1:] = 1.0 / (nc + nl * np.arange(1, 256))   # bm25.cpp:404-409
    tmax = {}
    for r in sorted(set(int(x) for x in ranks.reshape(-1))):
        d, f = seg.postings[r]   # the stream's largest T[tf][norm] = 1 - 1/(1 + tf * cache[norm])
        tmax[r] = float(np.max(1.0 - 1.0 / (1.0 + f * cache[norms[d - 1]])))
    subsets = [s for r in range(args.terms + 1) for s in itertools.combinations(range(args.terms), r)]
    G.update(seg=seg, ranks=ranks, scorer=scorer, cache=cache, norms=norms, k=args.k, tmax=tmax,
             subsets=subsets)
    sys.stderr.write("bounds ready after %.1f s\n" % (time.time() - t0))
    res = {}
    with mp.get_context("fork").Pool(args.procs) as pool:
        for n, (qi, out) in enumerate(pool.imap_unordered(one_query, range(len(ranks)), chunksize=4)):
            res[qi] = out
            if n % 100 == 99:
                sys.stderr.write("%d queries, %.0f s\n" % (n + 1, time.time() - t0))
    nq = len(res)
    rows = [res[i] for i in range(nq)]
    P = sum(r["P"] for r in rows)
    N = args.docs
    k = args.k
    print("# MaxScore split of the headline batch: %d queries x OR-%d, BM25, top-%d, %d docs "
          "(tools/essential_lists.py)" % (nq, args.terms, k, N))
    print("# bound of a term = c0 * (largest T[tf][norm] among ITS postings) — exact per stream; "
          "theta_k = exact k-th score, theta_3k = exact score of rank 3k (the pilot's aim)")
    print("postings per query (all lists)                 %12.0f" % (P / nq))
    print("docs matched per query (union)                 %12.0f" % (sum(r["union"] for r in rows) / nq))
    print("theta_k / theta_3k / sum of the 8 bounds (means) %10.3f %8.3f %8.3f" % (
        np.mean([r["thk"] for r in rows]), np.mean([r["th3k"] for r in rows]),
        np.mean([r["ubsum"] for r in rows])))

    def block(key):
        ne_t = sum(r[key][0] for r in rows) / nq
        ne_p = sum(r[key][1] for r in rows)
        ed = sum(r[key][2] for r in rows) / nq
        lk = sum(r[key][3] for r in rows) / nq
        print("non-essential terms per query                  %12.2f" % ne_t)
        print("postings in non-essential lists                %12.0f  = %.1f %% of all postings" % (
            ne_p / nq, 100.0 * ne_p / P))
        print("postings in essential lists                    %12.0f  = %.1f %%" % (
            (P - ne_p) / nq, 100.0 * (P - ne_p) / P))
        print("docs held by the essential lists               %12.0f  (wand-mode hits)" % ed)
        print("of those: essential partial + NE bounds >= th  %12.0f  (docs to look up)" % lk)

    print("-- largest non-essential set at theta_k (brute force over the 256 subsets)")
    block("k")
    print("-- largest non-essential set at theta_3k")
    block("3k")
    print("-- loose bounds (c0 = score at tf -> inf) at theta_k: %.1f %% of all postings non-essential" % (
        100.0 * sum(r["loose_k"] for r in rows) / P))
    print()
    print("# smaller non-essential sets: the most frequent lists first while sum(bound) <= alpha * theta_3k;")
    print("# a doc of the essential lists is looked up iff partial + sum(bound) >= theta_3k")
    print("# alpha  ne_terms  ne_postings(%)  essential_postings  essential_docs  lookups  lookups/essential_docs")
    for a in ALPHAS:
        ne_t = sum(r[a][0] for r in rows) / nq
        ne_p = sum(r[a][1] for r in rows)
        ed = sum(r[a][2] for r in rows) / nq
        lk = sum(r[a][3] for r in rows) / nq
        print("  %.2f   %6.2f   %12.1f   %16.0f   %13.0f  %8.0f   %.4f" % (
            a, ne_t, 100.0 * ne_p / P, (P - ne_p) / nq, ed, lk, lk / max(ed, 1)))
    print()
    print("# first %d queries: q postings union theta_k theta_3k sum_bounds | at theta_k: "
          "ne_terms ne_postings ess_docs lookups | at theta_3k: the same | alpha 0.5: the same" % args.rows)
    for i, r in enumerate(rows[:args.rows]):
        print("%4d %9d %9d %7.3f %7.3f %7.3f | %d %9d %8d %7d | %d %9d %8d %7d | %d %9d %8d %7d" % (
            i, r["P"], r["union"], r["thk"], r["th3k"], r["ubsum"], *r["k"], *r["3k"], *r[0.5]))
    sys.stderr.write("done in %.0f s\n" % (time.time() - t0))


if __name__ == "__main__":
    main()
