"""Parity helpers shared by the CPU (emulator) and GPU tests: run the same
prepared queries through the C ABI and through the oracle and compare."""
from __future__ import annotations

import numpy as np

import oracle
from iresearch_amd import search
from iresearch_amd._lib import OP_AND, OP_OR

REL_TOL = 1e-5  # BASELINE.json north_star: top-k within 1e-5 relative


def oracle_scorer(scorer) -> oracle.Scorer:
    if isinstance(scorer, search.BM25):
        return oracle.Scorer(oracle.SCORER_BM25, float(scorer.k), float(scorer.b), 0)
    return oracle.Scorer(oracle.SCORER_TFIDF, 0.0, 0.0, int(scorer.with_norms))


def oracle_op(flt, op):
    """Or with min_match > 1 -> ORC_OP_MINMATCH | (min_match << 8); the filter's merge type
    in bits 24..25 (ORC_MERGE_*)."""
    mm = int(getattr(flt, "min_match", 0) or 0)
    op = oracle.OP_MINMATCH | (mm << 8) if op == oracle.OP_MINMATCH else op
    return op | (int(getattr(flt, "merge", 0) or 0) << 24)


def oracle_view(seg) -> oracle.SegmentView:
    return oracle.SegmentView(seg.doc_file, seg.norms, seg.layout, seg.num_docs,
                              seg.docs_with_field, seg.total_term_freq,
                              getattr(seg, "norm_width", 1), getattr(seg, "wand_count", 0),
                              getattr(seg, "pos_file", None), getattr(seg, "pos_one_based", False),
                              getattr(seg, "doc_mask", None))


def segment_stats(seg) -> search.SegmentStats:
    return search.SegmentStats(seg.docs_with_field, seg.total_term_freq,
                               np.asarray(seg.metas["docs_count"]))


def metas_for(seg, terms):
    out = np.zeros(len(terms), oracle.TERM_META)
    for i, t in enumerate(terms):
        if t is not None and 0 <= t < len(seg.metas):
            for name in oracle.TERM_META.names:
                out[i][name] = seg.metas[t][name]
    return out


def check_single_segment(seg, filters, scorer, k, hits, counts, totals, all_segs=None):
    """Compares GPU results of `filters` on `seg` with the oracle.  Statistics are
    global over `all_segs` (default: just this segment)."""
    all_segs = all_segs or [seg]
    osc = oracle_scorer(scorer)
    view = oracle_view(seg)
    dwf = sum(s.docs_with_field for s in all_segs)
    ttf = sum(s.total_term_freq for s in all_segs)
    for q, flt in enumerate(filters):
        op, subs = search._terms_of(flt)
        op = oracle_op(flt, op)
        terms = [s.term for s in subs]
        boosts = [s.boost for s in subs]
        metas = metas_for(seg, terms)
        dwt = [sum(int(s.metas[t]["docs_count"]) if 0 <= t < len(s.metas) else 0
                   for s in all_segs) for t in terms]
        scores, matched = oracle.score_all(view, metas, op, osc, dwf, dwt, ttf, boosts)
        n_match = int(matched.sum())
        assert int(totals[q]) == n_match, ("total hits", q, int(totals[q]), n_match)
        n = int(counts[q])
        assert n == min(k, n_match), ("count", q, n, k, n_match)
        if n == 0:
            continue
        h = hits[q, :n]
        docs = h["doc"].astype(np.int64)
        assert len(set(docs.tolist())) == n, ("duplicate docs", q)
        assert matched[docs].all(), ("unmatched doc returned", q)
        ref = scores[docs]
        rel = np.abs(h["score"] - ref) / np.maximum(np.abs(ref), 1e-30)
        assert rel.max() <= REL_TOL, ("score mismatch", q, float(rel.max()))
        # ordered (score desc, doc asc)
        s, d = h["score"], h["doc"]
        assert ((s[:-1] > s[1:]) | ((s[:-1] == s[1:]) & (d[:-1] < d[1:]))).all(), ("order", q)
        # set parity around the k-th score
        ms = np.sort(scores[matched.astype(bool)])[::-1]
        thr = ms[n - 1]
        must = np.nonzero(matched.astype(bool) & (scores > thr * (1 + 2 * REL_TOL)))[0]
        assert np.isin(must, docs).all(), ("missing doc above the k-th score", q)
        assert (ref >= thr * (1 - 2 * REL_TOL)).all(), ("doc below the k-th score", q)


def oracle_topk(segs, filters, scorer, k):
    """The oracle's own harness run (index-search.cpp:719-787) per query."""
    osc = oracle_scorer(scorer)
    views = [oracle_view(s) for s in segs]
    out = []
    for flt in filters:
        op, subs = search._terms_of(flt)
        op = oracle_op(flt, op)
        terms = [s.term for s in subs]
        metas = np.stack([metas_for(s, terms) for s in segs])
        hits, total = oracle.search(views, metas, op, osc, k, [s.boost for s in subs])
        out.append((hits, total))
    return out



def check_phrase_segment(seg, phrases, scorer, k, hits, counts, totals, all_segs=None):
    """by_phrase results of the C ABI on `seg` vs the oracle's exhaustive phrase run
    (phrase frequency per doc is exact; scores within REL_TOL)."""
    all_segs = all_segs or [seg]
    osc = oracle_scorer(scorer)
    view = oracle_view(seg)
    dwf = sum(s.docs_with_field for s in all_segs)
    ttf = sum(s.total_term_freq for s in all_segs)
    for q, ph in enumerate(phrases):
        metas = metas_for(seg, ph.terms)
        dwt = [sum(int(s.metas[t]["docs_count"]) if 0 <= t < len(s.metas) else 0
                   for s in all_segs) for t in ph.terms]
        scores, pf = oracle.score_all_phrase(view, metas, ph.offsets, osc, dwf, dwt, ttf, ph.boost)
        matched = pf > 0
        n_match = int(matched.sum())
        assert int(totals[q]) == n_match, ("total hits", q, int(totals[q]), n_match)
        n = int(counts[q])
        assert n == min(k, n_match), ("count", q, n, k, n_match)
        if n == 0:
            continue
        h = hits[q, :n]
        docs = h["doc"].astype(np.int64)
        assert len(set(docs.tolist())) == n, ("duplicate docs", q)
        assert matched[docs].all(), ("doc without the phrase returned", q)
        ref = scores[docs]
        rel = np.abs(h["score"] - ref) / np.maximum(np.abs(ref), 1e-30)
        assert rel.max() <= REL_TOL, ("score mismatch", q, float(rel.max()))
        s, d = h["score"], h["doc"]
        assert ((s[:-1] > s[1:]) | ((s[:-1] == s[1:]) & (d[:-1] < d[1:]))).all(), ("order", q)
        ms = np.sort(scores[matched])[::-1]
        thr = ms[n - 1]
        must = np.nonzero(matched & (scores > thr * (1 + 2 * REL_TOL)))[0]
        assert np.isin(must, docs).all(), ("missing doc above the k-th score", q)
        assert (ref >= thr * (1 - 2 * REL_TOL)).all(), ("doc below the k-th score", q)


def check_expansions(segs, visits, limit, scorer, k, hits, counts, totals):
    """Scored multi-term expansion filters (MultiTermQuery::execute over the states
    limited_sample_collector left, multiterm_query.cpp:112-184) through the product's host layer
    (search.execute_expansions) against the oracle: scored states by the oracle's own restatement
    of the collector, per segment the scored terms' disjunction scored exhaustively
    (orc_score_all) + the unscored terms' bit_union; total hits, doc sets, scores, order, the
    members around the k-th score, and — where fewer than k docs score — that the zero-score
    rest is the smallest doc ids (this path's tie order).
    visits[q][s] = the term ordinals the filter's visitor yields in segment s."""
    osc = oracle_scorer(scorer)
    dwf = sum(s.docs_with_field for s in segs)
    ttf = sum(s.total_term_freq for s in segs)
    for q, per_seg in enumerate(visits):
        dcs = [[int(sg.metas[int(t)]["docs_count"]) for t in v] for v, sg in zip(per_seg, segs)]
        scored = oracle.scored_states(dcs, limit)
        scored_in = [set() for _ in segs]
        for s, off in scored:
            scored_in[s].add(int(per_seg[s][off]))
        # limited_sample_collector::score: a term's statistics from the segments where it is scored
        dwt_of = {}
        for s, ts in enumerate(scored_in):
            for t in ts:
                dwt_of[t] = dwt_of.get(t, 0) + int(segs[s].metas[t]["docs_count"])
        for s, seg in enumerate(segs):
            view = oracle_view(seg)
            st = sorted(scored_in[s])
            n1 = seg.num_docs + 1
            scores = np.zeros(n1, np.float32)
            matched = np.zeros(n1, bool)
            if st:
                sc, m = oracle.score_all(view, metas_for(seg, st), oracle.OP_OR, osc, dwf,
                                         [dwt_of[t] for t in st], ttf, [1.0] * len(st))
                scores, matched = sc[:n1].copy(), m[:n1].astype(bool)
            un = [int(t) for t in per_seg[s] if int(t) not in scored_in[s]]
            if un:
                bits, _ = oracle.bit_union(seg.doc_file, [seg.metas[t] for t in un], seg.layout, True,
                                           (seg.num_docs + 64) // 64,
                                           wand_count=int(getattr(seg, "wand_count", 0)))
                ub = np.unpackbits(bits.view(np.uint8), bitorder="little")[:n1].astype(bool)
                mask = getattr(seg, "doc_mask", None)
                if mask is not None:                   # (the oracle's bit_union is the unmasked one)
                    gone = np.asarray(mask, np.int64)
                    ub[gone[(gone >= 1) & (gone <= seg.num_docs)]] = False
                matched = matched | ub
            n_match = int(matched.sum())
            assert int(totals[s, q]) == n_match, ("total hits", q, s, int(totals[s, q]), n_match)
            n = int(counts[s, q])
            assert n == min(k, n_match), ("count", q, s, n, k, n_match)
            if n == 0:
                continue
            h = hits[s, q, :n]
            docs = h["doc"].astype(np.int64)
            assert len(set(docs.tolist())) == n and matched[docs].all(), ("docs", q, s)
            ref = scores[docs]
            rel = np.abs(h["score"] - ref) / np.maximum(np.abs(ref), 1e-30)
            assert (rel[ref > 0] <= REL_TOL).all() and (h["score"][ref == 0] == 0).all(), ("score", q, s)
            sc_, d_ = h["score"], h["doc"]
            assert ((sc_[:-1] > sc_[1:]) | ((sc_[:-1] == sc_[1:]) & (d_[:-1] < d_[1:]))).all(), ("order", q, s)
            ms = np.sort(scores[matched])[::-1]
            thr = ms[n - 1]
            must = np.nonzero(matched & (scores > thr * (1 + 2 * REL_TOL)))[0]
            assert np.isin(must, docs).all(), ("missing doc above the k-th score", q, s)
            assert (ref >= thr * (1 - 2 * REL_TOL)).all(), ("doc below the k-th score", q, s)
            zeros = docs[ref == 0]
            if zeros.size:      # the zero-score rest: the smallest ids among the docs that score 0
                cand = np.nonzero(matched & (scores == 0))[0]
                assert np.array_equal(np.sort(zeros), cand[:zeros.size]), ("zero-score fill", q, s)
