"""Host logic above the C ABI that needs no device: the vectorised per-batch prepare of a
serving loop (search.prepare_disjunctions) against the term-by-term mirror of the reference's
by_term::prepare + BM25::collect / TFIDF::collect (search.prepare)."""
import numpy as np
import pytest

from iresearch_amd import search
from iresearch_amd.search import BM25, TFIDF, Or, by_term


class _Seg:
    def __init__(self, n_terms):
        self.metas = np.zeros(n_terms)


@pytest.mark.parametrize("scorer", [BM25(), BM25(1.2, 0.0), BM25(0.0, 0.75), TFIDF(False), TFIDF(True)])
def test_prepare_disjunctions_equals_term_by_term(scorer):
    rng = np.random.default_rng(7)
    # two segments with different vocabularies: ordinals >= 300 are absent from the second
    stats = [search.SegmentStats(100_000, 9_700_000, rng.integers(0, 50_000, 400)),
             search.SegmentStats(60_000, 5_100_000, rng.integers(0, 30_000, 300))]
    segs = [_Seg(400), _Seg(300)]
    rows = rng.integers(0, 400, (64, 5))
    slow = search.QueryArrays.from_prepared(
        segs, search.prepare([Or([by_term(int(t), 1.5) for t in row]) for row in rows], scorer, stats), 17)
    fast = search.prepare_disjunctions(rows, scorer, stats, segs, 17, boost=1.5)
    assert slow.k == fast.k == 17
    assert slow.queries.tobytes() == fast.queries.tobytes()
    for name in search.TERM_SCORER.names:
        assert np.array_equal(slow.terms[name], fast.terms[name]), name
    assert (fast.terms["term"][1] == search.NO_TERM).any()


@pytest.mark.parametrize("scorer", [BM25(), BM25(1.2, 0.0), TFIDF(False), TFIDF(True)])
def test_prepare_filters_equals_term_by_term(scorer):
    """The vectorised prepare of ANY flat filter list (what bench.py --tasks runs per step) against
    the term-by-term mirror: boolean kinds mixed in one list, phrases in one of their own."""
    from iresearch_amd.search import And, by_phrase
    rng = np.random.default_rng(11)
    stats = [search.SegmentStats(100_000, 9_700_000, rng.integers(0, 50_000, 400)),
             search.SegmentStats(60_000, 5_100_000, rng.integers(0, 30_000, 300))]
    segs = [_Seg(400), _Seg(300)]
    booleans = []
    for i in range(96):
        n = int(rng.integers(1, 13))
        subs = [by_term(int(t), float(rng.choice([1.0, 0.5, 2.25]))) for t in rng.integers(0, 400, n)]
        kind = i % 4
        if kind == 0:
            booleans.append(subs[0])
        elif kind == 1:
            booleans.append(Or(subs, merge=int(rng.integers(0, 3))))
        elif kind == 2:
            booleans.append(Or(subs, min_match=int(rng.integers(2, 5))))
        else:
            booleans.append(And(subs))
    phrases = [by_phrase([int(t) for t in rng.integers(0, 400, int(rng.integers(1, 6)))],
                         boost=float(rng.choice([1.0, 3.0]))) for _ in range(40)]
    phrases.append(by_phrase([5, 9, 2], offsets=[0, 2, 5]))
    for filters in (booleans, phrases):
        slow = search.QueryArrays.from_prepared(segs, search.prepare(filters, scorer, stats), 9)
        fast = search.prepare_filters(filters, scorer, stats, segs, 9)
        assert slow.queries.tobytes() == fast.queries.tobytes()
        for name in search.TERM_SCORER.names:
            assert np.array_equal(slow.terms[name], fast.terms[name]), name


@pytest.mark.parametrize("scorer", [BM25(), BM25(1.2, 0.0), TFIDF(False)])
def test_prepare_expansions_statistics(scorer):
    """The scored terms of a multi-term filter carry the statistics limited_sample_collector::score
    gives them — the field's over the whole index, the term's from the segments where the term is
    SCORED — equal to what scorer.collect / term_scorer yield term by term."""
    rng = np.random.default_rng(2)
    stats = [search.SegmentStats(100_000, 9_700_000, rng.integers(1, 50_000, 400)),
             search.SegmentStats(60_000, 5_100_000, rng.integers(1, 30_000, 400))]
    visits = [[np.arange(10, 60, dtype=np.uint32), np.arange(10, 45, dtype=np.uint32)],
              [np.arange(100, 104, dtype=np.uint32), np.zeros(0, np.uint32)]]
    for limit in (16, 3, 0):
        prep = search.prepare_expansions(visits, limit, scorer, stats, boost=1.5)
        for p, per_seg in zip(prep, visits):
            assert sum(len(x) for x in p.scored_in) == min(limit, sum(len(v) for v in per_seg))
            for t, sc in zip(p.scored, p.scorers):
                dwt = sum(int(stats[s].docs_count[t]) for s in range(2) if t in p.scored_in[s])
                want = scorer.term_scorer(scorer.collect(160_000, dwt, 14_800_000), 1.5)
                assert tuple(sc) == tuple(want), (limit, t)
            for s in range(2):      # every visited term is scored or unscored, never both
                assert sorted(list(p.scored_in[s]) + p.unscored_in[s].tolist()) == per_seg[s].tolist()


def test_prepare_expansions_one_segment_equals_the_collector():
    """One segment: the array form (all filters at once) chooses what limited_sample_collector
    chooses filter by filter — ragged visits, equal docs_counts, empty and short visits."""
    rng = np.random.default_rng(7)
    stats = [search.SegmentStats(500_000, 40_000_000, rng.integers(1, 40, 3000))]   # many ties
    visits = [[np.sort(rng.choice(3000, int(n), replace=False)).astype(np.uint32)]
              for n in [0, 1, 5, 16, 17, 40, 333, 1200, 2, 0, 64]]
    scorer = BM25()
    for limit in (16, 1, 0, 5000):
        prep = search.prepare_expansions(visits, limit, scorer, stats)
        assert len(prep) == len(visits)
        for p, (va,) in zip(prep, visits):
            counts = np.asarray(stats[0].docs_count)[va.astype(np.int64)]
            want = sorted(int(va[off]) for _, off in search.scored_states([counts], limit))
            assert p.scored == want and p.scored_in == [set(want)]
            assert p.slots.tolist() == want and p.present.shape == (1, len(want)) and p.present.all()
            assert sorted(want + p.unscored_in[0].tolist()) == va.tolist()
            assert p.visited_in[0] is va or np.array_equal(p.visited_in[0], va)
            for t, sc in zip(p.scored, p.scorers):
                assert tuple(sc) == tuple(scorer.term_scorer(
                    scorer.collect(500_000, int(stats[0].docs_count[t]), 40_000_000)))
        arrays = search.expansion_arrays([None], prep, 10)
        at = 0
        for q, p in enumerate(prep):
            n = max(1, len(p.scored))
            assert tuple(arrays.queries[q]) == (search.OP_OR, n, at, 10, 1, search.MERGE_SUM)
            got = arrays.terms[0, at:at + n]
            if p.scored:
                assert got["term"].tolist() == p.scored
                assert [tuple(x) for x in got[["kind", "c0", "norm_const", "norm_length"]].tolist()] == \
                    [tuple(float(v) if i else int(v) for i, v in enumerate(sc)) for sc in p.scorers]
            else:
                assert got["term"].tolist() == [search.NO_TERM]
            at += n


def test_prepare_expansions_long_visits_prefiltered():
    """Long visits (wildcards: thousands of mostly rare terms) take the array form's prefilter — a
    docs_count cut that leaves every filter its `limit` largest keys — and filters the cut is too
    high for fall back to the plain form: the same choice as the collector's, filter by filter."""
    rng = np.random.default_rng(11)
    n_terms = 60_000
    dc = np.maximum(1, (2_000_000 / np.arange(1, n_terms + 1) ** 1.1)).astype(np.int64)   # Zipf: many ties at the tail
    rng.shuffle(dc)
    stats = [search.SegmentStats(3_000_000, 250_000_000, dc)]
    rare = np.flatnonzero(dc <= 20)
    visits = [[np.sort(rng.choice(n_terms, int(n), replace=False)).astype(np.uint32)]
              for n in [3000, 2500, 40, 0, 5000, 7, 1800]]
    visits.append([np.sort(rng.choice(rare, 900, replace=False)).astype(np.uint32)])   # only rare terms
    visits.append([np.sort(rng.choice(rare, 10, replace=False)).astype(np.uint32)])    # fewer than limit
    scorer = BM25()
    for limit in (16, 3, 64):
        prep = search.prepare_expansions(visits, limit, scorer, stats)
        for p, (va,) in zip(prep, visits):
            want = sorted(int(va[off]) for _, off in search.scored_states([dc[va.astype(np.int64)]], limit))
            assert p.scored == want, limit
            assert p.n_unscored(0) == len(va) - len(want)
            assert sorted(want + p.unscored_in[0].tolist()) == va.tolist()
